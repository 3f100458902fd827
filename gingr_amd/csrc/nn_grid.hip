// Exact nearest neighbour over a uniform grid of the (fixed) target cloud: the closest-point step of the ICP update
// (G/api/registration/utils/ClosestPointRegistrator.scala:33-44: closest target point per model vertex; ties -> lowest original
// index, as the reference's linear argmin).  Same distances and the same answer as nn_kernel (affinity.hip) -- separately rounded
// dx*dx + dy*dy + dz*dz, lowest original index on equal distances -- but each query tests the targets of a few grid cells instead of
// whole 64-point tiles: ~15 tests per query instead of ~740 at 50k points.
//
// Exactness.  Targets are binned on the host by c = floor((x - lo) * inv_h) per axis (clamped to the grid); the device evaluates the
// same expression for the query.  With the previous match as warm start and its distance r <= one cell, the query scans the cells the
// ball of radius r overlaps (1-8 cells): every target within r is binned in one of them (a slack of 1e-9 covers the rounding of the
// cell index), so the minimum over them and the warm candidate is the exact minimum.  Without such a bound: the cells
// [c - 1, c + 1]^3; after scanning [c - R, c + R]^3 every target not scanned is farther than R * h from the query, so the running
// minimum is exact as soon as best <= (R h)^2 (1 - 1e-9); if R = 1 does not certify it, one more pass with the R that covers
// sqrt(best), up to kMaxR; a query that needs more (far from every target, nothing in its 27 cells, not finite) is FLAGGED and left
// to nn_kernel, which runs masked right after (workgroups without a flagged query exit at once).
#include "common.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kGridBlock = 256;
constexpr int kMaxR = 3;  // largest half-width (in cells) of the block a query scans itself: 7^3 cells
// A row of cells that holds more targets than this (heaps of coincident points, a cloud that is one dense blob plus a far outlier) is
// not scanned by its one lane: the query is flagged and the tile scan, which spreads the work over a workgroup, answers it.
constexpr int kMaxRowTargets = 1024;
constexpr int64_t kBruteTargets = 4096;  // target clouds up to this size: uncertified queries scan everything inside nn_grid_kernel

__device__ __forceinline__ double grid_norm2_exact(double dx, double dy, double dz) {
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

__device__ __forceinline__ int clamp_cell(double f, int g) {  // floor(f) clamped to [0, g - 1]; NaN -> 0
    const double c = floor(f);
    return c >= (double)(g - 1) ? g - 1 : (c > 0.0 ? (int)c : 0);
}

// kLanes lanes serve one query: each takes every kLanes-th row (a run of cells along x at fixed y, z) of the block of cells being
// scanned, then the lanes combine their candidates (smallest distance, then lowest original index).  One lane per query leaves the
// chip with < 1 wave per SIMD at 50k queries and a chain of ~35 dependent loads per lane (measured 68 us at 50k); 16 lanes per query:
// 23 us; 8 lanes (lane 0 takes two rows of the first pass): 20.7 us at 50k and 5 % faster per ICP iteration at 100k, 2 % slower at
// 15k -- the launcher picks by the number of queries.  Past that the kernel is bound by the number of distinct cache lines its
// scattered loads touch (~40 per query), not by latency or instructions: forcing 8 waves per SIMD changes nothing.


template <bool COUNT, int kLanes>
__global__ __launch_bounds__(kGridBlock) void nn_grid_kernel(Cloud q, Cloud tgt, const int32_t *__restrict__ orig, NNGridDev g,
                                                             const int32_t *warm /* may alias idx: in-place warm start */, int32_t *idx,
                                                             double *__restrict__ d2out, uint8_t *__restrict__ flag,
                                                             int32_t *__restrict__ nflag, int32_t *__restrict__ nflag_next,
                                                             unsigned long long *tests) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *nflag_next = 0;  // the counter of the NEXT search (nobody reads it during this one)
    const int sub = threadIdx.x % kLanes;
    const int64_t i = (int64_t)blockIdx.x * (kGridBlock / kLanes) + threadIdx.x / kLanes;
    const bool ok = i < q.n;
    const double qx = ok ? q.x[i] : 0.0, qy = ok ? q.y[i] : 0.0, qz = ok ? q.z[i] : 0.0;
    double best = __builtin_huge_val();
    int32_t bo = INT32_MAX, bi = -1;
    unsigned long long ntests = 0;
    const double fx = (qx - g.lo[0]) * g.inv_h, fy = (qy - g.lo[1]) * g.inv_h, fz = (qz - g.lo[2]) * g.inv_h;
    const bool finite = fabs(fx) < 1e15 && fabs(fy) < 1e15 && fabs(fz) < 1e15;  // false for NaN / infinite queries
    bool flagged = ok && !finite;
    bool crowded = false;  // this lane skipped a row that holds too many targets (scan_row)
    const bool active = ok && finite;
    const double cxf = floor(fx), cyf = floor(fy), czf = floor(fz);
    const double h2 = g.h * g.h;
    auto combine = [&]() {  // the kLanes lanes of a query agree on the best candidate (all lanes of the wave take part)
#pragma unroll
        for (int off = 1; off < kLanes; off <<= 1) {
            const double d = __shfl_xor(best, off);
            const int32_t o = __shfl_xor(bo, off), p = __shfl_xor(bi, off);
            if (d < best || (d == best && o < bo)) best = d, bo = o, bi = p;
            const int other = __shfl_xor((int)crowded, off);
            crowded = crowded || other != 0;
        }
        if (crowded) flagged = true;  // a row was left out: the answer may be incomplete
    };
    auto test = [&](const GridPoint &p) {
        const double d = grid_norm2_exact(p.x - qx, p.y - qy, p.z - qz);
        if (d < best || (d == best && p.orig < bo)) best = d, bo = p.orig, bi = p.pos;
    };
    // the block of cells [c - R, c + R] per axis, clamped to the grid (an interval wholly outside the grid is empty: nrows = 0)
    struct Block {
        int x0, x1, y0, z0, ny, nrows;
    };
    auto block_of = [&](int R) {
        Block b{0, 0, 0, 0, 1, 0};
        const double xl = cxf - R, xh = cxf + R, yl = cyf - R, yh = cyf + R, zl = czf - R, zh = czf + R;
        if (xh < 0.0 || yh < 0.0 || zh < 0.0 || xl > (double)(g.g[0] - 1) || yl > (double)(g.g[1] - 1) || zl > (double)(g.g[2] - 1))
            return b;
        b.x0 = clamp_cell(xl, g.g[0]), b.x1 = clamp_cell(xh, g.g[0]);
        b.y0 = clamp_cell(yl, g.g[1]), b.z0 = clamp_cell(zl, g.g[2]);
        b.ny = clamp_cell(yh, g.g[1]) - b.y0 + 1;
        b.nrows = b.ny * (clamp_cell(zh, g.g[2]) - b.z0 + 1);
        return b;
    };
    // row r of a block: the targets [s, e) of its run of cells along x, and a lower bound (squared) of their distance to the query
    auto row_bounds = [&](const Block &b, int r, int32_t &s, int32_t &e, double &gap2) {
        const int cz = b.z0 + r / b.ny, cy = b.y0 + r % b.ny;
        const double gz = fmax(fmax((double)cz - fz, fz - (double)(cz + 1)), 0.0);
        const double gy = fmax(fmax((double)cy - fy, fy - (double)(cy + 1)), 0.0);
        gap2 = (gy * gy + gz * gz) * h2 * (1.0 - 1e-9);  // (the slack keeps a row whose nearest point could tie with the best)
        const int64_t row = ((int64_t)cz * g.g[1] + cy) * g.g[0];
        s = g.cell_start[row + b.x0], e = g.cell_start[row + b.x1 + 1];
    };
    auto scan_row = [&](int32_t s, int32_t e) {
        if (e - s > kMaxRowTargets) {
            crowded = true;
            return;
        }
        if (COUNT) ntests += (unsigned long long)(e - s);
        int32_t j = s;
        for (; j + 4 <= e; j += 4) {  // four loads in flight
            const GridPoint p0 = g.pts[j], p1 = g.pts[j + 1], p2 = g.pts[j + 2], p3 = g.pts[j + 3];
            test(p0), test(p1), test(p2), test(p3);
        }
        if (j < e) {  // up to three left: loads first (clamped addresses), tests under the count
            const GridPoint p0 = g.pts[j], p1 = g.pts[min(j + 1, e - 1)], p2 = g.pts[min(j + 2, e - 1)];
            test(p0);
            if (j + 1 < e) test(p1);
            if (j + 2 < e) test(p2);
        }
    };
    // The position this query matched last time: a valid candidate whose distance bounds the search (every lane of the query
    // evaluates it: same addresses).
    if (warm && ok) {
        const int32_t p = warm[i];
        if (p >= 0 && p < tgt.n) {
            const double d = grid_norm2_exact(tgt.x[p] - qx, tgt.y[p] - qy, tgt.z[p] - qz);
            if (d == d) best = d, bo = orig ? orig[p] : p, bi = p;
        }
    }
    // First pass, in 32-bit integer arithmetic (this is the path every query takes; kLanes lanes repeat its set-up).
    //   * With a warm bound no longer than one cell: the cells the BALL of radius sqrt(best) around the query overlaps -- per axis the
    //     query's cell and possibly one neighbour, so 1-8 cells instead of 27.  Every target at distance <= sqrt(best) is binned in
    //     one of them (the slack covers the rounding of its cell index), so the minimum over them and the warm candidate is exact;
    //     no certificate is needed.
    //   * Otherwise (first search, or the previous match is far): the block of 3 x 3 x 3 cells, certified afterwards.
    const int cxi = (int)fmin(fmax(cxf, -2.0), (double)(g.g[0] + 1)), cyi = (int)fmin(fmax(cyf, -2.0), (double)(g.g[1] + 1)),
              czi = (int)fmin(fmax(czf, -2.0), (double)(g.g[2] + 1));  // (clamped two cells outside the grid: such rows do not exist)
    const double rc = sqrt(best) * g.inv_h * (1.0 + 1e-9) + 1e-9;  // radius in cells; +inf without a warm start
    const bool ball = active && rc <= 1.0;
    int xlo = cxi - 1, xhi = cxi + 1, ylo = cyi - 1, yhi = cyi + 1, zlo = czi - 1, zhi = czi + 1;
    if (ball) {  // floor(f - rc) and floor(f + rc), which lie within one cell of floor(f)
        xlo = cxi - (fx - rc < cxf ? 1 : 0), xhi = cxi + (fx + rc >= cxf + 1.0 ? 1 : 0);
        ylo = cyi - (fy - rc < cyf ? 1 : 0), yhi = cyi + (fy + rc >= cyf + 1.0 ? 1 : 0);
        zlo = czi - (fz - rc < czf ? 1 : 0), zhi = czi + (fz + rc >= czf + 1.0 ? 1 : 0);
    }
    xlo = max(xlo, 0), ylo = max(ylo, 0), zlo = max(zlo, 0);
    xhi = min(xhi, g.g[0] - 1), yhi = min(yhi, g.g[1] - 1), zhi = min(zhi, g.g[2] - 1);
    const int ny = yhi - ylo + 1, nz = zhi - zlo + 1;  // <= 3 each
    const int nrows1 = (active && ny > 0 && nz > 0 && xhi >= xlo) ? ny * nz : 0;
    for (int r = sub; r < nrows1; r += kLanes) {
        const int rz = ny == 1 ? r : (ny == 2 ? r >> 1 : (r * 11) >> 5);  // r / ny for r < 16
        const int cy = ylo + (r - rz * ny), cz = zlo + rz;
        const double gz = fmax(fmax((double)cz - fz, fz - (double)(cz + 1)), 0.0);
        const double gy = fmax(fmax((double)cy - fy, fy - (double)(cy + 1)), 0.0);
        if ((gy * gy + gz * gz) * h2 * (1.0 - 1e-9) > best) continue;  // the row's nearest point is farther than the bound
        const int rowbase = (cz * g.g[1] + cy) * g.g[0];
        const int32_t s = g.cell_start[rowbase + xlo], e = g.cell_start[rowbase + xhi + 1];  // (the same cache line, mostly)
        scan_row(s, e);
    }
    auto scan = [&](int R) {
        const Block b = block_of(R);
        for (int r = sub; r < b.nrows; r += kLanes) {
            int32_t s, e;
            double gap2;
            row_bounds(b, r, s, e, gap2);
            if (!(gap2 > best)) scan_row(s, e);
        }
    };
    combine();
    // the 27 cells certify the minimum when best <= h^2; otherwise one pass over the block that covers sqrt(best), if a query may scan
    // that much (the decision is the same in the kLanes lanes of the query: they hold the same best)
    const bool more = active && !ball && !(best <= h2 * (1.0 - 1e-9));
    if (__any(more)) {  // (rare: none of the wave's four queries in the steady state of a registration)
        if (more) {
            const double need = ceil(sqrt(best) * g.inv_h * (1.0 + 1e-9));  // +inf when nothing was found; best is never NaN
            if (need <= (double)kMaxR)
                scan((int)need);
            else
                flagged = true;
        }
        combine();
    }
    // Small target clouds: a query the grid could not certify looks at EVERY target right here (same distances, same tie rule as the
    // masked tile scan), so no second launch has to follow (NNGridDev::brute_n; launch_nn_grid tells the caller).
    if (g.brute_n > 0 && __any(flagged)) {
        if (flagged) {
            for (int32_t j = sub; j < g.brute_n; j += kLanes) test(g.pts[j]);
            if (COUNT) ntests += (unsigned long long)((g.brute_n - sub + kLanes - 1) / kLanes);
            crowded = false;
        }
        combine();
        flagged = false;
    }
    if (ok && sub == 0) {
        if (!flagged || bi >= 0) idx[i] = bi;  // a flagged query keeps a valid warm start for the masked full scan
        if (!flagged) d2out[i] = best;
        flag[i] = flagged ? 1 : 0;
    }
    const unsigned long long fb = __ballot(flagged && sub == 0);
    if (fb && (threadIdx.x & 63) == 0) atomicAdd(nflag, (int32_t)__popcll(fb));
    if (COUNT) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ntests += __shfl_xor(ntests, off);
        if ((threadIdx.x & 63) == 0 && ntests) atomicAdd(tests, ntests);
    }
}

}  // namespace

void nn_grid_free(NNGrid *g) {
    if (g->cell_start) (void)hipFree(g->cell_start);
    if (g->pts) (void)hipFree(g->pts);
    if (g->flag) (void)hipFree(g->flag);
    if (g->nflag) (void)hipFree(g->nflag);
    *g = NNGrid{};
}

// target_xyz: host, original order, AoS [N][3]; perm[pos] = original index of the target at device position pos.  Synchronous.
int nn_grid_build(gingr_ctx *ctx, const double *target_xyz, int64_t N, const int32_t *perm, int64_t max_queries, NNGrid *g) {
    nn_grid_free(g);
    if (N < 1 || N > INT32_MAX || max_queries < 1) return GINGR_OK;  // no grid: the callers keep the full scan
    double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    int64_t nfinite = 0;
    for (int64_t k = 0; k < N; ++k) {
        const double *p = target_xyz + 3 * k;
        if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
        ++nfinite;
        for (int d = 0; d < 3; ++d) lo[d] = std::min(lo[d], p[d]), hi[d] = std::max(hi[d], p[d]);
    }
    if (nfinite == 0) return GINGR_OK;
    double size[3], maxext = 0.0;
    for (int d = 0; d < 3; ++d) size[d] = hi[d] - lo[d], maxext = std::max(maxext, size[d]);
    if (!(maxext < 1e300)) return GINGR_OK;  // absurd extents: keep the full scan
    // cell edge: ~4 cells per target over the axes the cloud really extends along (a surface then holds 2-3 targets per occupied cell)
    double h = 1.0;
    if (maxext > 0.0) {
        double vol = 1.0;
        int dims = 0;
        for (int d = 0; d < 3; ++d)
            if (size[d] > 1e-9 * maxext) vol *= size[d], ++dims;
        h = std::pow(vol / (4.0 * (double)nfinite), 1.0 / dims);
        if (!(h > 1e-12 * maxext)) h = 1e-12 * maxext;
    }
    int32_t gd[3];
    for (;;) {
        double cells = 1.0;
        for (int d = 0; d < 3; ++d) {
            const double c = std::floor(size[d] / h) + 1.0;
            gd[d] = (int32_t)std::min(c, 1024.0);
            cells *= std::min(c, 1e9);
            if (c > 1024.0) cells = 1e30;  // too fine along this axis
        }
        if (cells <= std::min(8.0 * (double)N + 4096.0, 1073741824.0)) break;  // (cell indices are 32-bit on the device)
        h *= 1.25;
    }
    const int64_t ncells = (int64_t)gd[0] * gd[1] * gd[2];
    const double inv_h = 1.0 / h;
    auto cell_of = [&](double x, int d) {  // the expression the kernel evaluates for a query (clamp_cell)
        const double c = std::floor((x - lo[d]) * inv_h);
        return c >= (double)(gd[d] - 1) ? gd[d] - 1 : (c > 0.0 ? (int32_t)c : 0);
    };
    std::vector<int32_t> start((size_t)ncells + 1, 0), cell((size_t)N);
    for (int64_t pos = 0; pos < N; ++pos) {
        const double *p = target_xyz + 3 * (int64_t)(perm ? perm[pos] : pos);
        const bool fin = std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
        const int64_t c = fin ? ((int64_t)cell_of(p[2], 2) * gd[1] + cell_of(p[1], 1)) * gd[0] + cell_of(p[0], 0) : 0;
        cell[(size_t)pos] = (int32_t)c;
        ++start[(size_t)c + 1];
    }
    for (int64_t c = 0; c < ncells; ++c) start[(size_t)c + 1] += start[(size_t)c];
    std::vector<GridPoint> pts((size_t)N);
    {
        std::vector<int32_t> fill(start.begin(), start.end() - 1);
        for (int64_t pos = 0; pos < N; ++pos) {  // ascending device position inside a cell
            const int32_t o = perm ? perm[pos] : (int32_t)pos;
            const double *p = target_xyz + 3 * (int64_t)o;
            pts[(size_t)fill[(size_t)cell[(size_t)pos]]++] = GridPoint{p[0], p[1], p[2], o, (int32_t)pos};
        }
    }
    HIP_TRY(ctx, hipMalloc(&g->cell_start, start.size() * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc(&g->pts, pts.size() * sizeof(GridPoint)));
    HIP_TRY(ctx, hipMalloc(&g->flag, (size_t)max_queries));
    HIP_TRY(ctx, hipMalloc(&g->nflag, 2 * sizeof(int32_t)));
    // on the CONTEXT's stream (a non-blocking stream does not order against the null stream a plain hipMemset runs on: a search
    // launched right behind the build -- the stateless gingr_nn -- could count its flagged queries before the counters were cleared)
    HIP_TRY(ctx, hipMemcpyAsync(g->cell_start, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(g->pts, pts.data(), pts.size() * sizeof(GridPoint), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(g->flag, 0, (size_t)max_queries, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(g->nflag, 0, 2 * sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the host arrays above go out of scope
    for (int d = 0; d < 3; ++d) g->v.lo[d] = lo[d], g->v.g[d] = gd[d];
    g->v.h = h;
    g->v.inv_h = inv_h;
    g->v.cell_start = g->cell_start;
    g->v.pts = static_cast<const GridPoint *>(g->pts);
    g->v.brute_n = N <= kBruteTargets ? (int32_t)N : 0;
    g->n = N;
    g->max_queries = max_queries;
    g->ready = true;
    return GINGR_OK;
}

// idx / d2 of every query the grid certifies; the others are flagged (g.flag, g.cur_nflag()) for the masked launch_nn that must follow
// -- unless the call returns true: the target cloud is small enough that the kernel answered them itself (nothing is flagged)
bool launch_nn_grid(gingr_ctx *ctx, Cloud query, Cloud target, const int32_t *target_orig, NNGrid &g, const int32_t *warm, int32_t *idx,
                    double *d2) {
    g.parity ^= 1;
    int32_t *cur = g.nflag + g.parity, *next = g.nflag + (g.parity ^ 1);
    TimerScope ts(ctx, 8);
    auto go = [&](auto kern, int lanes) {
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(query.n, kGridBlock / lanes)), dim3(kGridBlock), 0, ctx->stream, query, target,
                           target_orig, g.v, warm, idx, d2, g.flag, cur, next, ctx->nn_tests);
    };
    const bool many = query.n >= 32768;
    if (ctx->nn_tests) {
        if (many) go(nn_grid_kernel<true, 8>, 8);
        else go(nn_grid_kernel<true, 16>, 16);
    } else {
        if (many) go(nn_grid_kernel<false, 8>, 8);
        else go(nn_grid_kernel<false, 16>, 16);
    }
    return g.v.brute_n > 0;
}

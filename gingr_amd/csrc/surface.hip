// ICP with the surface correspondence: the reference's DEFAULT ICP method
//     ICPCorrespondence.estimate with TriangularClosestPoint (G/api/registration/config/ICP.scala:36-52,63)
//       -> ClosestPointTriangleMesh3D.closestPointCorrespondence (G/api/registration/utils/ClosestPointRegistrator.scala:75-100)
// For every template (fit) vertex p:  cp = target.operations.closestPointOnSurface(p);  v = the target VERTEX closest to cp;
// weight 0 when v is a boundary vertex (:53-55), when the vertex normals of p and v point into opposite half spaces (:57-60),
// or when the line through p along p - cp meets the template itself closer than |p - cp| (:62-72); else 1.  Only weight-1
// pairs become observations (ICP.scala:50).
//
// scalismo's mesh queries are restated (oracle/gingr_oracle.py, section f2): exact point-triangle closest point (Ericson
// 5.1.5), vertex normal = mean of the adjacent unit cell normals, boundary vertex = on an edge with one adjacent triangle,
// line-triangle intersection = Moeller-Trumbore on the infinite line with inclusive barycentric bounds (UNPINNED: scalismo's
// own intersection routine is not available; with this formulation a triangle that has p as a corner returns exactly p,
// which the reference filters out).  Arithmetic is written without FMA contraction in the oracle's operation order.
//
// Triangles live in a spatial (k-d leaf) order with one bounding box per 256-triangle tile; a wave owns 64 spatially
// coherent queries and skips every tile whose box cannot hold anything closer than what each lane already has -- the same
// exact pruning as the nearest-neighbour kernel (affinity.hip).
#include "common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

namespace {

constexpr int kTriTile = 256;
constexpr int kSurfThreads = 64;

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__device__ __forceinline__ double uniform_dd(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// closest point of triangle (A, B, C) to p: Ericson, Real-Time Collision Detection 5.1.5, with his region tests in his order but
// evaluated as selects: on a wavefront every lane lands in a different Voronoi region, so the branching form executes all seven
// paths (and their four divisions) one after the other.  Here the region picks a numerator, a denominator, a base corner and two
// edge vectors; ONE division; the result is  base + e1 * s1 + e2 * s2  -- the same floating-point expressions as the branching
// form (an edge region adds  e2 * 0  = +0, a vertex region adds two zeros).
__device__ __forceinline__ V3 closest_on_triangle(V3 p, V3 A, V3 B, V3 C) {
    const V3 ab = sub(B, A), ac = sub(C, A), bc = sub(C, B), ap = sub(p, A), bp = sub(p, B), cp = sub(p, C);
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap), d3 = dot3(ab, bp), d4 = dot3(ac, bp), d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    const bool rA = d1 <= 0.0 && d2 <= 0.0;
    const bool rB = !rA && d3 >= 0.0 && d4 <= d3;
    const bool rAB = !rA && !rB && vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0;
    const bool rC = !rA && !rB && !rAB && d6 >= 0.0 && d5 <= d6;
    const bool rAC = !rA && !rB && !rAB && !rC && vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0;
    const bool rBC = !rA && !rB && !rAB && !rC && !rAC && va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0;
    const bool vertex = rA || rB || rC;
    const double num = vertex ? 0.0 : (rAB ? d1 : (rAC ? d2 : (rBC ? (d4 - d3) : 1.0)));
    const double den = vertex ? 1.0 : (rAB ? (d1 - d3) : (rAC ? (d2 - d6) : (rBC ? ((d4 - d3) + (d5 - d6)) : ((va + vb) + vc))));
    const double q = num / den;
    const bool interior = !vertex && !rAB && !rAC && !rBC;
    const V3 base = (rB || rBC) ? B : (rC ? C : A);
    const V3 e1 = rBC ? bc : (rAC ? ac : ab);
    const double s1 = vertex ? 0.0 : (interior ? vb * q : q);
    const double s2 = interior ? vc * q : 0.0;
    return V3{(base.x + e1.x * s1) + ac.x * s2, (base.y + e1.y * s1) + ac.y * s2, (base.z + e1.z * s1) + ac.z * s2};
}

__device__ __forceinline__ double point_box_gap2(double qx, double qy, double qz, const double *__restrict__ bx) {
    const double gx = fmax(fmax(bx[0] - qx, qx - bx[3]), 0.0), gy = fmax(fmax(bx[1] - qy, qy - bx[4]), 0.0),
                 gz = fmax(fmax(bx[2] - qz, qz - bx[5]), 0.0);
    return __builtin_fma(gz, gz, __builtin_fma(gy, gy, gx * gx));
}

// cn (SoA [3][T]) = unit normal (b - a) x (c - a) of every triangle
__global__ __launch_bounds__(256) void cell_normals_kernel(Cloud v, const int32_t *__restrict__ tri, int64_t T,
                                                           double *__restrict__ cn) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int32_t a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
    const V3 A{v.x[a], v.y[a], v.z[a]}, B{v.x[b], v.y[b], v.z[b]}, C{v.x[c], v.y[c], v.z[c]};
    const V3 n = cross3(sub(B, A), sub(C, A));
    const double len = sqrt((n.x * n.x + n.y * n.y) + n.z * n.z);
    cn[t] = n.x / len;
    cn[T + t] = n.y / len;
    cn[2 * T + t] = n.z / len;
}

// vn (SoA [3][n]) = mean of the adjacent cell normals, adjacency lists in ascending ORIGINAL triangle index
__global__ __launch_bounds__(256) void vertex_normals_kernel(const int32_t *__restrict__ adj_ptr, const int32_t *__restrict__ adj_tri,
                                                             const double *__restrict__ cn, int64_t T, int64_t n,
                                                             double *__restrict__ vn) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    const int32_t b = adj_ptr[i], e = adj_ptr[i + 1];
    for (int32_t k = b; k < e; ++k) {
        const int32_t t = adj_tri[k];
        sx += cn[t];
        sy += cn[T + t];
        sz += cn[2 * T + t];
    }
    const double cnt = e > b ? (double)(e - b) : 1.0;
    vn[i] = sx / cnt;
    vn[n + i] = sy / cnt;
    vn[2 * n + i] = sz / cnt;
}

// boxes[tile] = {lo[3], hi[3]} over the corners of the triangles [tile*256, tile*256+256), followed (at boxes + 6 * ntiles) by the
// boxes of its four 64-triangle quarters [tile*4 + q] (the triangle order is a k-d order down to 64-triangle leaves)
// cn (nullable, SoA [3][T]): the unit cell normals as cell_normals_kernel writes them (same expressions), from the corners this kernel
// reads anyway -- one launch less per surface correspondence.
__global__ __launch_bounds__(256) void tri_tile_bbox_kernel(Cloud v, const int32_t *__restrict__ tri, int64_t T,
                                                            double *__restrict__ boxes, double *__restrict__ tribox,
                                                            double *__restrict__ cn) {
    __shared__ double sh[6][256];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double lo[3] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val()};
    double hi[3] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
    if (t < T) {
        V3 P[3];
        for (int c = 0; c < 3; ++c) {
            const int32_t a = tri[3 * t + c];
            const double p[3] = {v.x[a], v.y[a], v.z[a]};
            P[c] = V3{p[0], p[1], p[2]};
            for (int d = 0; d < 3; ++d) {
                lo[d] = fmin(lo[d], p[d]);
                hi[d] = fmax(hi[d], p[d]);
            }
        }
        if (cn) {
            const V3 n = cross3(sub(P[1], P[0]), sub(P[2], P[0]));
            const double len = sqrt((n.x * n.x + n.y * n.y) + n.z * n.z);
            cn[t] = n.x / len;
            cn[T + t] = n.y / len;
            cn[2 * T + t] = n.z / len;
        }
    }
    if (tribox && t < T) {  // per-triangle boxes: the scan kernels stage these (one 48-byte read) instead of rebuilding them from
        double *tb = tribox + 6 * t;  // three index loads and nine gathered coordinates per visited triangle and workgroup
        tb[0] = lo[0], tb[1] = lo[1], tb[2] = lo[2], tb[3] = hi[0], tb[4] = hi[1], tb[5] = hi[2];
    }
    for (int d = 0; d < 3; ++d) {
        sh[d][threadIdx.x] = lo[d];
        sh[3 + d][threadIdx.x] = hi[d];
    }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int d = 0; d < 3; ++d) {
                sh[d][threadIdx.x] = fmin(sh[d][threadIdx.x], sh[d][threadIdx.x + off]);
                sh[3 + d][threadIdx.x] = fmax(sh[3 + d][threadIdx.x], sh[3 + d][threadIdx.x + off]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) boxes[(int64_t)blockIdx.x * 6 + threadIdx.x] = sh[threadIdx.x][0];
    // quarter boxes: wave w reduces its own 64 triangles with shuffles
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fmin(lo[d], __shfl_xor(lo[d], off));
            hi[d] = fmax(hi[d], __shfl_xor(hi[d], off));
        }
    if ((threadIdx.x & 63) == 0) {
        double *sub = boxes + (int64_t)gridDim.x * 6 + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 6;
        for (int d = 0; d < 3; ++d) {
            sub[d] = lo[d];
            sub[3 + d] = hi[d];
        }
    }
}

struct Tri9 {
    double ax, ay, az, bx, by, bz, cx, cy, cz, orig;
};

// wave-wide bounding box of the valid lanes' points
__device__ __forceinline__ void wave_box(bool ok, double qx, double qy, double qz, double wb[6]) {
    double lo[3] = {ok ? qx : __builtin_huge_val(), ok ? qy : __builtin_huge_val(), ok ? qz : __builtin_huge_val()};
    double hi[3] = {ok ? qx : -__builtin_huge_val(), ok ? qy : -__builtin_huge_val(), ok ? qz : -__builtin_huge_val()};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fmin(lo[d], __shfl_xor(lo[d], off));
            hi[d] = fmax(hi[d], __shfl_xor(hi[d], off));
        }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        wb[d] = uniform_dd(lo[d]);
        wb[3 + d] = uniform_dd(hi[d]);
    }
}

__device__ __forceinline__ double box_box_gap2(const double a[6], const double *__restrict__ b) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double g = fmax(fmax(a[d] - b[3 + d], b[d] - a[3 + d]), 0.0);
        s = __builtin_fma(g, g, s);
    }
    return s;
}

// cp (SoA [3][nq]) / d2: closest point of the triangle soup to every query; exact ties go to the lowest ORIGINAL triangle.
// A workgroup of four waves serves 64 queries: every wave holds the same queries and scans ONE 64-triangle quarter of each
// staged tile, visited only if that quarter's box is not farther from a lane's query than the lane's best so far.  Sweep 0
// takes the tiles nearest to the queries' bounding box, then the four waves share their best distances (the bound only), and
// sweep 1 takes the remaining tiles under that bound.  Four times the parallelism of one wave per 64 queries, and the pruning
// works on quarters instead of tiles.
constexpr int kCpThreads = 256;

// H copies of every query per workgroup: 64 / H queries, each held by H lanes that take alternate triangles of the quarter.  Same
// arithmetic per (query, triangle) pair, a shorter scan per workgroup and more workgroups: the kernel is bound by its longest
// workgroups, not by the vector ALU.
// The exact closest-point test (~170 instructions) is not run where the box test passes: a step's 64 (query, triangle) pairs
// survive their box tests at a rate of 1.2 % (41k queries x 82k triangles), so running it for the whole wave whenever ANY pair
// survives wastes 98 % of the lanes (the first version of this kernel did; 232 us against 147 us).  Instead the
// survivors of the box tests are COMPACTED across the wave: every lane appends its surviving pair to a per-wave LDS queue (ballot
// + prefix count), and as soon as 64 pairs are queued every lane pops one and runs the exact test on ITS pair -- a different
// query and a different triangle in every lane.  The result goes to the owning query through LDS: an atomic minimum on the
// squared distance (non-negative doubles order like their bit patterns), then an atomic minimum on the original triangle number
// among the lanes that hold that minimum (the tie rule), then the winner stores its point.  All copies of a query prune against
// the shared best after every flush.  The queue is drained before the tile in LDS is replaced.
template <int H>
__global__ __launch_bounds__(kCpThreads) void surface_cp_queue_kernel(Cloud q, Cloud v, const int32_t *__restrict__ tri,
                                                                     const int32_t *__restrict__ tri_orig, int64_t T,
                                                                     const double *__restrict__ boxes, double *__restrict__ cp,
                                                                     double *__restrict__ d2out, int32_t *__restrict__ tri_out,
                                                                     const int32_t *warm_in, int32_t *pos_out /* may alias warm_in */,
                                                                     const double *__restrict__ tribox,
                                                                     const uint8_t *__restrict__ mask, const int32_t *__restrict__ nmask) {
    // mask / nmask (nullable): only the queries with mask[i] != 0 are answered -- what the triangle-grid search in front of this
    // launch could not certify (surface_cp_grid_kernel) -- and the whole launch is a no-op when *nmask == 0
    if (nmask && *nmask == 0) return;
    __shared__ double tbox[kTriTile][6];  // staged tile: the triangles' bounding boxes only (the exact test reads memory)
    constexpr int QPB = 64 / H;  // queries per workgroup
    __shared__ unsigned long long qbest[4][QPB];  // per wave and query: bits of the best squared distance so far
    __shared__ unsigned int qorig[4][QPB];        // ... its original triangle (lowest on ties)
    __shared__ double qpt[4][3][QPB];             // ... its point
    __shared__ int qpos[4][QPB];                  // ... its position in `tri` (the next call's warm start)
    __shared__ double sq[3][QPB];                 // the queries
    __shared__ unsigned int wqueue[4][128];  // (query slot << 26) | position of the triangle in `tri`
    __shared__ double sbound[4][QPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & (QPB - 1), half = lane / QPB;
    const int64_t i = (int64_t)blockIdx.x * QPB + ql;
    const bool ok = i < q.n && (!mask || mask[i] != 0);
    if (mask && !__syncthreads_or(ok)) return;  // nothing flagged in this workgroup
    const double qx = ok ? q.x[i] : 0.0, qy = ok ? q.y[i] : 0.0, qz = ok ? q.z[i] : 0.0;
    const unsigned long long kInfBits = 0x7FF0000000000000ull;
    // Warm start (nullable): the triangle (position in `tri`) that was closest to this query LAST time -- a template vertex moves
    // little between two ICP iterations.  Its exact distance is a valid candidate: the bound is tight before the first tile instead
    // of after the first 64 evaluated pairs of every wave.  Same minimum, same tie rule (an equally close triangle lies in a tile
    // whose gap does not exceed the bound, so it is still evaluated and wins on the lower original id).
    unsigned long long wbits = kInfBits;
    unsigned worig = 0xFFFFFFFFu;
    int wpos = -1;
    V3 wpt{qx, qy, qz};
    if (warm_in && ok) {
        const int32_t tg = warm_in[i];
        if (tg >= 0 && tg < T) {
            const int32_t va = tri[3 * (int64_t)tg], vb = tri[3 * (int64_t)tg + 1], vc = tri[3 * (int64_t)tg + 2];
            const V3 pq{qx, qy, qz};
            const V3 c = closest_on_triangle(pq, V3{v.x[va], v.y[va], v.z[va]}, V3{v.x[vb], v.y[vb], v.z[vb]}, V3{v.x[vc], v.y[vc], v.z[vc]});
            const V3 dd = sub(c, pq);
            const double dist = (dd.x * dd.x + dd.y * dd.y) + dd.z * dd.z;
            if (dist == dist) {  // NaN: cold start
                wbits = __builtin_bit_cast(unsigned long long, dist);
                worig = (unsigned)(tri_orig ? tri_orig[tg] : tg);
                wpos = tg;
                wpt = c;
            }
        }
    }
    if (half == 0) {
        qbest[wave][ql] = wbits;
        qorig[wave][ql] = worig;
        qpos[wave][ql] = wpos;
        qpt[wave][0][ql] = wpt.x, qpt[wave][1][ql] = wpt.y, qpt[wave][2][ql] = wpt.z;
        if (wave == 0) sq[0][ql] = qx, sq[1][ql] = qy, sq[2][ql] = qz;
    }
    __syncthreads();
    double best = __builtin_bit_cast(double, wbits), bound = __builtin_huge_val();
    int tail = 0;  // queued pairs (wave-uniform)
    double wb[6];
    wave_box(ok, qx, qy, qz, wb);
    const int nt = (int)((T + kTriTile - 1) / kTriTile);
    const double *qboxes = boxes + (int64_t)nt * 6;
    // pops up to 64 queued pairs, one per lane
    auto flush = [&](int count) {
        __builtin_amdgcn_wave_barrier();
        const bool mine = lane < count;
        const unsigned e = wqueue[wave][mine ? lane : 0];
        const int tq = (int)(e >> 26);
        const int64_t tg = (int64_t)(e & 0x3FFFFFFu);
        const V3 pp{sq[0][tq], sq[1][tq], sq[2][tq]};
        Tri9 tr;  // the queue outlives the staged tile: the triangle comes from memory (L2: it was staged a moment ago)
        {
            const int32_t va = tri[3 * tg], vb = tri[3 * tg + 1], vc = tri[3 * tg + 2];
            tr = Tri9{v.x[va], v.y[va], v.z[va], v.x[vb], v.y[vb], v.z[vb], v.x[vc], v.y[vc], v.z[vc],
                      (double)(tri_orig ? tri_orig[tg] : (int32_t)tg)};
        }
        const V3 c = closest_on_triangle(pp, V3{tr.ax, tr.ay, tr.az}, V3{tr.bx, tr.by, tr.bz}, V3{tr.cx, tr.cy, tr.cz});
        const V3 dd = sub(c, pp);
        const double dist = (dd.x * dd.x + dd.y * dd.y) + dd.z * dd.z;
        const unsigned long long db = __builtin_bit_cast(unsigned long long, dist);
        const unsigned long long old = qbest[wave][tq];
        if (mine && db <= old) atomicMin(&qbest[wave][tq], db);
        __builtin_amdgcn_wave_barrier();
        const unsigned long long now = qbest[wave][tq];
        const bool top = mine && db == now;
        if (top && now < old) qorig[wave][tq] = 0xFFFFFFFFu;  // a strictly better distance: the old triangle no longer competes
        __builtin_amdgcn_wave_barrier();
        if (top) atomicMin(&qorig[wave][tq], (unsigned)tr.orig);
        __builtin_amdgcn_wave_barrier();
        if (top && qorig[wave][tq] == (unsigned)tr.orig) {
            qpt[wave][0][tq] = c.x, qpt[wave][1][tq] = c.y, qpt[wave][2][tq] = c.z;
            qpos[wave][tq] = (int)tg;
        }
        __builtin_amdgcn_wave_barrier();
        best = __builtin_bit_cast(double, qbest[wave][ql]);  // every copy of the query prunes against the shared best
        // entries beyond `count` move to the front
        if (tail > count) {
            const unsigned rest = lane + count < tail ? wqueue[wave][lane + count] : 0u;
            __builtin_amdgcn_wave_barrier();
            if (lane + count < tail) wqueue[wave][lane] = rest;
        }
        tail -= count;
    };
    double gmin = __builtin_huge_val();
    for (int t = lane; t < nt; t += 64) gmin = fmin(gmin, box_box_gap2(wb, boxes + (int64_t)t * 6));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmin = fmin(gmin, __shfl_xor(gmin, off));
    gmin = uniform_dd(gmin);
    for (int phase = 0; phase < 2; ++phase) {
        double bmax = ok ? bound : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bmax = fmax(bmax, __shfl_xor(bmax, off));
        bmax = uniform_dd(bmax) * (1.0 + 1e-12);
        for (int tc = 0; tc < nt; tc += 64) {
            const int tl = tc + lane;
            const double g = tl < nt ? box_box_gap2(wb, boxes + (int64_t)tl * 6) : __builtin_huge_val();
            unsigned long long cand = __ballot(tl < nt && (phase == 0 ? !(g > gmin) : (g > gmin && !(g > bmax))));
            while (cand) {  // workgroup-uniform (same queries, same bound in every wave)
                const int t = tc + __builtin_ctzll(cand);
                cand &= cand - 1;
                const int64_t tb = (int64_t)t * kTriTile, q0 = tb + 64 * wave;
                const double pd = point_box_gap2(qx, qy, qz, qboxes + ((int64_t)t * 4 + wave) * 6);
                const bool need = ok && q0 < T && !(pd > fmin(best, bound) * (1.0 + 1e-12));
                const bool wave_needs = __any(need);
                if (!__syncthreads_or(wave_needs)) continue;
                {
                    const int64_t tt = tb + threadIdx.x;
                    if (tt < T) {
                        double *bb = tbox[threadIdx.x];
                        if (tribox) {  // precomputed by tri_tile_bbox_kernel: one contiguous read
                            const double *sb = tribox + 6 * tt;
                            bb[0] = sb[0], bb[1] = sb[1], bb[2] = sb[2], bb[3] = sb[3], bb[4] = sb[4], bb[5] = sb[5];
                        } else {
                            const int32_t a = tri[3 * tt], b = tri[3 * tt + 1], c = tri[3 * tt + 2];
                            const double ax = v.x[a], ay = v.y[a], az = v.z[a], bx = v.x[b], by = v.y[b], bz = v.z[b], cx = v.x[c],
                                         cy = v.y[c], cz = v.z[c];
                            bb[0] = fmin(fmin(ax, bx), cx), bb[1] = fmin(fmin(ay, by), cy), bb[2] = fmin(fmin(az, bz), cz);
                            bb[3] = fmax(fmax(ax, bx), cx), bb[4] = fmax(fmax(ay, by), cy), bb[5] = fmax(fmax(az, bz), cz);
                        }
                    }
                }
                __syncthreads();
                if (wave_needs) {
                    const int cnt = (int)min((int64_t)64, T - q0);
                    for (int j0 = 0; j0 < cnt; j0 += H) {
                        const int jj = j0 + half;
                        const bool live = jj < cnt;
                        const double gap = point_box_gap2(qx, qy, qz, tbox[64 * wave + (live ? jj : cnt - 1)]);
                        const bool pass = need && live && !(gap > fmin(best, bound) * (1.0 + 1e-12));
                        const unsigned long long m = __ballot(pass);
                        if (m) {
                            if (pass) wqueue[wave][tail + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = ((unsigned)ql << 26) | (unsigned)(q0 + jj);
                            tail += __builtin_popcountll(m);
                            if (tail >= 64) flush(64);
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (tail > 0) flush(tail);  // drain: the sweep's results feed the bound / the answer
        if (phase == 0) {  // share the distance bound of sweep 0 between the waves
            if (half == 0) sbound[wave][ql] = __builtin_bit_cast(double, qbest[wave][ql]);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) bound = fmin(bound, sbound[k][ql]);
            __syncthreads();
        }
    }
    __syncthreads();
    if (wave == 0 && half == 0 && ok) {  // combine the four waves: smallest distance, ties -> lowest original triangle
        int w = 0;
        for (int k = 1; k < 4; ++k)
            if (qbest[k][ql] < qbest[w][ql] || (qbest[k][ql] == qbest[w][ql] && qorig[k][ql] < qorig[w][ql])) w = k;
        cp[i] = qpt[w][0][ql];
        cp[q.n + i] = qpt[w][1][ql];
        cp[2 * q.n + i] = qpt[w][2][ql];
        d2out[i] = __builtin_bit_cast(double, qbest[w][ql]);
        if (tri_out) tri_out[i] = (int32_t)qorig[w][ql];
        if (pos_out) pos_out[i] = qpos[w][ql];
    }
}

// ---- closest surface point over a uniform grid of the (fixed) target triangles (round 4) -------------------------------------------
// The target mesh of a registration does not move, so gingr_fitter_set_meshes bins its triangles once: cell c lists (positions in
// `tri` of) the triangles whose bounding box overlaps it.  A query starts from the triangle that was closest to it in the previous
// scan (warm start; a template vertex moves little between two iterations): its exact distance r bounds the answer, and every
// triangle that holds a point within r of the query has a bounding box that overlaps a cell the ball of radius r overlaps -- the
// cell of that point -- so scanning the triangle lists of the cells [cell(q - r), cell(q + r)] (a slack of 1e-9 r covers the
// rounding of the cell index; host and device evaluate the same floor((x - lo) * inv_h), which is monotone in x) finds the exact
// minimum and, by the (distance, original triangle id) order, the same winner on ties as the tile scan.  Same point-triangle routine,
// same separately rounded distance: bit-identical closest points.  A query whose ball covers more than kTriGridMaxCells cells (far
// from the surface, the early iterations), a cold start, or a non-finite query is FLAGGED and answered by the masked tile scan
// (surface_cp_queue_kernel) that follows.  kLanes lanes per query take the cells of the block in turn; a triangle listed in several
// cells is evaluated more than once with the same result.
#ifndef GINGR_TRI_GRID_LANES
#define GINGR_TRI_GRID_LANES 16
#endif
#ifndef GINGR_TRI_GRID_CELLS
#define GINGR_TRI_GRID_CELLS 64
#endif
constexpr int kTriGridMaxCells = GINGR_TRI_GRID_CELLS;
constexpr int kTriGridCand = 96;  // candidates (entries that pass the box test and the home-cell rule) kept per query; more: the tile scan
constexpr int kTriRec = 10;       // doubles per grid entry behind its box: corners A, B, C, {position | original index << 32}
constexpr int kTriGridMaxSpan = 3;  // a listed triangle's box spans at most this many cell steps per axis (wider ones: the short list)

template <int kLanes>
__global__ __launch_bounds__(256) void surface_cp_grid_kernel(Cloud q, Cloud v, const int32_t *__restrict__ tri,
                                                             const int32_t *__restrict__ tri_orig, int64_t T, TriGridDev g,
                                                             double *__restrict__ cp, double *__restrict__ d2out,
                                                             int32_t *__restrict__ tri_out,
                                                             int32_t *warm /* in: last closest triangle, out: this one's */,
                                                             uint8_t *__restrict__ flag, int32_t *__restrict__ nflag,
                                                             int32_t *__restrict__ nflag_next) {
    static_assert(kLanes == 32 || kLanes == 16 || kLanes == 8, "sub-wave ballots below");
    constexpr int QPB = 256 / kLanes;
    __shared__ int32_t cell_s[QPB][kTriGridMaxCells], cell_off[QPB][kTriGridMaxCells + 1];
    __shared__ int32_t cand[QPB][kTriGridCand];
    if (blockIdx.x == 0 && threadIdx.x == 0) *nflag_next = 0;  // the counter of the NEXT search (nobody reads it during this one)
    const int ql = threadIdx.x % kLanes, qi = threadIdx.x / kLanes, wslot = (threadIdx.x & 63) / kLanes;  // wslot: the query's slot in its wave
    const int64_t i = (int64_t)blockIdx.x * QPB + qi;
    const bool ok = i < q.n;
    const double qx = ok ? q.x[i] : 0.0, qy = ok ? q.y[i] : 0.0, qz = ok ? q.z[i] : 0.0;
    const V3 p{qx, qy, qz};
    double best = __builtin_huge_val();
    unsigned bo = 0xFFFFFFFFu;
    int bpos = -1;
    V3 bp{qx, qy, qz};
    auto consider = [&](V3 A, V3 B, V3 C, int pos, unsigned o) {
        const V3 c = closest_on_triangle(p, A, B, C);
        const V3 dd = sub(c, p);
        const double dist = (dd.x * dd.x + dd.y * dd.y) + dd.z * dd.z;
        if (dist < best || (dist == best && o < bo)) {
            best = dist;
            bo = o;
            bpos = pos;
            bp = c;
        }
    };
    bool fl = ok;  // flagged unless the block below certifies the answer
    if (ok) {  // (uniform over the kLanes lanes of a query)
        const int32_t tg = warm[i];
        if (tg >= 0 && tg < T) {  // every lane of the query evaluates the warm triangle: the same bound everywhere
            const int32_t va = tri[3 * (int64_t)tg], vb = tri[3 * (int64_t)tg + 1], vc = tri[3 * (int64_t)tg + 2];
            consider(V3{v.x[va], v.y[va], v.z[va]}, V3{v.x[vb], v.y[vb], v.z[vb]}, V3{v.x[vc], v.y[vc], v.z[vc]}, (int)tg,
                     (unsigned)(tri_orig ? tri_orig[tg] : tg));
        }
        if (best < __builtin_huge_val()) {  // (NaN / no warm triangle: stays flagged)
            const double bound = best;  // the ball every candidate is tested against (the running best only shrinks inside it)
            const double r = sqrt(bound) * (1.0 + 1e-9) + 1e-300;
            const double f0[3] = {(qx - r - g.lo[0]) * g.inv_h, (qy - r - g.lo[1]) * g.inv_h, (qz - r - g.lo[2]) * g.inv_h};
            const double f1[3] = {(qx + r - g.lo[0]) * g.inv_h, (qy + r - g.lo[1]) * g.inv_h, (qz + r - g.lo[2]) * g.inv_h};
            bool fin = true;
            int c0[3], c1[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                fin = fin && fabs(f0[d]) < 1e15 && fabs(f1[d]) < 1e15;
                const double a = floor(f0[d]), b = floor(f1[d]);
                c0[d] = a >= (double)(g.g[d] - 1) ? g.g[d] - 1 : (a > 0.0 ? (int)a : 0);
                c1[d] = b >= (double)(g.g[d] - 1) ? g.g[d] - 1 : (b > 0.0 ? (int)b : 0);
            }
            // every listed triangle sits in the cell of its box's lower corner: the cells [c0 - span, c1] hold all that reach the ball;
            // along x they are ONE contiguous run of entries per (y, z) row
            const int x0 = c0[0] > g.span[0] ? c0[0] - g.span[0] : 0, y0 = c0[1] > g.span[1] ? c0[1] - g.span[1] : 0,
                      z0 = c0[2] > g.span[2] ? c0[2] - g.span[2] : 0;
            const int ny = c1[1] - y0 + 1, nz = c1[2] - z0 + 1;
            if (fin && ny * nz + 1 <= kTriGridMaxCells) {
                const int nrow = ny * nz, nrun = nrow + (g.n_big > 0 ? 1 : 0);
                // (1) the entry run of every row, one row per lane; the short list of wide triangles is one more run
                for (int k = ql; k < nrun; k += kLanes) {
                    int32_t s0, n0;
                    if (k < nrow) {
                        const int rz = k / ny, ry = k - rz * ny;
                        const int64_t rowbase = ((int64_t)(z0 + rz) * g.g[1] + (y0 + ry)) * g.g[0];
                        s0 = g.cell_start[rowbase + x0];
                        n0 = g.cell_start[rowbase + c1[0] + 1] - s0;
                    } else {
                        s0 = g.n_listed;
                        n0 = g.n_big;
                    }
                    cell_s[qi][k] = s0;
                    cell_off[qi][k + 1] = n0;
                }
                __threadfence_block();
                if (ql == 0) {  // exclusive prefix over at most 64 counts
                    int32_t off = 0;
                    for (int k = 0; k < nrun; ++k) {
                        const int32_t n = cell_off[qi][k + 1];
                        cell_off[qi][k] = off;
                        off += n;
                    }
                    cell_off[qi][nrun] = off;
                }
                __threadfence_block();
                const int32_t total = cell_off[qi][nrun];
                // (2) all entries of the runs, flattened over the lanes: the box against the ball; survivors go to the query's list
                int ncand = 0, k = 0;
                bool overflow = false;
                for (int32_t base = 0; base < total; base += kLanes) {
                    const int32_t idx = base + ql;
                    bool pass = false;
                    int32_t e = 0;
                    if (idx < total) {
                        while (cell_off[qi][k + 1] <= idx) ++k;
                        e = cell_s[qi][k] + (idx - cell_off[qi][k]);
                        pass = !(point_box_gap2(qx, qy, qz, g.boxes + (int64_t)e * 6) > bound);  // (equal: a possible tie, evaluated)
                    }
                    const unsigned hm = (unsigned)(__ballot(pass) >> (kLanes * wslot)) & (kLanes == 32 ? 0xffffffffu : ((1u << (kLanes & 31)) - 1u));
                    if (pass) {
                        const int slot = ncand + __builtin_popcount(hm & ((1u << ql) - 1u));
                        if (slot < kTriGridCand) cand[qi][slot] = e;
                    }
                    ncand += __builtin_popcount(hm);
                }
                if (ncand > kTriGridCand) overflow = true, ncand = 0;
                __threadfence_block();
                // (3) the exact routine on the survivors, one per lane
                for (int c = ql; c < ncand; c += kLanes) {
                    const double *rc = g.recs + (int64_t)cand[qi][c] * kTriRec;
                    const long long meta = __builtin_bit_cast(long long, rc[9]);
                    consider(V3{rc[0], rc[1], rc[2]}, V3{rc[3], rc[4], rc[5]}, V3{rc[6], rc[7], rc[8]}, (int)(meta & 0xffffffffLL),
                             (unsigned)((unsigned long long)meta >> 32));
                }
                fl = overflow;
            }
        }
    }
    // combine the lanes of the query: smallest distance, then lowest original triangle
#pragma unroll
    for (int off = kLanes / 2; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const unsigned oo = (unsigned)__shfl_xor((int)bo, off);
        const int op = __shfl_xor(bpos, off);
        const double ox = __shfl_xor(bp.x, off), oy = __shfl_xor(bp.y, off), oz = __shfl_xor(bp.z, off);
        if (ob < best || (ob == best && oo < bo)) {
            best = ob;
            bo = oo;
            bpos = op;
            bp = V3{ox, oy, oz};
        }
    }
    if (ok && ql == 0) {
        flag[i] = fl ? 1 : 0;
        if (!fl) {
            cp[i] = bp.x;
            cp[q.n + i] = bp.y;
            cp[2 * q.n + i] = bp.z;
            d2out[i] = best;
            if (tri_out) tri_out[i] = (int32_t)bo;
            warm[i] = bpos;
        }
    }
    // flagged queries of the workgroup: one atomic
    const unsigned long long m = __ballot(ok && ql == 0 && fl);
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt, __builtin_popcountll(m));
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(nflag, cnt);
}

// Barycentric weights (of A, B, C) of the closest point of triangle (A, B, C) to p: the region logic of closest_on_triangle with
// the weights spelled out -- vertex regions (1,0,0), edge regions (1-q, q, 0), interior (1 - v - w, v, w).
__device__ __forceinline__ V3 closest_barycentric(V3 p, V3 A, V3 B, V3 C) {
    const V3 ab = sub(B, A), ac = sub(C, A), ap = sub(p, A), bp = sub(p, B), cp = sub(p, C);
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap), d3 = dot3(ab, bp), d4 = dot3(ac, bp), d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d1 <= 0.0 && d2 <= 0.0) return V3{1.0, 0.0, 0.0};
    if (d3 >= 0.0 && d4 <= d3) return V3{0.0, 1.0, 0.0};
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double q = d1 / (d1 - d3);
        return V3{1.0 - q, q, 0.0};
    }
    if (d6 >= 0.0 && d5 <= d6) return V3{0.0, 0.0, 1.0};
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
        const double q = d2 / (d2 - d6);
        return V3{1.0 - q, 0.0, q};
    }
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double q = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        return V3{0.0, 1.0 - q, q};
    }
    const double q = 1.0 / ((va + vb) + vc);
    const double v = vb * q, w = vc * q;
    return V3{(1.0 - v) - w, v, w};
}

// bary[3 i + k]: weight of corner k of triangle tri_id[i] (original numbering; corners given in `tri_corners` [3 T] as positions in
// the cloud v) for query i
__global__ __launch_bounds__(256) void barycentric_kernel(Cloud q, Cloud v, const int32_t *__restrict__ tri_by_orig,
                                                          const int32_t *__restrict__ tri_id, double *__restrict__ bary) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= q.n) return;
    const int64_t t = tri_id[i];
    const int32_t a = tri_by_orig[3 * t], b = tri_by_orig[3 * t + 1], c = tri_by_orig[3 * t + 2];
    const V3 w = closest_barycentric(V3{q.x[i], q.y[i], q.z[i]}, V3{v.x[a], v.y[a], v.z[a]}, V3{v.x[b], v.y[b], v.z[b]},
                                     V3{v.x[c], v.y[c], v.z[c]});
    bary[3 * i] = w.x;
    bary[3 * i + 1] = w.y;
    bary[3 * i + 2] = w.z;
}

// flag[i] = 1 when the line through fit_i along fit_i - cp_i meets the mesh (v, tri) in a point != fit_i that is closer to fit_i
// than cp_i is (ClosestPointRegistrator.scala:62-72).  Lanes with skip[i] != 0 do no work (their weight is already 0).
// Same structure as surface_cp_queue_kernel: the staged tile holds bounding boxes only, (point,
// triangle) pairs whose box reaches into the ball of radius |v| around the point are compacted across the wave, and every lane runs
// the line / triangle test on its own pair, reading the triangle from memory.  A hit is OR-ed into the point's flag in LDS; points
// that are already hit stop producing pairs ("some triangle holds a closer intersection" does not depend on the order).
template <int H>
__global__ __launch_bounds__(kCpThreads) void self_intersect_queue_kernel(Cloud fit, const double *__restrict__ cp, Cloud v,
                                                                         const int32_t *__restrict__ tri, int64_t T,
                                                                         const double *__restrict__ boxes,
                                                                         const int32_t *__restrict__ skip,
                                                                         int32_t *flag /* may alias F.found (the along-normal flavour) */, const double *__restrict__ tribox,
                                                                         const uint8_t *__restrict__ only, const int32_t *__restrict__ nonly,
                                                                         SelfIntersectFuse F) {
    if (nonly && *nonly == 0) return;  // masked launch (what the grid kernel could not certify): nothing left over
    __shared__ double tbox[kTriTile][6];
    constexpr int QPB = 64 / H;
    __shared__ int qhit[QPB];
    __shared__ double sp[3][QPB], sdir[3][QPB], snorm[QPB];
    __shared__ unsigned int wqueue[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & (QPB - 1), half = lane / QPB;
    const int64_t i = (int64_t)blockIdx.x * QPB + ql;
    const bool mine = i < fit.n && (!only || only[i] != 0);
    if (only && !__syncthreads_or(mine)) return;  // none of this workgroup's queries was left over
    bool rejected = false;
    if (F.nn_vertex) {  // the first two rejection tests (surface_prereject_kernel: boundary vertex, opposite normals), made here
        if (mine) {
            const int32_t j = F.nn_vertex[i];
            if (F.found && !F.found[i])
                rejected = true;
            else if (j < 0)
                rejected = true;
            else if (F.boundary[j])
                rejected = true;
            else
                rejected = (F.q_vn[i] * F.t_vn[j] + F.q_vn[fit.n + i] * F.t_vn[F.Nt + j]) + F.q_vn[2 * fit.n + i] * F.t_vn[2 * F.Nt + j] < 0.0;
            if (wave == 0 && half == 0) F.pre_out[i] = rejected ? 1 : 0;
        }
    } else {
        rejected = skip && mine && skip[i];
    }
    const bool ok = mine && !rejected;
    const int64_t ic = i < fit.n ? i : 0;
    const V3 p{fit.x[ic], fit.y[ic], fit.z[ic]};
    const V3 dir = sub(p, V3{cp[ic], cp[fit.n + ic], cp[2 * fit.n + ic]});
    const double vv = dot3(dir, dir);
    const double vnorm = sqrt(vv);
    if (wave == 0 && half == 0) {
        qhit[ql] = 0;
        sp[0][ql] = p.x, sp[1][ql] = p.y, sp[2][ql] = p.z;
        sdir[0][ql] = dir.x, sdir[1][ql] = dir.y, sdir[2][ql] = dir.z;
        snorm[ql] = vnorm;
    }
    __syncthreads();
    int hit = 0, tail = 0;
    auto flush = [&](int count) {
        __builtin_amdgcn_wave_barrier();
        const bool mine = lane < count;
        const unsigned e = wqueue[wave][mine ? lane : 0];
        const int tq = (int)(e >> 26);
        const int64_t tg = (int64_t)(e & 0x3FFFFFFu);
        const V3 pp{sp[0][tq], sp[1][tq], sp[2][tq]}, dd0{sdir[0][tq], sdir[1][tq], sdir[2][tq]};
        const int32_t va = tri[3 * tg], vb = tri[3 * tg + 1], vc = tri[3 * tg + 2];
        const V3 A{v.x[va], v.y[va], v.z[va]};
        const V3 e1 = sub(V3{v.x[vb], v.y[vb], v.z[vb]}, A), e2 = sub(V3{v.x[vc], v.y[vc], v.z[vc]}, A);
        const V3 pv = cross3(dd0, e2);
        const double det = dot3(e1, pv);
        const double inv = 1.0 / det;
        const V3 tv = sub(pp, A);
        const double u = dot3(tv, pv) * inv;
        const V3 qv = cross3(tv, e1);
        const double w = dot3(qv, dd0) * inv;
        const double tt = dot3(e2, qv) * inv;
        if (mine && det != 0.0 && u >= 0.0 && u <= 1.0 && w >= 0.0 && u + w <= 1.0) {
            const V3 ip{pp.x + tt * dd0.x, pp.y + tt * dd0.y, pp.z + tt * dd0.z};
            if (ip.x != pp.x || ip.y != pp.y || ip.z != pp.z) {
                const V3 dd = sub(ip, pp);
                if (sqrt((dd.x * dd.x + dd.y * dd.y) + dd.z * dd.z) < snorm[tq]) qhit[tq] = 1;  // same value from every writer
            }
        }
        __builtin_amdgcn_wave_barrier();
        hit = qhit[ql];
        if (tail > count) {
            const unsigned rest = lane + count < tail ? wqueue[wave][lane + count] : 0u;
            __builtin_amdgcn_wave_barrier();
            if (lane + count < tail) wqueue[wave][lane] = rest;
        }
        tail -= count;
    };
    const int nt = (int)((T + kTriTile - 1) / kTriTile);
    const double *qboxes = boxes + (int64_t)nt * 6;
    double wb[6];
    wave_box(ok, p.x, p.y, p.z, wb);
    double vmax = ok ? vv : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, off));
    vmax = uniform_dd(vmax) * (1.0 + 1e-12);
    for (int tc = 0; tc < nt; tc += 64) {
        const int tl = tc + lane;
        unsigned long long cand = __ballot(tl < nt && !(box_box_gap2(wb, boxes + (int64_t)tl * 6) > vmax));
        while (cand) {  // workgroup-uniform
            const int t = tc + __builtin_ctzll(cand);
            cand &= cand - 1;
            const int64_t tb = (int64_t)t * kTriTile, q0 = tb + 64 * wave;
            const double pd = point_box_gap2(p.x, p.y, p.z, qboxes + ((int64_t)t * 4 + wave) * 6);
            const bool need = ok && !hit && q0 < T && !(pd > vv * (1.0 + 1e-12));
            const bool wave_needs = __any(need);
            // Round 5: a wave stages only ITS quarter of the tile (64 boxes into its own slice of tbox) and only when one of its queries
            // reaches into that quarter's box: no workgroup barrier in the tile loop -- the four waves walk the candidate list (identical
            // in all of them: same queries, same tile boxes) decoupled -- and a quarter nobody needs is never read.  (Until round 4 all
            // 256 boxes of a tile were staged, between two barriers, as soon as ANY wave needed one quarter: 67 us at 41k x 82k.)
            if (wave_needs) {
                __builtin_amdgcn_wave_barrier();  // (the previous quarter's reads of the slice are issued before it is rewritten)
                const int64_t tt = q0 + lane;
                if (tt < T) {
                    double *bb = tbox[64 * wave + lane];
                    if (tribox) {  // precomputed by tri_tile_bbox_kernel
                        const double *sb = tribox + 6 * tt;
                        bb[0] = sb[0], bb[1] = sb[1], bb[2] = sb[2], bb[3] = sb[3], bb[4] = sb[4], bb[5] = sb[5];
                    } else {
                        const int32_t a = tri[3 * tt], b = tri[3 * tt + 1], c = tri[3 * tt + 2];
                        const double ax = v.x[a], ay = v.y[a], az = v.z[a], bx = v.x[b], by = v.y[b], bz = v.z[b], cx = v.x[c], cy = v.y[c],
                                     cz = v.z[c];
                        bb[0] = fmin(fmin(ax, bx), cx), bb[1] = fmin(fmin(ay, by), cy), bb[2] = fmin(fmin(az, bz), cz);
                        bb[3] = fmax(fmax(ax, bx), cx), bb[4] = fmax(fmax(ay, by), cy), bb[5] = fmax(fmax(az, bz), cz);
                    }
                }
                __builtin_amdgcn_wave_barrier();  // LDS serves one wave's accesses in order: the reads below see the writes above
                const int cnt = (int)min((int64_t)64, T - q0);
                for (int jb = 0; jb < cnt; jb += H) {
                    const int jj = jb + half;
                    const bool live = jj < cnt;
                    const double gap = point_box_gap2(p.x, p.y, p.z, tbox[64 * wave + (live ? jj : cnt - 1)]);
                    const bool pass = need && live && !hit && !(gap > vv * (1.0 + 1e-9));
                    const unsigned long long m = __ballot(pass);
                    if (m) {
                        if (pass) wqueue[wave][tail + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = ((unsigned)ql << 26) | (unsigned)(q0 + jj);
                        tail += __builtin_popcountll(m);
                        if (tail >= 64) flush(64);
                    }
                }
            }
        }
    }
    if (tail > 0) flush(tail);
    __syncthreads();
    if (wave == 0 && half == 0 && mine) {
        flag[i] = qhit[ql];
        if (F.w01) {  // surface_weight_kernel: w in {0, 1}, weight_in = w / sigma2
            const double w = (rejected || qhit[ql]) ? 0.0 : 1.0;
            F.w01[i] = w;
            F.weight_in[i] = w / F.sigma2[0];
        }
    }
}

// ClosestPointAlongNormalTriangleMesh3D (ClosestPointRegistrator.scala:102-131): for every fit vertex the intersection of the
// line {p + t n} (n = its vertex normal, both directions) with the mesh (v, tri) that is closest to p and != p; found[i] = 0 and
// cp = p when there is none.  A 256-triangle tile is visited only if some lane's line passes through its (slightly inflated) box and
// the box is not farther from p than the lane's current hit; then the same test on its four 64-triangle quarters (boxes behind the
// tile boxes: tri_tile_bbox_kernel), and only a quarter some lane needs is staged.  Exact ties go to the lowest ORIGINAL triangle.
// Round 6 (1 067 -> see DESIGN.md at 41k x 82k, where it was 85 % of an iteration of this ICP flavour): the slab test multiplies by
// the line's reciprocal direction (six float64 divisions per box before), quarters instead of whole tiles, and a triangle whose
// barycentric numerators are clearly outside [0, det] is dropped before the division of the Moeller-Trumbore test -- the survivors go
// through the same expressions as before.
struct LineSlab {
    double p[3], inv[3];
    bool par[3];  // direction component exactly zero
    __device__ __forceinline__ bool hits(const double *bx) const {
        double tmin = -__builtin_huge_val(), tmax = __builtin_huge_val();
        bool miss = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double eps = 1e-9 * (fabs(bx[d]) + fabs(bx[3 + d]) + fabs(p[d]) + 1e-300);
            const double lo = bx[d] - eps, hi = bx[3 + d] + eps;
            if (par[d]) {
                if (p[d] < lo || p[d] > hi) miss = true;
            } else {
                const double t1 = (lo - p[d]) * inv[d], t2 = (hi - p[d]) * inv[d];
                tmin = fmax(tmin, fmin(t1, t2));
                tmax = fmin(tmax, fmax(t1, t2));
            }
        }
        return !(miss || tmin > tmax);
    }
};

constexpr int kLineGroup = 16;       // tiles per group box
#ifndef GINGR_LINE_COPIES
#define GINGR_LINE_COPIES 4
#endif
constexpr int kLineCopies = GINGR_LINE_COPIES;  // lanes per query: they take alternate triangles of a staged quarter and alternate group boxes
constexpr int kLineQueries = kSurfThreads / kLineCopies;

// boxes of the groups of 16 tiles (the triangle order is a k-d order: aligned runs are compact), behind the tile and quarter boxes
__global__ __launch_bounds__(64) void line_group_boxes_kernel(double *__restrict__ boxes, int nt) {
    const int idx = blockIdx.x * 64 + threadIdx.x, g = idx / 6, d = idx - 6 * g;
    if (g >= (nt + kLineGroup - 1) / kLineGroup) return;
    double vals[kLineGroup];
#pragma unroll
    for (int u = 0; u < kLineGroup; ++u) {
        const int t = min(g * kLineGroup + u, nt - 1);
        vals[u] = boxes[(int64_t)t * 6 + d];
    }
    double r = vals[0];
#pragma unroll
    for (int u = 1; u < kLineGroup; ++u) r = d < 3 ? fmin(r, vals[u]) : fmax(r, vals[u]);
    boxes[(int64_t)nt * 30 + (int64_t)g * 6 + d] = r;
}

// Four lanes per query (16 queries a wave): the union of the quarters the lines of a wave pierce is smaller, a staged quarter costs 16
// steps instead of 64, and there are four times the waves to hide each other's staging latency (one wave per SIMD otherwise).
__global__ __launch_bounds__(kSurfThreads) void line_nearest_kernel(Cloud fit, const double *__restrict__ dirs, Cloud v,
                                                                   const int32_t *__restrict__ tri,
                                                                   const int32_t *__restrict__ tri_orig, int64_t T,
                                                                   const double *__restrict__ boxes, double *__restrict__ cp,
                                                                   int32_t *__restrict__ found) {
    __shared__ Tri9 quarter[64];
    const int lane = threadIdx.x, copy = lane & (kLineCopies - 1);
    const int64_t i = (int64_t)blockIdx.x * kLineQueries + lane / kLineCopies;
    const bool ok = i < fit.n;
    const int64_t ic = ok ? i : 0;
    const V3 p{fit.x[ic], fit.y[ic], fit.z[ic]};
    const V3 dir{dirs[ic], dirs[fit.n + ic], dirs[2 * fit.n + ic]};
    LineSlab line;
    line.p[0] = p.x, line.p[1] = p.y, line.p[2] = p.z;
    {
        const double da[3] = {dir.x, dir.y, dir.z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            line.par[d] = da[d] == 0.0;
            line.inv[d] = line.par[d] ? 0.0 : 1.0 / da[d];
        }
    }
    const int nt = (int)((T + kTriTile - 1) / kTriTile);
    const double *qboxes = boxes + (int64_t)nt * 6;
    const double *gboxes = boxes + (int64_t)nt * 30;  // line_group_boxes_kernel
    // Sweeps over shells round the workgroup's own points, the radius doubling: tiles come roughly nearest first, so a line's hit in
    // one shell culls (by distance) what it pierces in the later ones -- the far side of a closed mesh, the fat boxes of slanted
    // patches.  A tile belongs to the shell its box's gap from the points' centre falls into; the sweeps end when every line has a hit
    // nearer than the shell reached, or the farthest box corner is inside it.
    const int ngroups = (nt + kLineGroup - 1) / kLineGroup;
    double wb[6];
    wave_box(ok, p.x, p.y, p.z, wb);
    const double cx = 0.5 * (wb[0] + wb[3]), cy = 0.5 * (wb[1] + wb[4]), cz = 0.5 * (wb[2] + wb[5]);
    const double ext = sqrt((wb[3] - wb[0]) * (wb[3] - wb[0]) + (wb[4] - wb[1]) * (wb[4] - wb[1]) + (wb[5] - wb[2]) * (wb[5] - wb[2]));
    double gmin2 = __builtin_huge_val(), gfar2 = 0.0;  // nearest gap / farthest corner of the group boxes from the centre
    for (int g = 0; g < ngroups; ++g) {
        const double *gb = gboxes + (int64_t)g * 6;
        gmin2 = fmin(gmin2, point_box_gap2(cx, cy, cz, gb));
        const double fx = fmax(fabs(cx - gb[0]), fabs(cx - gb[3])), fy = fmax(fabs(cy - gb[1]), fabs(cy - gb[4])),
                     fz = fmax(fabs(cz - gb[2]), fabs(cz - gb[5]));
        gfar2 = fmax(gfar2, fx * fx + fy * fy + fz * fz);
    }
    double radius = fmax(fmax(3.0 * ext, 1.5 * sqrt(gmin2)), sqrt(gfar2) * (1.0 / 64.0));
    double inner2 = -1.0;  // tiles with inner2 < gap2 <= radius^2 belong to the sweep
    double best = __builtin_huge_val(), bo = __builtin_huge_val();  // distance |p - ip| and the original triangle that holds it
    V3 bp = p;
    double bound = __builtin_huge_val();  // the smallest `best` of the query's four lanes (culling only)
    for (;;) {
        const double outer2 = radius * radius;
        for (int g0 = 0; g0 < nt; g0 += kLineGroup) {
            const int g = g0 / kLineGroup;
            {
                const double *gb = gboxes + (int64_t)g * 6;
                if (point_box_gap2(cx, cy, cz, gb) > outer2) continue;  // (uniform) the whole group lies in a later shell
                // one of the query's four lanes tests the group (the wave only needs the union)
                const bool need_group = ok && (g & (kLineCopies - 1)) == copy && line.hits(gb) &&
                                        !(point_box_gap2(p.x, p.y, p.z, gb) > bound * bound * (1.0 + 1e-12));
                if (!__any(need_group)) continue;
            }
            // the query's four lanes share the box tests of the group's 16 tiles (four each) and of a tile's four quarters (one each);
            // two shuffles give every lane the query's whole mask
            unsigned tmask = 0;
#pragma unroll
            for (int u = 0; u < (kLineGroup + kLineCopies - 1) / kLineCopies; ++u) {
                const int k = kLineCopies * u + copy, t = g0 + k;
                if (k < kLineGroup && t < nt) {
                    const double *bx = boxes + (int64_t)t * 6;
                    const double cg2 = point_box_gap2(cx, cy, cz, bx);
                    if (cg2 > inner2 && cg2 <= outer2 && ok && line.hits(bx) &&
                        !(point_box_gap2(p.x, p.y, p.z, bx) > bound * bound * (1.0 + 1e-12)))
                        tmask |= 1u << k;
                }
            }
#pragma unroll
            for (int off = 1; off < kLineCopies; off <<= 1) tmask |= __shfl_xor(tmask, off);
            for (int k = 0; k < kLineGroup && g0 + k < nt; ++k) {
                if (!__any((tmask >> k) & 1u)) continue;
                const int t = g0 + k;
                const bool need_tile = (tmask >> k) & 1u;
                unsigned qmask = 0;
#pragma unroll
                for (int u = 0; u < (kTriTile / 64 + kLineCopies - 1) / kLineCopies; ++u) {
                    const int q = kLineCopies * u + copy;
                    if (q < kTriTile / 64) {
                        const int64_t q0 = (int64_t)t * kTriTile + 64 * q;
                        const double *qb = qboxes + ((int64_t)t * 4 + q) * 6;
                        if (q0 < T && need_tile && line.hits(qb) && !(point_box_gap2(p.x, p.y, p.z, qb) > bound * bound * (1.0 + 1e-12)))
                            qmask |= 1u << q;
                    }
                }
#pragma unroll
                for (int off = 1; off < kLineCopies; off <<= 1) qmask |= __shfl_xor(qmask, off);
                for (int q = 0; q < kTriTile / 64; ++q) {
                    if (!__any((qmask >> q) & 1u)) continue;
                    const int64_t q0 = (int64_t)t * kTriTile + 64 * q;
                    // (the bound may have dropped since the mask was made)
                    const bool need = ((qmask >> q) & 1u) &&
                                      !(point_box_gap2(p.x, p.y, p.z, qboxes + ((int64_t)t * 4 + q) * 6) > bound * bound * (1.0 + 1e-12));
                    __syncthreads();
                    if (q0 + lane < T) {
                        const int64_t tq = q0 + lane;
                        const int32_t a = tri[3 * tq], b = tri[3 * tq + 1], c = tri[3 * tq + 2];
                        quarter[lane] = Tri9{v.x[a], v.y[a], v.z[a], v.x[b], v.y[b], v.z[b], v.x[c], v.y[c], v.z[c],
                                             (double)(tri_orig ? tri_orig[tq] : (int32_t)tq)};
                    }
                    __syncthreads();
                    const int cnt = (int)min((int64_t)64, T - q0);
                    if (need)
                        for (int jj = copy; jj < cnt; jj += kLineCopies) {
                            const Tri9 tr = quarter[jj];
                            const V3 A{tr.ax, tr.ay, tr.az};
                            const V3 e1 = sub(V3{tr.bx, tr.by, tr.bz}, A), e2 = sub(V3{tr.cx, tr.cy, tr.cz}, A);
                            const V3 pv = cross3(dir, e2);
                            const double det = dot3(e1, pv);
                            const V3 tv = sub(p, A);
                            const double nu = dot3(tv, pv);
                            const double ad = fabs(det), su = det > 0.0 ? nu : -nu;
                            if (su < -1e-9 * ad || su > ad * (1.0 + 1e-9)) continue;  // u clearly outside [0, 1]
                            const V3 qv = cross3(tv, e1);
                            const double nw = dot3(qv, dir);
                            const double sw = det > 0.0 ? nw : -nw;
                            if (sw < -1e-9 * ad || su + sw > ad * (1.0 + 2e-9)) continue;  // w < 0 or u + w > 1, clearly
                            const double inv = 1.0 / det;
                            const double u = nu * inv;
                            const double w = nw * inv;
                            const double tt = dot3(e2, qv) * inv;
                            if (det != 0.0 && u >= 0.0 && u <= 1.0 && w >= 0.0 && u + w <= 1.0) {
                                const V3 ip{p.x + tt * dir.x, p.y + tt * dir.y, p.z + tt * dir.z};
                                if (ip.x != p.x || ip.y != p.y || ip.z != p.z) {
                                    const V3 dd = sub(ip, p);
                                    const double dist = sqrt((dd.x * dd.x + dd.y * dd.y) + dd.z * dd.z);
                                    if (dist < best || (dist == best && tr.orig < bo)) {
                                        best = dist;
                                        bo = tr.orig;
                                        bp = ip;
                                    }
                                }
                            }
                        }
                    bound = best;
#pragma unroll
                    for (int off = 1; off < kLineCopies; off <<= 1) bound = fmin(bound, __shfl_xor(bound, off));
                }
            }
        }
        if (outer2 >= gfar2) break;                                    // every tile has been in a shell
        if (__all(!ok || bound <= radius - ext)) break;                // what is left is farther than every line's hit
        inner2 = outer2;
        radius *= 2.0;
    }
    // the best of the four lanes: smallest distance, exact ties to the lowest original triangle
#pragma unroll
    for (int off = 1; off < kLineCopies; off <<= 1) {
        const double od = __shfl_xor(best, off), oo = __shfl_xor(bo, off);
        const double ox = __shfl_xor(bp.x, off), oy = __shfl_xor(bp.y, off), oz = __shfl_xor(bp.z, off);
        if (od < best || (od == best && oo < bo)) {
            best = od;
            bo = oo;
            bp = V3{ox, oy, oz};
        }
    }
    if (ok && copy == 0) {
        cp[i] = bp.x;
        cp[fit.n + i] = bp.y;
        cp[2 * fit.n + i] = bp.z;
        found[i] = best < __builtin_huge_val() ? 1 : 0;
    }
}

// The same search over the target's triangle GRID (TriGridDev: a triangle is listed in the cell of its box's lower corner, its box
// reaches at most span[d] cells further; wide triangles sit in a short list): four lanes per line walk the listing slabs along the
// line's dominant axis outward from the vertex, nearest first.  A triangle listed in slab j can meet the line only over the axis
// interval [j, j + 1 + span] h, so the slab's share of the line -- clipped to the distance of the best hit so far -- bounds the cells of
// the other two axes; the lanes take those cells in turn and run the Moeller-Trumbore test (same expressions and early-outs as
// line_nearest_kernel) on their entries.  A direction is finished once the slab's nearest point of the line is farther than the best
// hit.  Every triangle whose box the line can reach within that distance is seen, so the result is the tile scan's, bit for bit
// (nearest intersection, exact ties to the lowest original triangle).
template <int kLanes>
__global__ __launch_bounds__(256) void line_grid_kernel(Cloud fit, const double *__restrict__ dirs, TriGridDev g, double *__restrict__ cp,
                                                        int32_t *__restrict__ found) {
    constexpr int QPB = 256 / kLanes;
    const int ql = threadIdx.x % kLanes, qi = threadIdx.x / kLanes;
    const int64_t i = (int64_t)blockIdx.x * QPB + qi;
    const bool ok = i < fit.n;
    const int64_t ic = ok ? i : 0;
    const V3 p{fit.x[ic], fit.y[ic], fit.z[ic]};
    const V3 dir{dirs[ic], dirs[fit.n + ic], dirs[2 * fit.n + ic]};
    const double pa[3] = {p.x, p.y, p.z}, da[3] = {dir.x, dir.y, dir.z};
    double best = __builtin_huge_val();
    unsigned bo = 0xFFFFFFFFu;
    V3 bp = p;
    auto test_entry = [&](int64_t e) {
        const double *rc = g.recs + e * kTriRec;
        const V3 A{rc[0], rc[1], rc[2]};
        const V3 e1 = sub(V3{rc[3], rc[4], rc[5]}, A), e2 = sub(V3{rc[6], rc[7], rc[8]}, A);
        const V3 pv = cross3(dir, e2);
        const double det = dot3(e1, pv);
        const V3 tv = sub(p, A);
        const double nu = dot3(tv, pv);
        const double ad = fabs(det), su = det > 0.0 ? nu : -nu;
        if (su < -1e-9 * ad || su > ad * (1.0 + 1e-9)) return;  // u clearly outside [0, 1]
        const V3 qv = cross3(tv, e1);
        const double nw = dot3(qv, dir);
        const double sw = det > 0.0 ? nw : -nw;
        if (sw < -1e-9 * ad || su + sw > ad * (1.0 + 2e-9)) return;  // w < 0 or u + w > 1, clearly
        const double inv = 1.0 / det;
        const double u = nu * inv;
        const double w = nw * inv;
        const double tt = dot3(e2, qv) * inv;
        if (det != 0.0 && u >= 0.0 && u <= 1.0 && w >= 0.0 && u + w <= 1.0) {
            const V3 ip{p.x + tt * dir.x, p.y + tt * dir.y, p.z + tt * dir.z};
            if (ip.x != p.x || ip.y != p.y || ip.z != p.z) {
                const V3 dd = sub(ip, p);
                const double dist = sqrt((dd.x * dd.x + dd.y * dd.y) + dd.z * dd.z);
                const unsigned o = (unsigned)((unsigned long long)__builtin_bit_cast(long long, rc[9]) >> 32);
                if (dist < best || (dist == best && o < bo)) {
                    best = dist;
                    bo = o;
                    bp = ip;
                }
            }
        }
    };
    // dominant axis (the same in the kLanes lanes of a line)
    int a = 0;
    if (fabs(da[1]) > fabs(da[a])) a = 1;
    if (fabs(da[2]) > fabs(da[a])) a = 2;
    const int b = a == 0 ? 1 : 0, c = a == 2 ? 1 : 2;
    const double len = sqrt((da[0] * da[0] + da[1] * da[1]) + da[2] * da[2]);
    const bool walk = ok && fabs(da[a]) > 0.0 && len < 1.7976931348623157e308 && pa[0] == pa[0] && pa[1] == pa[1] && pa[2] == pa[2];
    if (walk) {
        for (int64_t e = g.n_listed + ql; e < (int64_t)g.n_listed + g.n_big; e += kLanes) test_entry(e);  // the wide triangles
    }
    double bound = best;
#pragma unroll
    for (int off = 1; off < kLanes; off <<= 1) bound = fmin(bound, __shfl_xor(bound, off));
    if (walk) {
        const double inv_da = 1.0 / da[a], unit = len / fabs(da[a]);  // distance along the line per unit of the dominant axis
        const double fa = (pa[a] - g.lo[a]) * g.inv_h;
        const int ga = g.g[a];
        const int j0 = fa >= (double)(ga - 1) ? ga - 1 : (fa > 0.0 ? (int)fa : 0);
        const double hs = g.h * (double)(1 + g.span[a]);
        bool live[2] = {true, true};
        for (int k = 0; live[0] || live[1]; ++k) {
            for (int sgn = 0; sgn < 2; ++sgn) {
                if (!live[sgn] || (k == 0 && sgn == 1)) continue;
                const int j = sgn == 0 ? j0 + k : j0 - k;
                if (j < 0 || j >= ga) {
                    live[sgn] = false;
                    continue;
                }
                // the axis interval a triangle listed in slab j can occupy, slightly widened
                const double epsa = 1e-9 * (fabs(pa[a]) + fabs(g.lo[a]) + hs * (double)(j + 1)) + 1e-300;
                const double A0 = g.lo[a] + g.h * (double)j - epsa, A1 = g.lo[a] + g.h * (double)j + hs + epsa;
                const double gap = fmax(fmax(A0 - pa[a], pa[a] - A1), 0.0);
                if (gap * unit > bound * (1.0 + 1e-9)) {  // (monotone in k: this direction is done)
                    live[sgn] = false;
                    continue;
                }
                double t0 = (A0 - pa[a]) * inv_da, t1 = (A1 - pa[a]) * inv_da;
                if (t0 > t1) {
                    const double tmp = t0;
                    t0 = t1;
                    t1 = tmp;
                }
                if (bound < __builtin_huge_val()) {  // nothing farther than the best hit matters
                    const double tl = bound / len * (1.0 + 1e-9);
                    t0 = fmax(t0, -tl);
                    t1 = fmin(t1, tl);
                }
                if (t0 <= t1) {
                    int lo_c[2], n_c[2];
                    bool any = true;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int d = s2 == 0 ? b : c;
                        const double x0 = pa[d] + t0 * da[d], x1 = pa[d] + t1 * da[d];
                        const double eps = 1e-9 * (fabs(x0) + fabs(x1) + fabs(g.lo[d]) + g.h) + 1e-300;
                        const double f0 = (fmin(x0, x1) - eps - g.lo[d]) * g.inv_h, f1 = (fmax(x0, x1) + eps - g.lo[d]) * g.inv_h;
                        const int gd = g.g[d];
                        if (!(f1 >= 0.0) || !(f0 < (double)gd + (double)g.span[d] + 1.0)) any = false;  // (also NaN)
                        const double c0 = floor(f0) - (double)g.span[d], c1 = floor(f1);
                        const int i0 = c0 > 0.0 ? (c0 < (double)gd ? (int)c0 : gd) : 0;
                        const int i1 = c1 < (double)(gd - 1) ? (c1 >= 0.0 ? (int)c1 : -1) : gd - 1;
                        lo_c[s2] = i0;
                        n_c[s2] = i1 - i0 + 1;
                        if (n_c[s2] <= 0) any = false;
                    }
                    if (any && a != 0) {
                        // b is the x axis: the cells of a row are one contiguous run of entries; the lanes take rows
                        for (int r = ql; r < n_c[1]; r += kLanes) {
                            int cell3[3];
                            cell3[a] = j;
                            cell3[b] = lo_c[0];
                            cell3[c] = lo_c[1] + r;
                            const int64_t idx = ((int64_t)cell3[2] * g.g[1] + cell3[1]) * g.g[0] + cell3[0];
                            const int32_t e0 = g.cell_start[idx], e1 = g.cell_start[idx + n_c[0]];
                            for (int32_t e = e0; e < e1; ++e) test_entry(e);
                        }
                    } else if (any) {
                        const int ncell = n_c[0] * n_c[1];
                        for (int r = ql; r < ncell; r += kLanes) {
                            const int rb = r % n_c[0], rc2 = r / n_c[0];
                            const int64_t idx = ((int64_t)(lo_c[1] + rc2) * g.g[1] + (lo_c[0] + rb)) * g.g[0] + j;  // (a = x, b = y, c = z)
                            const int32_t e0 = g.cell_start[idx], e1 = g.cell_start[idx + 1];
                            for (int32_t e = e0; e < e1; ++e) test_entry(e);
                        }
                    }
                }
                bound = best;
#pragma unroll
                for (int off = 1; off < kLanes; off <<= 1) bound = fmin(bound, __shfl_xor(bound, off));
            }
        }
    }
    // the best of the line's lanes: smallest distance, exact ties to the lowest original triangle
#pragma unroll
    for (int off = 1; off < kLanes; off <<= 1) {
        const double od = __shfl_xor(best, off);
        const unsigned oo = (unsigned)__shfl_xor((int)bo, off);
        const double ox = __shfl_xor(bp.x, off), oy = __shfl_xor(bp.y, off), oz = __shfl_xor(bp.z, off);
        if (od < best || (od == best && oo < bo)) {
            best = od;
            bo = oo;
            bp = V3{ox, oy, oz};
        }
    }
    if (ok && ql == 0) {
        cp[i] = bp.x;
        cp[fit.n + i] = bp.y;
        cp[2 * fit.n + i] = bp.z;
        found[i] = best < __builtin_huge_val() ? 1 : 0;
    }
}

// first two rejection tests (boundary vertex, opposite normals): pre[i] = 1 when the pair is already rejected
__global__ __launch_bounds__(256) void surface_prereject_kernel(int64_t M, const int32_t *__restrict__ nn_vertex,
                                                                const int32_t *__restrict__ tgt_boundary,
                                                                const double *__restrict__ fit_vn, const double *__restrict__ tgt_vn,
                                                                int64_t N, const int32_t *__restrict__ found,
                                                                int32_t *__restrict__ pre) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int32_t j = nn_vertex[i];
    int r = 0;
    if (found && !found[i])  // along-normal flavour: no intersection -> (p, weight 0)
        r = 1;
    else if (j < 0)
        r = 1;
    else if (tgt_boundary[j])
        r = 1;
    else {
        const double d = (fit_vn[i] * tgt_vn[j] + fit_vn[M + i] * tgt_vn[N + j]) + fit_vn[2 * M + i] * tgt_vn[2 * N + j];
        if (d < 0.0) r = 1;
    }
    pre[i] = r;
}

// w[i] in {0, 1}; weight_in[i] = w[i] / sigma2 (isotropic observation noise, ICP.scala:90-92)
__global__ __launch_bounds__(256) void surface_weight_kernel(int64_t M, const int32_t *__restrict__ pre,
                                                             const int32_t *__restrict__ hit, const double *__restrict__ sigma2,
                                                             double *__restrict__ w01, double *__restrict__ weight_in) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const double w = (pre[i] || hit[i]) ? 0.0 : 1.0;
    w01[i] = w;
    weight_in[i] = w / sigma2[0];
}

// ---- reversed correspondence direction (ClosestPointRegistrator.scala:34-49): N entries (template vertex, target vertex, w)
// keys[j] = template vertex of target j when accepted, else `sentinel` (sorts last); vals[j] = j
__global__ __launch_bounds__(256) void reversal_keys_kernel(int64_t N, const int32_t *__restrict__ nn_vertex,
                                                            const int32_t *__restrict__ pre, const int32_t *__restrict__ hit,
                                                            int32_t sentinel, int32_t *__restrict__ keys, int32_t *__restrict__ vals,
                                                            double *__restrict__ w01) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const bool rejected = (pre && pre[j]) || (hit && hit[j]) || nn_vertex[j] < 0;
    keys[j] = rejected ? sentinel : nn_vertex[j];
    vals[j] = (int32_t)j;
    if (w01) w01[j] = rejected ? 0.0 : 1.0;
}

// Per template vertex i: the accepted target vertices that map to it are a run of the (stably) sorted keys; their mean is the
// observed point and their number times 1 / sigma2 the observation weight (k isotropic observations of one point = one
// observation of their mean with k-fold precision).  The run is summed in ascending target position: deterministic.
__global__ __launch_bounds__(256) void reversal_gather_kernel(int64_t M, int64_t N, const int32_t *__restrict__ skeys,
                                                              const int32_t *__restrict__ svals, Cloud tgt,
                                                              const double *__restrict__ sigma2, double *__restrict__ obs,
                                                              double *__restrict__ weight_in) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    int64_t lo = 0, hi = N;  // first position with key >= i
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (skeys[mid] < (int32_t)i)
            lo = mid + 1;
        else
            hi = mid;
    }
    double sx = 0.0, sy = 0.0, sz = 0.0;
    int64_t k = 0;
    for (int64_t p = lo; p < N && skeys[p] == (int32_t)i; ++p, ++k) {
        const int32_t j = svals[p];
        sx += tgt.x[j];
        sy += tgt.y[j];
        sz += tgt.z[j];
    }
    const double kk = k > 0 ? (double)k : 1.0;
    obs[i] = sx / kk;
    obs[M + i] = sy / kk;
    obs[2 * M + i] = sz / kk;
    weight_in[i] = (double)k / sigma2[0];
}

// Row shard (round 5): the same runs, for a RANGE of the target queries, left as sums -- out[4][M] = {sum x, sum y, sum z, count} per
// template vertex of the WHOLE template -- so that the shards' ranges add up to the totals with one all-reduce; every entry is
// written (zeros where no accepted query of the range maps to the vertex).
__global__ __launch_bounds__(256) void reversal_sums_kernel(int64_t M, int64_t N, const int32_t *__restrict__ skeys,
                                                            const int32_t *__restrict__ svals, Cloud tgt, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    int64_t lo = 0, hi = N;  // first position with key >= i
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (skeys[mid] < (int32_t)i)
            lo = mid + 1;
        else
            hi = mid;
    }
    double sx = 0.0, sy = 0.0, sz = 0.0;
    int64_t k = 0;
    for (int64_t p = lo; p < N && skeys[p] == (int32_t)i; ++p, ++k) {
        const int32_t j = svals[p];
        sx += tgt.x[j];
        sy += tgt.y[j];
        sz += tgt.z[j];
    }
    out[i] = sx;
    out[M + i] = sy;
    out[2 * M + i] = sz;
    out[3 * M + i] = (double)k;
}

// ---------------------------------------------------------------------------------------- surface distance statistics
// IndependentPointDistanceEvaluator (G/api/sampling/evaluators/IndependentPointDistanceEvaluator.scala:54-70) and the accuracy
// metrics of RegistrationComparison (G/api/helper/RegistrationComparison.scala:24-73) are reductions over
// d_i = |p_i - closestPointOnSurface(p_i)|: partial[b] = {sum d, max d, count, sum log N(d; 0, sdev)} of block b, points counted
// when orig[i] < orig_limit (the first orig_limit points in the caller's numbering; orig == null: all) and, with `boundary`,
// when the mesh vertex nearest to the surface point is not a boundary vertex (:67-69).  Fixed grid, fixed reduction order.
constexpr int kStatBlocks = 64;

__global__ __launch_bounds__(256) void dist_stats_kernel(int64_t n, const double *__restrict__ d2, const int32_t *__restrict__ orig,
                                                         int64_t orig_limit, const int32_t *__restrict__ nn,
                                                         const int32_t *__restrict__ boundary, double sdev, double lognorm,
                                                         double *__restrict__ partial) {
    __shared__ double sh[4][256];
    double s = 0.0, mx = 0.0, cnt = 0.0, ll = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kStatBlocks * 256) {
        bool take = !orig || orig[i] < orig_limit;
        if (take && boundary) {
            const int32_t j = nn[i];
            take = j >= 0 && !boundary[j];
        }
        if (!take) continue;
        const double d = sqrt(d2[i]);
        s += d;
        mx = fmax(mx, d);
        cnt += 1.0;
        if (sdev > 0.0) {
            const double u = d / sdev;
            ll += -u * u / 2.0 - lognorm;  // breeze Gaussian.logPdf
        }
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = mx;
    sh[2][threadIdx.x] = cnt;
    sh[3][threadIdx.x] = ll;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
            sh[1][threadIdx.x] = fmax(sh[1][threadIdx.x], sh[1][threadIdx.x + off]);
            sh[2][threadIdx.x] += sh[2][threadIdx.x + off];
            sh[3][threadIdx.x] += sh[3][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(int64_t)blockIdx.x * 4 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void dist_stats_finish_kernel(const double *__restrict__ partial, double *__restrict__ out) {
    if (threadIdx.x >= 4) return;
    double v = 0.0;
    for (int b = 0; b < kStatBlocks; ++b) {
        const double x = partial[b * 4 + threadIdx.x];
        v = threadIdx.x == 1 ? fmax(v, x) : v + x;
    }
    out[threadIdx.x] = v;
}

// Both launches in one for up to kStatBlocks * 256 points (a Metropolis-Hastings step evaluates the likelihood of ~1 600 vertices:
// two dependent launches of 4 us each were all latency).  One wave stands for one block of dist_stats_kernel -- lane l holds the
// elements t = l, l + 64, l + 128, l + 192 of its block -- and adds them in the order of that kernel's LDS tree ((t, t + 128), (t, t + 64),
// then the lanes 32, 16, ... 1 apart), the blocks are added in ascending order as dist_stats_finish_kernel does: the same bits.
__global__ __launch_bounds__(1024) void dist_stats_small_kernel(int64_t n, const double *__restrict__ d2, const int32_t *__restrict__ orig,
                                                                int64_t orig_limit, const int32_t *__restrict__ nn,
                                                                const int32_t *__restrict__ boundary, double sdev, double lognorm,
                                                                double *__restrict__ out) {
    __shared__ double part[kStatBlocks][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblocks = (int)((n + 255) / 256);  // (blocks past the last point hold zeros: adding them changes nothing)
    for (int b = wave; b < nblocks; b += 16) {
        double s[4], mx[4], cnt[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = (int64_t)b * 256 + k * 64 + lane;
            s[k] = mx[k] = cnt[k] = ll[k] = 0.0;
            bool take = i < n && (!orig || orig[i] < orig_limit);
            if (take && boundary) {
                const int32_t j = nn[i];
                take = j >= 0 && !boundary[j];
            }
            if (take) {
                const double d = sqrt(d2[i]);
                s[k] = 0.0 + d;
                mx[k] = fmax(0.0, d);
                cnt[k] = 1.0;
                if (sdev > 0.0) {
                    const double u = d / sdev;
                    ll[k] = 0.0 + (-u * u / 2.0 - lognorm);
                }
            }
        }
        // off = 128, 64: between the four elements of a lane; off = 32 .. 1: between lanes
        double a = (s[0] + s[2]) + (s[1] + s[3]), m = fmax(fmax(mx[0], mx[2]), fmax(mx[1], mx[3])), c = (cnt[0] + cnt[2]) + (cnt[1] + cnt[3]),
               l = (ll[0] + ll[2]) + (ll[1] + ll[3]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            m = fmax(m, __shfl_down(m, off));
            c += __shfl_down(c, off);
            l += __shfl_down(l, off);
        }
        if (lane == 0) part[b][0] = a, part[b][1] = m, part[b][2] = c, part[b][3] = l;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            const double x = part[b][threadIdx.x];
            v = threadIdx.x == 1 ? fmax(v, x) : v + x;
        }
        out[threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------- grid over a MOVING mesh, rebuilt on the device (round 5)
constexpr int kMovGridSetupBlocks = 256;
constexpr int kMovGridScanBlocks = 128;      // all resident at once (the scan's look-back spins on the predecessors' totals)
constexpr int32_t kMovGridMinCells = kMovGridScanBlocks * 1024;  // cell capacity: a power of two between these
constexpr int32_t kMovGridMaxCells = 1 << 20;
constexpr int kMovGridMaxBig = 256;

__device__ __forceinline__ int32_t mov_cell_of(double x, double lo, double inv_h, int32_t gd) {  // the clamped floor every user evaluates
    const double c = floor((x - lo) * inv_h);
    return c >= (double)(gd - 1) ? gd - 1 : (c > 0.0 ? (int32_t)c : 0);
}

// block 0: bounding box of the tile boxes -> geometry at cell edge h (grown until the cells fit), counters of the description zeroed;
// all blocks: the cell counters zeroed
__global__ __launch_bounds__(256) void mov_grid_setup_kernel(const double *__restrict__ tile_boxes, int ntiles, int64_t T, double h, int32_t ncap,
                                                             MovGridParams *__restrict__ P, int32_t *__restrict__ cnt,
                                                             const int32_t *cell_start, const double *boxes, const double *recs) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k <= ncap; k += (int64_t)gridDim.x * 256) cnt[k] = 0;
    if (blockIdx.x != 0) return;
    __shared__ double sh[6][256];
    double lo[3] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val()};
    double hi[3] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
    for (int t = threadIdx.x; t < ntiles; t += 256)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fmin(lo[d], tile_boxes[(int64_t)t * 6 + d]);   // (fmin / fmax skip NaN: a tile of non-finite triangles does not poison the box)
            hi[d] = fmax(hi[d], tile_boxes[(int64_t)t * 6 + 3 + d]);
        }
#pragma unroll
    for (int d = 0; d < 3; ++d) sh[d][threadIdx.x] = lo[d], sh[3 + d][threadIdx.x] = hi[d];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                sh[d][threadIdx.x] = fmin(sh[d][threadIdx.x], sh[d][threadIdx.x + st]);
                sh[3 + d][threadIdx.x] = fmax(sh[3 + d][threadIdx.x], sh[3 + d][threadIdx.x + st]);
            }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double size[3], maxext = 0.0;
    bool fin = true;
    for (int d = 0; d < 3; ++d) {
        size[d] = sh[3 + d][0] - sh[d][0];
        fin = fin && fabs(sh[d][0]) < 1e300 && fabs(sh[3 + d][0]) < 1e300 && size[d] >= 0.0;
        maxext = fmax(maxext, size[d]);
    }
    int32_t gd[3] = {1, 1, 1};
    // cell edge: the caller's, or about the edge of a triangle of a closed surface that fills the box (T triangles on the box's surface
    // area); any value is correct -- the spans measured by the count pass make the queries look far enough
    double hh = h > 0.0 ? h : 0.75 * sqrt(4.0 * (size[0] * size[1] + size[1] * size[2] + size[0] * size[2]) / (double)(T > 0 ? T : 1));
    if (fin && maxext > 0.0 && hh > 0.0) {
        if (!(hh > 1e-9 * maxext)) hh = 1e-9 * maxext;
        for (int it = 0; it < 200; ++it) {
            double cells = 1.0;
            for (int d = 0; d < 3; ++d) {
                const double c = floor(size[d] / hh) + 1.0;
                gd[d] = (int32_t)fmin(c, 512.0);
                cells *= fmin(c, 1e9);
                if (c > 512.0) cells = 1e30;
            }
            if (cells <= (double)ncap) break;
            hh *= 1.25;
        }
        if ((double)gd[0] * gd[1] * gd[2] > (double)ncap) fin = false;
    } else {
        fin = false;
    }
    for (int d = 0; d < 3; ++d) {
        P->v.lo[d] = fin ? sh[d][0] : 0.0;
        P->v.g[d] = fin ? gd[d] : 1;
        P->v.span[d] = 0;
    }
    P->v.h = hh;
    P->v.inv_h = fin ? 1.0 / hh : 0.0;
    P->v.n_listed = 0;
    P->v.n_big = 0;
    P->v.cell_start = cell_start;
    P->v.boxes = boxes;
    P->v.recs = recs;
    P->valid = fin ? 1 : 0;
}

// tri_cell[t]: the cell of the lower corner of triangle t's box (>= 0), -1 a triangle with a non-finite corner (never a candidate),
// -2 - k the k-th entry of the short list of wide triangles
__global__ __launch_bounds__(256) void mov_grid_count_kernel(int64_t T, const double *__restrict__ tribox, MovGridParams *P,
                                                             int32_t *__restrict__ cnt, int32_t *__restrict__ tri_cell) {
    __shared__ int wspan[3][4];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int valid = P->valid;
    const double lo[3] = {P->v.lo[0], P->v.lo[1], P->v.lo[2]}, inv_h = P->v.inv_h;
    const int32_t gd[3] = {P->v.g[0], P->v.g[1], P->v.g[2]};
    int32_t ex[3] = {0, 0, 0};
    int32_t code = -1;
    bool listed = false;
    if (t < T && valid) {
        const double *b = tribox + 6 * t;
        bool fin = true, wide = false;
        int32_t a[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            fin = fin && fabs(b[d]) < 1e300 && fabs(b[3 + d]) < 1e300;
            a[d] = mov_cell_of(b[d], lo[d], inv_h, gd[d]);
            ex[d] = mov_cell_of(b[3 + d], lo[d], inv_h, gd[d]) - a[d];
            wide = wide || ex[d] > kTriGridMaxSpan;
        }
        if (fin && wide) {
            const int32_t k = atomicAdd(&P->v.n_big, 1);
            if (k >= kMovGridMaxBig) P->valid = 0;  // many huge triangles in a fine grid: every query falls back to the tile scan
            code = -2 - k;
        } else if (fin) {
            code = (a[2] * gd[1] + a[1]) * gd[0] + a[0];
            listed = true;
        }
    }
    if (t < T) tri_cell[t] = code;
    if (listed) atomicAdd(&cnt[code], 1);
    // the largest extent of a listed triangle's box, per axis: one atomic per workgroup (every triangle of the mesh would otherwise
    // meet in three words of one cache line: measured 41-90 us for 82k triangles)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int m = listed ? ex[d] : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0) wspan[d][threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        const int m = max(max(wspan[d][0], wspan[d][1]), max(wspan[d][2], wspan[d][3]));
        if (m > 0) atomicMax(&P->v.span[d], m);
    }
}

// start[c] = number of listed triangles in cells < c: exclusive scan of cnt over all ncap cells (cells past the grid hold zeros) by
// kMovGridScanBlocks workgroups in ONE launch -- every workgroup sums its slice, publishes the total tagged with this build's epoch,
// adds up its predecessors' totals (they are all resident: spinning on them is safe) and scans its slice again from there; cnt is
// zeroed on the way (the fill pass uses it as the cursor of every cell); the last workgroup leaves the number of listed entries.
__global__ __launch_bounds__(256) void mov_grid_scan_kernel(MovGridParams *P, int32_t *__restrict__ cnt, int32_t *__restrict__ start, int32_t ncap,
                                                            unsigned long long *agg, unsigned epoch) {
    __shared__ int32_t sh[256];
    __shared__ int32_t s_prefix;
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    const int32_t per = ncap / nb;  // a multiple of 1024
    const int4 *src = reinterpret_cast<const int4 *>(cnt + (int64_t)b * per);
    const int nchunk = per / 1024;
    int32_t tot = 0;
    for (int k = 0; k < nchunk; ++k) {
        const int4 v = src[k * 256 + t];
        tot += (v.x + v.y) + (v.z + v.w);
    }
    sh[t] = tot;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (t < st) sh[t] += sh[t + st];
        __syncthreads();
    }
    const int32_t block_total = sh[0];
    __syncthreads();
    if (t == 0) __hip_atomic_store(&agg[b], ((unsigned long long)epoch << 32) | (unsigned)block_total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int32_t mine = 0;
    if (t < b) {  // (b <= 127 < 256: one predecessor per thread)
        unsigned long long a;
        while ((unsigned)((a = __hip_atomic_load(&agg[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) __builtin_amdgcn_s_sleep(1);
        mine = (int32_t)(unsigned)a;
    }
    sh[t] = mine;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (t < st) sh[t] += sh[t + st];
        __syncthreads();
    }
    if (t == 0) s_prefix = sh[0];
    __syncthreads();
    int32_t carry = s_prefix;
    int4 *dst = reinterpret_cast<int4 *>(start + (int64_t)b * per);
    int4 *zero = reinterpret_cast<int4 *>(cnt + (int64_t)b * per);
    for (int k = 0; k < nchunk; ++k) {
        const int4 v = src[k * 256 + t];
        const int32_t local = (v.x + v.y) + (v.z + v.w);
        sh[t] = local;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {  // inclusive scan of the 256 thread sums
            const int32_t u = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += u;
            __syncthreads();
        }
        const int32_t excl = carry + sh[t] - local;
        const int32_t chunk_total = sh[255];
        dst[k * 256 + t] = int4{excl, excl + v.x, excl + v.x + v.y, excl + v.x + v.y + v.z};
        zero[k * 256 + t] = int4{0, 0, 0, 0};
        carry += chunk_total;
        __syncthreads();
    }
    if (b == nb - 1 && t == 0) {
        start[ncap] = carry;
        P->v.n_listed = carry;
        if (P->v.n_big > kMovGridMaxBig) P->v.n_big = kMovGridMaxBig;  // (valid is 0 then: nobody reads the list)
    }
}

__global__ __launch_bounds__(256) void mov_grid_fill_kernel(int64_t T, Cloud v, const int32_t *__restrict__ tri,
                                                            const int32_t *__restrict__ tri_orig, const double *__restrict__ tribox,
                                                            const MovGridParams *__restrict__ P, const int32_t *__restrict__ start,
                                                            int32_t *__restrict__ cursor, const int32_t *__restrict__ tri_cell,
                                                            double *__restrict__ boxes, double *__restrict__ recs) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T || !P->valid) return;
    const int32_t c = tri_cell[t];
    if (c == -1) return;
    int64_t slot;
    if (c >= 0) {
        slot = start[c] + atomicAdd(&cursor[c], 1);
    } else {
        const int32_t k = -2 - c;
        if (k >= kMovGridMaxBig) return;
        slot = (int64_t)P->v.n_listed + k;
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) boxes[slot * 6 + d] = tribox[6 * t + d];
    double *rc = recs + slot * kTriRec;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int32_t a = tri[3 * t + k];
        rc[3 * k] = v.x[a];
        rc[3 * k + 1] = v.y[a];
        rc[3 * k + 2] = v.z[a];
    }
    const long long meta = (long long)(((unsigned long long)(uint32_t)(tri_orig ? tri_orig[t] : (int32_t)t) << 32) | (unsigned long long)(uint32_t)t);
    rc[9] = __builtin_bit_cast(double, meta);
}

// Self-intersection test (see self_intersect_queue_kernel: same line / triangle arithmetic on the same corner coordinates, same
// margins) over the moving grid: kLanes lanes per query share the entries of the cells the ball of radius |v| around the query reaches.
// A query whose ball covers more than kTriGridMaxCells rows of cells, a non-finite query or an invalid grid is flagged for the
// masked tile scan.  "Some triangle holds a closer intersection" does not depend on the order the entries are visited in.
template <int kLanes>
__global__ __launch_bounds__(256) void self_intersect_grid_kernel(Cloud fit, const double *__restrict__ cp, const MovGridParams *__restrict__ P,
                                                                 const int32_t *__restrict__ skip, int32_t *__restrict__ out,
                                                                 uint8_t *__restrict__ flag, int32_t *__restrict__ nflag,
                                                                 int32_t *__restrict__ nflag_next) {
    constexpr int QPB = 256 / kLanes;
    __shared__ int32_t cell_s[QPB][kTriGridMaxCells], cell_off[QPB][kTriGridMaxCells + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) *nflag_next = 0;
    const TriGridDev g = P->v;
    const bool valid = P->valid != 0;
    const int ql = threadIdx.x % kLanes, qi = threadIdx.x / kLanes;
    const int64_t i = (int64_t)blockIdx.x * QPB + qi;
    const bool inr = i < fit.n;
    const bool ok = inr && !(skip && skip[i]);
    const int64_t ic = inr ? i : 0;
    const V3 pp{fit.x[ic], fit.y[ic], fit.z[ic]};
    const V3 dd0 = sub(pp, V3{cp[ic], cp[fit.n + ic], cp[2 * fit.n + ic]});
    const double vv = dot3(dd0, dd0);
    const double vnorm = sqrt(vv);
    bool fl = ok;  // flagged unless the block below certifies the answer
    int hit = 0;
    if (ok && valid) {
        const double r = vnorm * (1.0 + 1e-9) + 1e-300;
        const double f0[3] = {(pp.x - r - g.lo[0]) * g.inv_h, (pp.y - r - g.lo[1]) * g.inv_h, (pp.z - r - g.lo[2]) * g.inv_h};
        const double f1[3] = {(pp.x + r - g.lo[0]) * g.inv_h, (pp.y + r - g.lo[1]) * g.inv_h, (pp.z + r - g.lo[2]) * g.inv_h};
        bool fin = true;
        int c0[3], c1[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            fin = fin && fabs(f0[d]) < 1e15 && fabs(f1[d]) < 1e15;
            const double a = floor(f0[d]), b = floor(f1[d]);
            c0[d] = a >= (double)(g.g[d] - 1) ? g.g[d] - 1 : (a > 0.0 ? (int)a : 0);
            c1[d] = b >= (double)(g.g[d] - 1) ? g.g[d] - 1 : (b > 0.0 ? (int)b : 0);
        }
        const int x0 = c0[0] > g.span[0] ? c0[0] - g.span[0] : 0, y0 = c0[1] > g.span[1] ? c0[1] - g.span[1] : 0,
                  z0 = c0[2] > g.span[2] ? c0[2] - g.span[2] : 0;
        const int ny = c1[1] - y0 + 1, nz = c1[2] - z0 + 1;
        if (fin && ny * nz + 1 <= kTriGridMaxCells) {
            const int nrow = ny * nz, nrun = nrow + (g.n_big > 0 ? 1 : 0);
            for (int k = ql; k < nrun; k += kLanes) {
                int32_t s0, n0;
                if (k < nrow) {
                    const int rz = k / ny, ry = k - rz * ny;
                    const int64_t rowbase = ((int64_t)(z0 + rz) * g.g[1] + (y0 + ry)) * g.g[0];
                    s0 = g.cell_start[rowbase + x0];
                    n0 = g.cell_start[rowbase + c1[0] + 1] - s0;
                } else {
                    s0 = g.n_listed;
                    n0 = g.n_big;
                }
                cell_s[qi][k] = s0;
                cell_off[qi][k + 1] = n0;
            }
            __threadfence_block();
            if (ql == 0) {
                int32_t off = 0;
                for (int k = 0; k < nrun; ++k) {
                    const int32_t n = cell_off[qi][k + 1];
                    cell_off[qi][k] = off;
                    off += n;
                }
                cell_off[qi][nrun] = off;
            }
            __threadfence_block();
            const int32_t total = cell_off[qi][nrun];
            int k = 0;
            for (int32_t base = 0; base < total && !hit; base += kLanes) {
                const int32_t idx = base + ql;
                if (idx < total) {
                    while (cell_off[qi][k + 1] <= idx) ++k;
                    const int64_t e = cell_s[qi][k] + (idx - cell_off[qi][k]);
                    if (!(point_box_gap2(pp.x, pp.y, pp.z, g.boxes + e * 6) > vv * (1.0 + 1e-9))) {
                        const double *rc = g.recs + e * kTriRec;
                        const V3 A{rc[0], rc[1], rc[2]};
                        const V3 e1 = sub(V3{rc[3], rc[4], rc[5]}, A), e2 = sub(V3{rc[6], rc[7], rc[8]}, A);
                        const V3 pv = cross3(dd0, e2);
                        const double det = dot3(e1, pv);
                        const double inv = 1.0 / det;
                        const V3 tv = sub(pp, A);
                        const double u = dot3(tv, pv) * inv;
                        const V3 qv = cross3(tv, e1);
                        const double w = dot3(qv, dd0) * inv;
                        const double tt = dot3(e2, qv) * inv;
                        if (det != 0.0 && u >= 0.0 && u <= 1.0 && w >= 0.0 && u + w <= 1.0) {
                            const V3 ip{pp.x + tt * dd0.x, pp.y + tt * dd0.y, pp.z + tt * dd0.z};
                            if (ip.x != pp.x || ip.y != pp.y || ip.z != pp.z) {
                                const V3 dd = sub(ip, pp);
                                if (sqrt((dd.x * dd.x + dd.y * dd.y) + dd.z * dd.z) < vnorm) hit = 1;
                            }
                        }
                    }
                }
                // any lane of the query found one: done (sub-wave OR over the query's lanes)
#pragma unroll
                for (int off = kLanes / 2; off > 0; off >>= 1) hit |= __shfl_xor(hit, off);
            }
            fl = false;
        }
    }
    if (inr && ql == 0) {
        flag[i] = fl ? 1 : 0;
        if (!fl) out[i] = ok ? hit : 0;  // (a skipped query: 0, as the tile scan writes)
    }
    const unsigned long long m = __ballot(inr && ql == 0 && fl);
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt, __builtin_popcountll(m));
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(nflag, cnt);
}

}  // namespace

// bits of the largest sort key of the reversed direction (template vertex ids 0 .. M - 1 and the sentinel M)
static int key_bits(int64_t M) {
    int b = 1;
    while (b < 31 && ((int64_t)1 << b) <= M) ++b;
    return b;
}

size_t reversal_sort_temp_bytes(int64_t N) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr,
                                             (int32_t *)nullptr, (int)N);
    return bytes;
}

void launch_reversal_observations(gingr_ctx *ctx, int64_t M, Cloud tgt, const int32_t *nn_vertex, const int32_t *pre,
                                  const int32_t *hit, const double *sigma2_dev, int32_t *keys, int32_t *vals, int32_t *skeys,
                                  int32_t *svals, void *sort_temp, size_t sort_temp_bytes, double *w01_targets, double *obs_soa,
                                  double *weight_in) {
    const int64_t N = tgt.n;
    hipLaunchKernelGGL(reversal_keys_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, nn_vertex, pre, hit,
                       (int32_t)M, keys, vals, w01_targets);
    // LSD radix sort: stable, so equal keys keep ascending target positions; only the bits a key can have (keys <= M, the sentinel)
    (void)hipcub::DeviceRadixSort::SortPairs(sort_temp, sort_temp_bytes, keys, skeys, vals, svals, (int)N, 0, key_bits(M), ctx->stream);
    hipLaunchKernelGGL(reversal_gather_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, M, N, skeys, svals, tgt,
                       sigma2_dev, obs_soa, weight_in);
}

void launch_reversal_sums(gingr_ctx *ctx, int64_t M, Cloud tgt, const int32_t *nn_vertex, const int32_t *pre, const int32_t *hit,
                          int32_t *keys, int32_t *vals, int32_t *skeys, int32_t *svals, void *sort_temp, size_t sort_temp_bytes,
                          double *w01_targets, double *sums4) {
    const int64_t N = tgt.n;
    if (N > 0) {
        hipLaunchKernelGGL(reversal_keys_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, nn_vertex, pre, hit,
                           (int32_t)M, keys, vals, w01_targets);
        (void)hipcub::DeviceRadixSort::SortPairs(sort_temp, sort_temp_bytes, keys, skeys, vals, svals, (int)N, 0, key_bits(M), ctx->stream);
    }
    hipLaunchKernelGGL(reversal_sums_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, M, N, skeys, svals, tgt, sums4);
}

void launch_cell_normals(gingr_ctx *ctx, Cloud v, const int32_t *tri, int64_t T, double *cn) {
    if (T <= 0) return;
    hipLaunchKernelGGL(cell_normals_kernel, dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, ctx->stream, v, tri, T, cn);
}
void launch_vertex_normals(gingr_ctx *ctx, const int32_t *adj_ptr, const int32_t *adj_tri, const double *cn, int64_t T,
                           int64_t n, double *vn) {
    hipLaunchKernelGGL(vertex_normals_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, adj_ptr, adj_tri, cn,
                       T, n, vn);
}
void launch_tri_tile_bbox(gingr_ctx *ctx, Cloud v, const int32_t *tri, int64_t T, double *boxes, double *tribox, double *cell_normals) {
    if (T <= 0) return;
    hipLaunchKernelGGL(tri_tile_bbox_kernel, dim3((unsigned)ceil_div(T, kTriTile)), dim3(256), 0, ctx->stream, v, tri, T, boxes, tribox,
                       cell_normals);
}
// copies of every query held per workgroup (see surface_cp_queue_kernel), from the number of queries
static int surface_h(int64_t nq) {
    // measured (femur chain, 1 622 queries x 3 240 triangles: 1 007 / 1 151 / 1 237 / 1 257 steps per second at H = 2 / 4 / 8 / 16;
    // 41k queries x 82k triangles: 1 452 / 1 470 / 1 441 / 1 275 iterations per second): small meshes want many short workgroups
    return nq <= 4096 ? 16 : (nq <= 16384 ? 8 : 4);
}
void launch_barycentric(gingr_ctx *ctx, Cloud q, Cloud v, const int32_t *tri_by_orig, const int32_t *tri_id, double *bary) {
    hipLaunchKernelGGL(barycentric_kernel, dim3((unsigned)ceil_div(q.n, 256)), dim3(256), 0, ctx->stream, q, v, tri_by_orig, tri_id, bary);
}
void launch_surface_closest_point(gingr_ctx *ctx, Cloud q, Cloud v, const int32_t *tri, const int32_t *tri_orig, int64_t T,
                                  const double *boxes, double *cp_soa, double *d2, int32_t *tri_out, int32_t *warm, bool warm_valid,
                                  const double *tribox, const uint8_t *mask, const int32_t *nmask) {
    // queries per workgroup = 64 / H.  The kernel is bound by its longest workgroups: fewer queries per workgroup = more, shorter
    // workgroups and a tighter query box for the tile pruning.
    const int h = surface_h(q.n);
    // warm: one int32 per query, read as last call's winning triangles when warm_valid, rewritten with this call's
    const int32_t *win = (warm && warm_valid) ? warm : (const int32_t *)nullptr;
    auto go = [&](auto kern, int qpb) {
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(q.n, qpb)), dim3(kCpThreads), 0, ctx->stream, q, v, tri, tri_orig, T, boxes,
                           cp_soa, d2, tri_out, win, warm, tribox, mask, nmask);
    };
    if (h == 8)
        go(surface_cp_queue_kernel<8>, 8);
    else if (h == 16)
        go(surface_cp_queue_kernel<16>, 4);
    else
        go(surface_cp_queue_kernel<4>, 16);
}

int mov_grid_alloc(gingr_ctx *ctx, int64_t T, int64_t max_queries, MovGrid *g) {
    mov_grid_free(g);
    if (T < 1 || T > INT32_MAX || max_queries < 1) return GINGR_OK;
    int64_t ncap = kMovGridMinCells;  // a power of two (the scan splits it evenly): about 4 cells per triangle -- a surface fills few of a box's cells
    while (ncap < 4 * T && ncap < kMovGridMaxCells) ncap *= 2;
    g->ncap = (int32_t)ncap;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->scan_agg), kMovGridScanBlocks * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(g->scan_agg, 0, kMovGridScanBlocks * sizeof(unsigned long long), ctx->stream));
    g->T = T;
    g->max_queries = max_queries;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->params), sizeof(MovGridParams)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->cell_cnt), (size_t)(ncap + 1) * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->cell_start), (size_t)(ncap + 1) * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->tri_cell), (size_t)T * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->boxes), (size_t)(T + kMovGridMaxBig) * 6 * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->recs), (size_t)(T + kMovGridMaxBig) * kTriRec * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->flag), (size_t)max_queries));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->nflag), 2 * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemsetAsync(g->params, 0, sizeof(MovGridParams), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(g->flag, 0, (size_t)max_queries, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(g->nflag, 0, 2 * sizeof(int32_t), ctx->stream));
    g->h = 0.0;
    g->ready = true;
    return GINGR_OK;
}
void mov_grid_free(MovGrid *g) {
    void *ptrs[] = {g->params, g->cell_cnt, g->cell_start, g->tri_cell, g->big, g->boxes, g->recs, g->flag, g->nflag, g->scan_agg};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    *g = MovGrid{};
}
void launch_mov_grid_build(gingr_ctx *ctx, MovGrid &g, Cloud v, const int32_t *tri, const int32_t *tri_orig, const double *tribox,
                           const double *tile_boxes) {
    const int64_t T = g.T;
    const int ntiles = (int)ceil_div(T, kTriTile);
    hipLaunchKernelGGL(mov_grid_setup_kernel, dim3(kMovGridSetupBlocks), dim3(256), 0, ctx->stream, tile_boxes, ntiles, T, g.h, g.ncap, g.params,
                       g.cell_cnt, g.cell_start, g.boxes, g.recs);
    hipLaunchKernelGGL(mov_grid_count_kernel, dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, ctx->stream, T, tribox, g.params, g.cell_cnt, g.tri_cell);
    hipLaunchKernelGGL(mov_grid_scan_kernel, dim3(kMovGridScanBlocks), dim3(256), 0, ctx->stream, g.params, g.cell_cnt, g.cell_start, g.ncap,
                       g.scan_agg, ++g.epoch);
    hipLaunchKernelGGL(mov_grid_fill_kernel, dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, ctx->stream, T, v, tri, tri_orig, tribox, g.params,
                       g.cell_start, g.cell_cnt, g.tri_cell, g.boxes, g.recs);
}
void launch_self_intersect_grid(gingr_ctx *ctx, Cloud fit, const double *cp_soa, MovGrid &g, const int32_t *skip, int32_t *flag) {
    g.parity ^= 1;
    int32_t *cur = g.nflag + g.parity, *next = g.nflag + (g.parity ^ 1);
    constexpr int kLanes = 8;
    hipLaunchKernelGGL(self_intersect_grid_kernel<kLanes>, dim3((unsigned)ceil_div(fit.n, 256 / kLanes)), dim3(256), 0, ctx->stream, fit, cp_soa,
                       g.params, skip, flag, g.flag, cur, next);
}

void tri_grid_free(TriGrid *g) {
    if (g->cell_start) (void)hipFree(g->cell_start);
    if (g->boxes) (void)hipFree(g->boxes);
    if (g->recs) (void)hipFree(g->recs);
    if (g->flag) (void)hipFree(g->flag);
    if (g->nflag) (void)hipFree(g->nflag);
    *g = TriGrid{};
}

// vsoa: host, the mesh vertices as SoA planes [3][n] in DEVICE order; tri: host, [3 T] vertex positions in the (spatially sorted)
// triangle order of the device.  Synchronous.  No grid (g->ready false) for degenerate extents: the callers keep the tile scan.
int tri_grid_build(gingr_ctx *ctx, const double *vsoa, int64_t n, const int32_t *tri, const int32_t *tri_orig, int64_t T, int64_t max_queries,
                   TriGrid *g) {
    tri_grid_free(g);
    if (T < 1 || T > INT32_MAX || n < 1 || max_queries < 1) return GINGR_OK;
    double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    std::vector<double> tb((size_t)6 * T);
    std::vector<char> good((size_t)T, 0);
    double ext_sum = 0.0;
    int64_t ngood = 0;
    for (int64_t t = 0; t < T; ++t) {
        double bl[3], bh[3];
        bool fin = true;
        for (int d = 0; d < 3; ++d) {
            const double a = vsoa[(size_t)d * n + tri[3 * t]], b = vsoa[(size_t)d * n + tri[3 * t + 1]], c = vsoa[(size_t)d * n + tri[3 * t + 2]];
            fin = fin && std::isfinite(a) && std::isfinite(b) && std::isfinite(c);
            bl[d] = std::min(a, std::min(b, c));
            bh[d] = std::max(a, std::max(b, c));
        }
        if (!fin) continue;  // a triangle with a non-finite corner is never the closest one (its distance is NaN)
        good[(size_t)t] = 1;
        ++ngood;
        double ext = 0.0;
        for (int d = 0; d < 3; ++d) {
            tb[(size_t)6 * t + d] = bl[d];
            tb[(size_t)6 * t + 3 + d] = bh[d];
            lo[d] = std::min(lo[d], bl[d]);
            hi[d] = std::max(hi[d], bh[d]);
            ext = std::max(ext, bh[d] - bl[d]);
        }
        ext_sum += ext;
    }
    if (ngood == 0) return GINGR_OK;
    double size[3], maxext = 0.0;
    for (int d = 0; d < 3; ++d) size[d] = hi[d] - lo[d], maxext = std::max(maxext, size[d]);
    if (!(maxext > 0.0) || !(maxext < 1e300)) return GINGR_OK;
    // cell edge = the mean extent of a triangle's box
    double h = ext_sum / (double)ngood;
    if (!(h > 1e-9 * maxext)) h = 1e-9 * maxext;
    int32_t gd[3];
    for (;;) {
        double cells = 1.0;
        for (int d = 0; d < 3; ++d) {
            const double c = std::floor(size[d] / h) + 1.0;
            gd[d] = (int32_t)std::min(c, 512.0);
            cells *= std::min(c, 1e9);
            if (c > 512.0) cells = 1e30;
        }
        if (cells <= std::min(16.0 * (double)T + 4096.0, 134217728.0)) break;
        h *= 1.25;
    }
    const double inv_h = 1.0 / h;
    auto cell_of = [&](double x, int d) {  // the expression the kernel evaluates (clamped floor)
        const double c = std::floor((x - lo[d]) * inv_h);
        return c >= (double)(gd[d] - 1) ? gd[d] - 1 : (c > 0.0 ? (int32_t)c : 0);
    };
    const int64_t ncells = (int64_t)gd[0] * gd[1] * gd[2];
    // Every triangle is listed ONCE, in the cell of its box's lower corner; a query then looks at the cells [c0 - E, c1] per axis,
    // E = the largest extent (in cells) of a listed triangle's box -- any triangle whose box reaches into the ball [c0, c1] has its
    // lower corner there.  Triangles spanning more than kTriGridMaxSpan cells of an axis go to a short list every query tests.
    std::vector<int32_t> start((size_t)ncells + 1, 0), hcell((size_t)T, -1), big;
    int32_t E[3] = {0, 0, 0};
    for (int64_t t = 0; t < T; ++t) {
        if (!good[(size_t)t]) continue;
        int32_t a[3], ex[3];
        bool wide = false;
        for (int d = 0; d < 3; ++d) {
            a[d] = cell_of(tb[(size_t)6 * t + d], d);
            ex[d] = cell_of(tb[(size_t)6 * t + 3 + d], d) - a[d];
            wide = wide || ex[d] > kTriGridMaxSpan;
        }
        if (wide) {
            big.push_back((int32_t)t);
            continue;
        }
        for (int d = 0; d < 3; ++d) E[d] = std::max(E[d], ex[d]);
        hcell[(size_t)t] = (int32_t)(((int64_t)a[2] * gd[1] + a[1]) * gd[0] + a[0]);
        start[(size_t)hcell[(size_t)t] + 1]++;
    }
    if (big.size() > 256) return GINGR_OK;  // many huge triangles in a fine grid: keep the tile scan
    for (int64_t c = 0; c < ncells; ++c) start[(size_t)c + 1] += start[(size_t)c];
    const int64_t n_listed = start[(size_t)ncells], total = n_listed + (int64_t)big.size();
    std::vector<int32_t> list((size_t)(total > 0 ? total : 1)), fill(start.begin(), start.end() - 1);
    for (int64_t t = 0; t < T; ++t)  // ascending triangle position inside a cell
        if (hcell[(size_t)t] >= 0) list[(size_t)fill[(size_t)hcell[(size_t)t]]++] = (int32_t)t;
    for (size_t k = 0; k < big.size(); ++k) list[(size_t)n_listed + k] = big[k];
    // per ENTRY, contiguous in cell order: the box (48 bytes, all the first test reads) and, apart from it, corners + {device position |
    // original index} (80 bytes, read for the survivors).  The mesh is fixed (the target), so nothing is chased through vertex ids.
    std::vector<double> boxes(list.size() * (size_t)6, 0.0), recs(list.size() * (size_t)kTriRec, 0.0);
    for (size_t e2 = 0; e2 < (size_t)total; ++e2) {
        const int64_t t = list[e2];
        for (int d = 0; d < 6; ++d) boxes[e2 * 6 + d] = tb[(size_t)6 * t + d];
        double *rc = recs.data() + e2 * kTriRec;
        for (int c = 0; c < 3; ++c)
            for (int d = 0; d < 3; ++d) rc[3 * c + d] = vsoa[(size_t)d * n + tri[3 * t + c]];
        const long long meta = (long long)(((unsigned long long)(uint32_t)(tri_orig ? tri_orig[t] : (int32_t)t) << 32) | (unsigned long long)(uint32_t)t);
        memcpy(rc + 9, &meta, sizeof(meta));
    }
    HIP_TRY(ctx, hipMalloc(&g->boxes, boxes.size() * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(g->boxes, boxes.data(), boxes.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMalloc(&g->recs, recs.size() * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(g->recs, recs.data(), recs.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMalloc(&g->cell_start, start.size() * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc(&g->flag, (size_t)max_queries));
    HIP_TRY(ctx, hipMalloc(&g->nflag, 2 * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpyAsync(g->cell_start, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(g->flag, 0, (size_t)max_queries, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(g->nflag, 0, 2 * sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int d = 0; d < 3; ++d) g->v.lo[d] = lo[d], g->v.g[d] = gd[d], g->v.span[d] = E[d];
    g->v.h = h;
    g->v.inv_h = inv_h;
    g->v.cell_start = g->cell_start;
    g->v.boxes = g->boxes;
    g->v.recs = g->recs;
    g->v.n_listed = (int32_t)n_listed;
    g->v.n_big = (int32_t)big.size();
    g->max_queries = max_queries;
    g->list_entries = total;
    g->ready = true;
    return GINGR_OK;
}

// closest point of every query the grid certifies (warm start required: `warm` holds last scan's triangles); the others are flagged
// (g.flag, g.cur_nflag()) for the masked launch_surface_closest_point that must follow
void launch_surface_cp_grid(gingr_ctx *ctx, Cloud q, Cloud v, const int32_t *tri, const int32_t *tri_orig, int64_t T, TriGrid &g,
                            double *cp_soa, double *d2, int32_t *tri_out, int32_t *warm) {
    g.parity ^= 1;
    int32_t *cur = g.nflag + g.parity, *next = g.nflag + (g.parity ^ 1);
    constexpr int kLanes = GINGR_TRI_GRID_LANES;  // queries per wave = 64 / kLanes: their entries and candidates are spread over the lanes
    hipLaunchKernelGGL(surface_cp_grid_kernel<kLanes>, dim3((unsigned)ceil_div(q.n, 256 / kLanes)), dim3(256), 0, ctx->stream, q, v, tri, tri_orig,
                       T, g.v, cp_soa, d2, tri_out, warm, g.flag, cur, next);
}

int distance_stats_ws_doubles() { return kStatBlocks * 4; }
void launch_distance_stats(gingr_ctx *ctx, int64_t n, const double *d2, const int32_t *orig, int64_t orig_limit, const int32_t *nn,
                           const int32_t *boundary, double sdev, double *partial, double *out4) {
    const double lognorm = sdev > 0.0 ? log(sqrt(2.0 * M_PI)) + log(sdev) : 0.0;
    if (n <= (int64_t)kStatBlocks * 256) {  // every (block, thread) of the two-launch form holds at most one point
        hipLaunchKernelGGL(dist_stats_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, n, d2, orig, orig_limit, nn, boundary, sdev, lognorm, out4);
        return;
    }
    hipLaunchKernelGGL(dist_stats_kernel, dim3(kStatBlocks), dim3(256), 0, ctx->stream, n, d2, orig, orig_limit, nn, boundary, sdev,
                       lognorm, partial);
    hipLaunchKernelGGL(dist_stats_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, partial, out4);
}
void launch_self_intersect(gingr_ctx *ctx, Cloud fit, const double *cp_soa, const int32_t *tri, int64_t T, const double *boxes,
                           const int32_t *skip, int32_t *flag, const double *tribox, const Cloud *mesh, const uint8_t *only,
                           const int32_t *nonly, const SelfIntersectFuse *fuse) {
#ifdef GINGR_SI_H
    const int h = GINGR_SI_H;
#else
    const int h = surface_h(fit.n);
#endif
    const Cloud v = mesh ? *mesh : fit;
    auto go = [&](auto kern, int qpb) {
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(fit.n, qpb)), dim3(kCpThreads), 0, ctx->stream, fit, cp_soa, v, tri, T, boxes,
                           skip, flag, tribox, only, nonly, fuse ? *fuse : SelfIntersectFuse{});
    };
    if (h == 8)
        go(self_intersect_queue_kernel<8>, 8);
    else if (h == 16)
        go(self_intersect_queue_kernel<16>, 4);
    else
        go(self_intersect_queue_kernel<4>, 16);
}
void launch_surface_prereject(gingr_ctx *ctx, int64_t M, const int32_t *nn_vertex, const int32_t *tgt_boundary,
                              const double *fit_vn, const double *tgt_vn, int64_t N, const int32_t *found, int32_t *pre) {
    hipLaunchKernelGGL(surface_prereject_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, M, nn_vertex,
                       tgt_boundary, fit_vn, tgt_vn, N, found, pre);
}
void launch_line_nearest(gingr_ctx *ctx, Cloud fit, const double *dirs_soa, Cloud v, const int32_t *tri, const int32_t *tri_orig,
                         int64_t T, double *boxes, double *cp_soa, int32_t *found) {
    const int nt = (int)ceil_div(T, kTriTile);
    hipLaunchKernelGGL(line_group_boxes_kernel, dim3((unsigned)ceil_div((int64_t)6 * ceil_div(nt, kLineGroup), 64)), dim3(64), 0, ctx->stream,
                       boxes, nt);
    hipLaunchKernelGGL(line_nearest_kernel, dim3((unsigned)ceil_div(fit.n, kLineQueries)), dim3(kSurfThreads), 0, ctx->stream, fit,
                       dirs_soa, v, tri, tri_orig, T, boxes, cp_soa, found);
}
#ifndef GINGR_LINE_GRID_LANES
#define GINGR_LINE_GRID_LANES 4  // (measured at 41k x 82k: 1 / 2 / 4 / 8 / 16 / 32 lanes per line: 0.321 / 0.265 / 0.235 / 0.237 / 0.244 / 0.280 ms per iteration)
#endif
void launch_line_nearest_grid(gingr_ctx *ctx, Cloud fit, const double *dirs_soa, const TriGrid &g, double *cp_soa, int32_t *found) {
    constexpr int kLanes = GINGR_LINE_GRID_LANES;
    hipLaunchKernelGGL(line_grid_kernel<kLanes>, dim3((unsigned)ceil_div(fit.n, 256 / kLanes)), dim3(256), 0, ctx->stream, fit, dirs_soa, g.v, cp_soa,
                       found);
}
void launch_surface_weight(gingr_ctx *ctx, int64_t M, const int32_t *pre, const int32_t *hit, const double *sigma2_dev, double *w01,
                           double *weight_in) {
    hipLaunchKernelGGL(surface_weight_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, M, pre, hit, sigma2_dev,
                       w01, weight_in);
}

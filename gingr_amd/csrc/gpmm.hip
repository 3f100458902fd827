// GPMM construction on the device: the low-rank model of
//     GPMMTriangleMesh3D(reference, relativeTolerance).Gaussian / GaussianMixture / AutomaticGaussian
//         (G/api/gpmm/GPMMHelper.scala:96-130)  and  automaticGPMMfromTemplate (G/api/registration/utils/GPMMHelper.scala:39-69)
// i.e.  LowRankGaussianProcess.approximateGPCholesky(reference, GaussianProcess(DiagonalKernel(k, 3)), relativeTolerance,
// NearestNeighborInterpolator)  (GPMM.construct, GPMMHelper.scala:39-55)  with  k = sum_i scaling_i exp(-|x-y|^2/sigma_i^2),
// built straight into a gingr_model: the 3M x r basis never crosses PCIe.
//
// scalismo's algorithm (restated in oracle/gingr_oracle.py: pivoted_cholesky_matrix_valued, approximate_eig): pivoted
// Cholesky over the 3M (point, coordinate) indices, stopped when the residual trace falls below relTol * trace, then the
// eigen-decomposition of L L^T through the SVD of L^T L.  For a DiagonalKernel the three coordinates never interact and
// ties are broken by position, so the generic pivot sequence is (P0,x),(P0,y),(P0,z),(P1,x),... with P0,P1,.. the pivot
// sequence of the SCALAR kernel; after n = 3j + e pivots the coordinates d < e own j+1 scalar columns and the others j,
// and the residual trace is (3-e) tr_s(j) + e tr_s(j+1).  So the device runs the scalar factorisation (M x k_s), finds
// n from the recorded scalar traces, eigen-decomposes the leading (j+1)x(j+1) and jxj blocks of L_s^T L_s (cyclic Jacobi,
// one workgroup) and scatters  Q0 = U sqrt(lambda) = L_s V  into the three coordinate planes of the basis.
//
// The other kernels of GPMMTriangleMesh3D (GPMMHelper.scala:103-142) go through gingr_gpmm_build_diagonal: one scalar kernel per
// coordinate (Gaussian mixture with an optional x-mirrored copy, linear, lookup table).  Three equal kernels take the scalar route
// above; different kernels (GaussianSymmetry) run the generic factorisation over the 3M entries themselves (pc_*_kernel<true>).
// Exact ties between residuals follow scalismo's rule -- the first maximum in the permuted index order -- through a log of the
// swaps that is replayed for tied candidates (PosLog).
//
// Kernels here are one-off (model construction), not the per-iteration path; they are written for exactness of the pivot
// rule (unfused |x-y|^2, multiply-then-add dot products in ascending column order like the JVM) rather than for speed.
#include "gp.h"

#include "fastexp.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kMaxMix = 8;
constexpr int kPcBlock = 256;

struct Mixture {
    int32_t n;
    double c[kMaxMix];  // -2048 log2(e) / sigma^2
    double s[kMaxMix];  // scaling
};

__device__ __forceinline__ double mixture_value(const Mixture &mix, double d2, const double *T) {
    double v = 0.0;
    for (int i = 0; i < mix.n; ++i) {
        const double e = mix.s[i] * fastexp2_scaled<3>(fmin(d2, fastexp_d2_limit(mix.c[i])), mix.c[i], T);
        v = i == 0 ? e : v + e;
    }
    return v;
}

// One scalar kernel of a DiagonalKernel(k_x, k_y, k_z) (GPMMHelper.scala:96-142, KernelHelper.scala:25-84):
//   kind 0  sum_i scaling_i exp(-|x-y|^2/sigma_i^2)  [+ mirror * the same kernel at (Mx, y), M = diag(-1,1,1): the x-mirrored
//           kernel of KernelHelper.symmetrizeKernel -- xMirroredKernel3D evaluates kernel(Point(-x0, x1, x2), y)]
//   kind 1  scale * (x . y)          DotProductKernel.k returns x.dot(y) whatever kernel / gamma it wraps (KernelHelper.scala:43-51)
//   kind 2  scale * m[i][j]          LookupKernel on the reference points themselves (closest reference point of a reference
//                                     point = the point), m = pinv(graph Laplacian) (LaplacianHelper.scala:27-41)
struct KSpec {
    int32_t kind;
    Mixture mix;
    double mirror;
    double scale;
    const double *lookup;  // device, M x M row-major
};

__device__ __forceinline__ double kspec_value(const KSpec &k, const Cloud &pts, int64_t c, int64_t p, double px, double py,
                                              double pz, const double *T) {
    if (k.kind == 1) {
        const double dot = __dadd_rn(__dadd_rn(__dmul_rn(pts.x[c], px), __dmul_rn(pts.y[c], py)), __dmul_rn(pts.z[c], pz));
        return __dmul_rn(dot, k.scale);
    }
    if (k.kind == 2) return __dmul_rn(k.lookup[c * pts.n + p], k.scale);
    const double dx = pts.x[c] - px, dy = pts.y[c] - py, dz = pts.z[c] - pz;
    const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
    double v = mixture_value(k.mix, d2, T);
    if (k.mirror != 0.0) {
        const double mx = __dmul_rn(pts.x[c], -1.0) - px;
        const double m2 = __dadd_rn(__dadd_rn(__dmul_rn(mx, mx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
        v = __dadd_rn(v, __dmul_rn(mixture_value(k.mix, m2, T), k.mirror));
    }
    return v;
}

struct Best {
    double v;
    int32_t i;
};

// scalismo's loop takes the FIRST maximal residual in its current PERMUTED index order: step s swaps the pivot into slot s and the
// element that sat there into the pivot's old slot.  With distinct residuals the order is irrelevant, but structured inputs tie
// exactly (a regular grid under a stationary kernel, mirror partners under the mirrored kernel, all entries at step 0), and a rank
// cut inside a group of tied pivots then depends on who went first.  The swaps are kept as a log -- step s: pivot piv[s], displaced
// element dis[s], the pivot's old slot old[s] -- and a tie is resolved by replaying it for the two candidates (O(steps), ties are
// rare).  `vp / vd / vo` is one more entry that is not in memory yet (the step a kernel is executing).
struct PosLog {
    const int32_t *piv, *dis, *old;
    int32_t n;           // entries in memory
    int32_t vp, vd, vo;  // virtual entry (vp < 0: none)
};

// Slot of element e after the logged swaps.  The replay -- "a pivot stays where it was put; a displaced element goes to the pivot's old
// slot" -- only depends on the step that pivoted e (at most one, and e never moves after it) and otherwise on the LAST step that displaced
// it, so the loop carries two selects and no branch: its reads do not depend on one another and run pipelined (as a literal replay with
// an early return every logged step was a dependent round trip, and a mirrored kernel's structural ties replay the log dozens of times
// per step: pc_step_kernel<true> 131 -> 113 us on average with the log in LDS, -> see profiles/r06_gpmm_build_ab_blocked_kernels.txt).
__device__ __forceinline__ int32_t log_position(const PosLog &lg, int32_t e) {
    int32_t sp = -1, sl = -1;
    for (int32_t s = 0; s < lg.n; ++s) {
        sp = lg.piv[s] == e ? s : sp;
        sl = lg.dis[s] == e ? s : sl;
    }
    if (sp >= 0) return sp;  // pivots stay where they were put (they never compete again)
    int32_t q = sl >= 0 ? lg.old[sl] : e;
    if (lg.vp >= 0) {
        if (lg.vp == e) return lg.n;
        if (lg.vd == e) q = lg.vo;
    }
    return q;
}

// the element in slot `slot` (>= lg.n) after the logged swaps: the one the last swap into that slot put there
__device__ __forceinline__ int32_t log_occupant(const PosLog &lg, int32_t slot) {
    int32_t sl = -1;
    for (int32_t s = 0; s < lg.n; ++s) sl = lg.old[s] == slot ? s : sl;
    return sl >= 0 ? lg.dis[sl] : slot;
}

__device__ __forceinline__ Best better(Best a, Best b, const PosLog &lg) {  // larger value; ties -> earlier slot; NaN never wins
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    if (b.v > a.v) return b;
    if (b.v == a.v && log_position(lg, b.i) < log_position(lg, a.i)) return b;
    return a;
}

// block-wide (max, argmax, sum) with a fixed tree: identical in every workgroup that reduces the same data
__device__ __forceinline__ void block_best_sum(Best &b, double &sum, Best *shb, double *shs, const PosLog &lg) {
    const int t = threadIdx.x;
    shb[t] = b;
    shs[t] = sum;
    __syncthreads();
    for (int off = kPcBlock / 2; off > 0; off >>= 1) {
        if (t < off) {
            shb[t] = better(shb[t], shb[t + off], lg);
            shs[t] += shs[t + off];
        }
        __syncthreads();
    }
    b = shb[0];
    sum = shs[0];
    __syncthreads();
}

// GEN = false: one scalar kernel, entries = points.  GEN = true: the generic factorisation over the 3M (point, coordinate) entries
// of a DiagonalKernel(k_x, k_y, k_z) whose coordinates have DIFFERENT kernels (entry e = 3 point + coordinate; entries of
// different coordinates never interact: their column values are exact zeros).
struct KSpec3 {
    KSpec k[3];
};

template <bool GEN>
__global__ __launch_bounds__(kPcBlock) void pc_init_kernel(Cloud pts, KSpec3 specs, double *__restrict__ diag,
                                                           int32_t *__restrict__ pivoted, double *__restrict__ pmax,
                                                           int32_t *__restrict__ pidx, double *__restrict__ ptr,
                                                           int32_t *__restrict__ ctl) {
    __shared__ double T[GINGR_EXP_TABLE];
    __shared__ Best shb[kPcBlock];
    __shared__ double shs[kPcBlock];
    fastexp_table_init(T);
    __syncthreads();
    const int64_t n = GEN ? 3 * pts.n : pts.n;
    const int64_t c = (int64_t)blockIdx.x * kPcBlock + threadIdx.x;
    Best b{0.0, -1};
    double tr = 0.0;
    if (c < n) {
        const int64_t pc = GEN ? c / 3 : c;
        const KSpec &spec = specs.k[GEN ? (int)(c - 3 * pc) : 0];
        double d0;
        if (spec.kind == 0 && spec.mirror == 0.0) {
            d0 = 0.0;
            for (int i = 0; i < spec.mix.n; ++i) d0 = i == 0 ? spec.mix.s[i] : d0 + spec.mix.s[i];  // k(x,x) = sum scaling_i * exp(0)
        } else {
            d0 = kspec_value(spec, pts, pc, pc, pts.x[pc], pts.y[pc], pts.z[pc], T);
        }
        diag[c] = d0;
        pivoted[c] = 0;
        b = Best{d0, (int32_t)c};
        tr = d0;
    }
    const PosLog none{nullptr, nullptr, nullptr, 0, -1, 0, 0};  // nothing swapped yet: ties -> lowest index
    block_best_sum(b, tr, shb, shs, none);
    if (threadIdx.x == 0) {
        pmax[blockIdx.x] = b.v;
        pidx[blockIdx.x] = b.i;
        ptr[blockIdx.x] = tr;
        if (blockIdx.x == 0) ctl[0] = ctl[1] = 0;
    }
}

// One pivot step.  Every workgroup first reduces the previous step's block partials (same data, same tree => same pivot
// everywhere, no grid synchronisation), then fills its rows of column k.
template <bool GEN>
__global__ __launch_bounds__(kPcBlock) void pc_step_kernel(Cloud pts, KSpec3 specs, int32_t k, int32_t kmax, double rel_tol,
                                                           int32_t nblocks, double *__restrict__ L /* [kmax][n] */,
                                                           double *__restrict__ diag, int32_t *__restrict__ pivoted,
                                                           const double *__restrict__ pmax_in,
                                                           const int32_t *__restrict__ pidx_in,
                                                           const double *__restrict__ ptr_in, double *__restrict__ pmax,
                                                           int32_t *__restrict__ pidx, double *__restrict__ ptr,
                                                           int32_t *__restrict__ ctl, double *__restrict__ trace,
                                                           int32_t *__restrict__ pivots, int32_t *__restrict__ sw_dis,
                                                           int32_t *__restrict__ sw_old) {
    __shared__ double T[GINGR_EXP_TABLE];
    __shared__ Best shb[kPcBlock];
    __shared__ double shs[kPcBlock];
    __shared__ double Lp[512];
    __shared__ int32_t s_piv[512], s_dis[512], s_old[512];
    __shared__ int32_t finished, sh_dis, sh_old;
    const int t = threadIdx.x;
    if (t == 0) finished = ctl[1];  // set by an earlier launch (or, harmlessly, by workgroup 0 of this one)
    __syncthreads();
    if (finished) return;
    fastexp_table_init(T);
    // The swaps of steps 0 .. k-1 (written by earlier launches), copied to LDS first: every workgroup replays the log twice per step
    // for this step's swap, and once per candidate for every exact tie (the y and z entries of a point under a mirrored kernel tie
    // structurally).  Read from global memory the replay was one dependent round trip per logged step -- 0.5 us per step of the
    // log for a single-kernel model, 0.8 us for a mirrored one: 365 ms of the 11 A/B builds of tools/experiments/gpmm_build_ab.py,
    // more than everything else of the set-up together.
    for (int r = t; r < k; r += kPcBlock) {
        s_piv[r] = pivots[r];
        s_dis[r] = sw_dis[r];
        s_old[r] = sw_old[r];
    }
    __syncthreads();
    PosLog lg{s_piv, s_dis, s_old, k, -1, 0, 0};
    Best g{0.0, -1};
    double gtr = 0.0;
    for (int b = t; b < nblocks; b += kPcBlock) {
        g = better(g, Best{pmax_in[b], pidx_in[b]}, lg);
        gtr += ptr_in[b];
    }
    block_best_sum(g, gtr, shb, shs, lg);
    const double tol = rel_tol * trace[0];  // trace[0] was written by step 0 (k == 0 uses gtr itself)
    const bool stop = k >= kmax || g.i < 0 || !(gtr >= (k == 0 ? rel_tol * gtr : tol)) || !(g.v > 0.0);
    if (t == 0 && !stop) {  // this step's swap: every workgroup derives it, workgroup 0 records it
        sh_old = log_position(lg, g.i);
        sh_dis = log_occupant(lg, k);
    }
    if (blockIdx.x == 0 && t == 0) {
        trace[k] = gtr;
        if (stop) {
            ctl[0] = k;
            ctl[1] = 1;
        } else {
            pivots[k] = g.i;
            sw_dis[k] = sh_dis;
            sw_old[k] = sh_old;
        }
    }
    if (stop) return;
    const int64_t M = pts.n, n = GEN ? 3 * M : M;
    const int64_t p = g.i;
    const int64_t pp = GEN ? p / 3 : p;
    const int pd = GEN ? (int)(p - 3 * pp) : 0;
    const KSpec &spec = specs.k[pd];
    const double lpk = sqrt(g.v);
    for (int r = t; r < k; r += kPcBlock) Lp[r] = L[(int64_t)r * n + p];
    __syncthreads();
    lg.vp = (int32_t)p, lg.vd = sh_dis, lg.vo = sh_old;  // the candidates of the NEXT step compete in the order after this swap
    const double px = pts.x[pp], py = pts.y[pp], pz = pts.z[pp];
    const int64_t c = (int64_t)blockIdx.x * kPcBlock + t;
    Best b{0.0, -1};
    double tr = 0.0;
    if (c < n) {
        const int64_t pc = GEN ? c / 3 : c;
        double l;
        if (c == p) {
            l = lpk;
            pivoted[c] = 1;
        } else if (pivoted[c]) {
            l = 0.0;
        } else if (GEN && (int)(c - 3 * pc) != pd) {
            l = 0.0;  // another coordinate: exact zero, the residual is untouched
            b = Best{diag[c], (int32_t)c};
            tr = b.v;
        } else {
            double S = 0.0;
            for (int r = 0; r < k; ++r) S = __dadd_rn(S, __dmul_rn(L[(int64_t)r * n + c], Lp[r]));
            l = (kspec_value(spec, pts, pc, pp, px, py, pz, T) - S) / lpk;
            const double dc = __dadd_rn(diag[c], -__dmul_rn(l, l));
            diag[c] = dc;
            b = Best{dc, (int32_t)c};
            tr = dc;
        }
        L[(int64_t)k * n + c] = l;
    }
    block_best_sum(b, tr, shb, shs, lg);
    if (t == 0) {
        pmax[blockIdx.x] = b.v;
        pidx[blockIdx.x] = b.i;
        ptr[blockIdx.x] = tr;
    }
}

// G[a][b] = sum_c L[a][c] L[b][c]   (blocks on and above the diagonal computed, mirrored); grid (ceil(k / 4), ceil(k / 4)).
// A workgroup forms a 4 x 4 block of entries: per entry the arithmetic of the one-entry-per-workgroup kernel of rounds 1-5 (thread t
// adds the columns t, t + 256, ... by fma in ascending order, then the tree over the threads) -- the same bits -- with eight loads per
// sixteen multiply-adds instead of thirty-two (that kernel read 12 GB through the caches for the 171 columns of a rank-512 model at 50k points).
constexpr int kPcGramTile = 4;
__global__ __launch_bounds__(kPcBlock) void pc_gram_kernel(const double *__restrict__ L, int64_t M, int32_t k,
                                                           double *__restrict__ G) {
    __shared__ double shs[kPcGramTile * kPcGramTile][kPcBlock];
    const int a0 = blockIdx.x * kPcGramTile, b0 = blockIdx.y * kPcGramTile;
    if (a0 > b0) return;
    const double *la[kPcGramTile], *lb[kPcGramTile];
#pragma unroll
    for (int u = 0; u < kPcGramTile; ++u) {  // (columns past k: a clamped, valid column; the entries are not stored)
        la[u] = L + (int64_t)min(a0 + u, k - 1) * M;
        lb[u] = L + (int64_t)min(b0 + u, k - 1) * M;
    }
    double s[kPcGramTile][kPcGramTile];
#pragma unroll
    for (int u = 0; u < kPcGramTile; ++u)
#pragma unroll
        for (int v = 0; v < kPcGramTile; ++v) s[u][v] = 0.0;
    for (int64_t c = threadIdx.x; c < M; c += kPcBlock) {
        double va[kPcGramTile], vb[kPcGramTile];
#pragma unroll
        for (int u = 0; u < kPcGramTile; ++u) va[u] = la[u][c], vb[u] = lb[u][c];
#pragma unroll
        for (int u = 0; u < kPcGramTile; ++u)
#pragma unroll
            for (int v = 0; v < kPcGramTile; ++v) s[u][v] = __builtin_fma(va[u], vb[v], s[u][v]);
    }
#pragma unroll
    for (int u = 0; u < kPcGramTile; ++u)
#pragma unroll
        for (int v = 0; v < kPcGramTile; ++v) shs[u * kPcGramTile + v][threadIdx.x] = s[u][v];
    __syncthreads();
    for (int off = kPcBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int e = 0; e < kPcGramTile * kPcGramTile; ++e) shs[e][threadIdx.x] += shs[e][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < kPcGramTile * kPcGramTile) {
        const int a = a0 + (int)threadIdx.x / kPcGramTile, b = b0 + (int)threadIdx.x % kPcGramTile;
        if (a < k && b < k && a <= b) {  // (below the diagonal of a diagonal block: the mirrored entry's own sum, the same number)
            G[(int64_t)a * k + b] = shs[threadIdx.x][0];
            G[(int64_t)b * k + a] = shs[threadIdx.x][0];
        }
    }
}

// Cyclic Jacobi eigen-decomposition of the leading n x n block of G (row stride ldg), one workgroup, matrices in global
// memory (L2 resident).  Round-robin ordering: n' - 1 rounds of n'/2 disjoint rotations; a round is applied as
// A <- A J (columns), then A <- J^T A (rows), V <- V J.  Output: evals descending, Vs[:, rank] the matching eigenvectors.
constexpr int kJacThreads = 1024;
constexpr int kJacMaxN = 512;

__global__ __launch_bounds__(kJacThreads) void jacobi_eig_kernel(const double *__restrict__ G, int32_t ldg, int32_t n,
                                                                 double *__restrict__ A, double *__restrict__ V,
                                                                 double *__restrict__ evals, double *__restrict__ Vs,
                                                                 int32_t *__restrict__ sweeps_out) {
    __shared__ double cs[kJacMaxN / 2][2];
    __shared__ int32_t pq[kJacMaxN / 2][2];
    __shared__ int32_t nrot;
    const int t = threadIdx.x;
    for (int idx = t; idx < n * n; idx += kJacThreads) {
        const int i = idx / n, j = idx - i * n;
        A[idx] = G[(int64_t)i * ldg + j];
        V[idx] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    const int np = (n + 1) & ~1;  // even number of players; index np-1 >= n is a bye
    const int half = np / 2;
    int sweep = 0;
    for (; sweep < 60 && n > 1; ++sweep) {
        if (t == 0) nrot = 0;
        __syncthreads();
        for (int rd = 0; rd < np - 1; ++rd) {
            if (t < half) {
                int a = t == 0 ? np - 1 : (rd + t) % (np - 1);
                int b = (rd + np - 1 - t) % (np - 1);
                if (a > b) {
                    const int tmp = a;
                    a = b;
                    b = tmp;
                }
                double c = 1.0, s = 0.0;
                if (b < n) {
                    const double app = A[a * n + a], aqq = A[b * n + b], apq = A[a * n + b];
                    if (fabs(apq) > 1.1102230246251565e-16 * sqrt(fabs(app) * fabs(aqq)) && apq != 0.0) {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(tt * tt + 1.0);
                        s = tt * c;
                        atomicAdd(&nrot, 1);
                    }
                } else {
                    b = -1;
                }
                cs[t][0] = c;
                cs[t][1] = s;
                pq[t][0] = a;
                pq[t][1] = b;
            }
            __syncthreads();
            // columns of A and V
            for (int idx = t; idx < half * n; idx += kJacThreads) {
                const int i = idx / half, m = idx - i * half;  // neighbouring lanes: one row, different pairs
                const int p = pq[m][0], q = pq[m][1];
                const double c = cs[m][0], s = cs[m][1];
                if (q < 0 || s == 0.0) continue;
                const double aip = A[i * n + p], aiq = A[i * n + q];
                A[i * n + p] = c * aip - s * aiq;
                A[i * n + q] = s * aip + c * aiq;
                const double vip = V[i * n + p], viq = V[i * n + q];
                V[i * n + p] = c * vip - s * viq;
                V[i * n + q] = s * vip + c * viq;
            }
            __syncthreads();
            // rows of A
            for (int idx = t; idx < half * n; idx += kJacThreads) {
                const int m = idx / n, j = idx - m * n;
                const int p = pq[m][0], q = pq[m][1];
                const double c = cs[m][0], s = cs[m][1];
                if (q < 0 || s == 0.0) continue;
                const double apj = A[p * n + j], aqj = A[q * n + j];
                A[p * n + j] = c * apj - s * aqj;
                A[q * n + j] = s * apj + c * aqj;
            }
            __syncthreads();
        }
        const int done = nrot == 0;
        __syncthreads();
        if (done) break;
    }
    if (t == 0 && sweeps_out) *sweeps_out = sweep;
    // sort descending (rank sort; ties keep index order)
    for (int i = t; i < n; i += kJacThreads) {
        const double li = A[i * n + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double lj = A[j * n + j];
            rank += (lj > li || (lj == li && j < i)) ? 1 : 0;
        }
        evals[rank] = li;
        for (int r = 0; r < n; ++r) Vs[r * n + rank] = V[r * n + i];
    }
}

// The same decomposition on several workgroups (n > 128: beyond the register kernel of eig.hip the one-workgroup form above moves
// 3 n^2 entries per round through ONE compute unit's path to L2 -- 150 ms at n = 301).  Every workgroup works out all rotations of a
// round itself (the same numbers from the same entries), then forms its share of A' = J^T (A J) entry by entry from the round's
// source buffer into the other one (an entry needs the four entries at (i | partner of i, j | partner of j); first the column
// combination, then the row combination: the one-workgroup kernel's arithmetic in its order, so the same bits) and rotates its share
// of V in place (row-local).  One barrier over the grid per round instead of three workgroup barriers; the grid is far smaller than
// the device (64 workgroups of 256 threads) so that all of it is resident.  A barrier that is not met within two seconds (a device
// shared with something that never yields) raises the flag in sync[1]: the call reports a failure instead of hanging.
constexpr int kJacGridThreads = 256;
constexpr int kJacGridWgs = 64;

// One wave's thread does the device-scope part for its workgroup (as a cooperative-groups grid barrier does): the workgroup barrier
// in front has every wave's stores completed, the release fence then writes the compute unit's and the XCD's dirty lines back ONCE,
// and the acquire fence behind the wait drops the stale lines for the whole compute unit -- with the two fences in every thread the
// cache maintenance ran eight times per workgroup and round (22 us a round at n = 301 against 12).
__device__ __forceinline__ bool jac_grid_barrier(unsigned *sync, unsigned target, int *s_ok) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();  // (100 MHz)
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0u) {
                if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    good = 0;
                    break;
                }
                if (wall_clock64() - t0 > 200000000LL) {
                    __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_ok = good;
    }
    __syncthreads();
    return *s_ok != 0;
}

__global__ __launch_bounds__(kJacGridThreads) void jacobi_eig_grid_kernel(const double *__restrict__ G, int32_t ldg, int32_t n, double *A0,
                                                                          double *A1, double *V, double *__restrict__ evals,
                                                                          double *__restrict__ Vs, int32_t *__restrict__ sweeps_out,
                                                                          unsigned *sync) {
    __shared__ double cs[kJacMaxN / 2][2];
    __shared__ int32_t pq[kJacMaxN / 2][2];
    __shared__ int16_t pair_of[kJacMaxN];  // index -> its pair of the round, -1: none (the bye, or a pair that does not rotate)
    __shared__ int32_t nrot;
    __shared__ int s_ok;
    const int t = threadIdx.x;
    const unsigned nwg = gridDim.x;
    const unsigned gtid = blockIdx.x * kJacGridThreads + t, gthreads = nwg * kJacGridThreads;  // (n <= 512: 32-bit index arithmetic)
    const unsigned nn = (unsigned)n * (unsigned)n, un = (unsigned)n;
    unsigned epoch = 0;
    for (unsigned idx = gtid; idx < nn; idx += gthreads) {
        const int i = (int)(idx / un), j = (int)(idx - (unsigned)i * un);
        A0[idx] = G[(int64_t)i * ldg + j];
        V[idx] = i == j ? 1.0 : 0.0;
    }
    if (!jac_grid_barrier(sync, ++epoch * nwg, &s_ok)) return;
    double *A = A0, *B = A1;  // source and destination of the round
    const int np = (n + 1) & ~1;
    const int half = np / 2;
    int sweep = 0;
    for (; sweep < 60 && n > 1; ++sweep) {
        if (t == 0) nrot = 0;
        __syncthreads();
        for (int rd = 0; rd < np - 1; ++rd) {
            for (int i = t; i < n; i += kJacGridThreads) pair_of[i] = -1;
            __syncthreads();
            if (t < half) {  // (the one-workgroup kernel's pairing and rotation, statement by statement)
                int a = t == 0 ? np - 1 : (rd + t) % (np - 1);
                int b = (rd + np - 1 - t) % (np - 1);
                if (a > b) {
                    const int tmp = a;
                    a = b;
                    b = tmp;
                }
                double c = 1.0, s = 0.0;
                if (b < n) {
                    const double app = A[a * n + a], aqq = A[b * n + b], apq = A[a * n + b];
                    if (fabs(apq) > 1.1102230246251565e-16 * sqrt(fabs(app) * fabs(aqq)) && apq != 0.0) {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(tt * tt + 1.0);
                        s = tt * c;
                        atomicAdd(&nrot, 1);
                    }
                } else {
                    b = -1;
                }
                cs[t][0] = c;
                cs[t][1] = s;
                pq[t][0] = a;
                pq[t][1] = b;
                if (b >= 0 && s != 0.0) {
                    pair_of[a] = (int16_t)t;
                    pair_of[b] = (int16_t)t;
                }
            }
            __syncthreads();
            // A' = J^T (A J), entry by entry
            for (unsigned idx = gtid; idx < nn; idx += gthreads) {
                const int i = (int)(idx / un), j = (int)(idx - (unsigned)i * un);
                const int mi = pair_of[i], mj = pair_of[j];
                // the column combination at row r: (A J)[r][j]
                auto col = [&](int r) -> double {
                    if (mj < 0) return A[r * n + j];
                    const int p = pq[mj][0], q = pq[mj][1];
                    const double c = cs[mj][0], sn = cs[mj][1];
                    const double arp = A[r * n + p], arq = A[r * n + q];
                    return j == p ? c * arp - sn * arq : sn * arp + c * arq;
                };
                double out;
                if (mi < 0) {
                    out = col(i);
                } else {
                    const int p = pq[mi][0], q = pq[mi][1];
                    const double c = cs[mi][0], sn = cs[mi][1];
                    const double apj = col(p), aqj = col(q);
                    out = i == p ? c * apj - sn * aqj : sn * apj + c * aqj;
                }
                B[idx] = out;
            }
            // V <- V J, in place (an item touches the two entries of its row and pair only)
            for (unsigned idx = gtid; idx < (unsigned)half * un; idx += gthreads) {
                const int i = (int)(idx / (unsigned)half), m = (int)(idx - (unsigned)i * (unsigned)half);
                const int p = pq[m][0], q = pq[m][1];
                const double c = cs[m][0], sn = cs[m][1];
                if (q < 0 || sn == 0.0) continue;
                const double vip = V[i * n + p], viq = V[i * n + q];
                V[i * n + p] = c * vip - sn * viq;
                V[i * n + q] = sn * vip + c * viq;
            }
            if (!jac_grid_barrier(sync, ++epoch * nwg, &s_ok)) return;
            double *tmp = A;
            A = B;
            B = tmp;
        }
        const int done = nrot == 0;
        __syncthreads();
        if (done) break;
    }
    if (gtid == 0 && sweeps_out) *sweeps_out = sweep;
    // sort descending (rank sort; ties keep index order)
    for (int i = (int)gtid; i < n; i += (int)gthreads) {
        const double li = A[i * n + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double lj = A[j * n + j];
            rank += (lj > li || (lj == li && j < i)) ? 1 : 0;
        }
        evals[rank] = li;
        for (int r = 0; r < n; ++r) Vs[r * n + rank] = V[r * n + i];
    }
}

// B[i][c] = sum_r L[r][c] V[r][i]   (c over ALL points, i < n): column i of  U sqrt(lambda) = L V
constexpr int kLvCols = 8;  // columns of B per thread: L is read n / 8 times instead of n times (the same fma sequence per entry)
__global__ __launch_bounds__(256) void lv_kernel(const double *__restrict__ L, int64_t M, int32_t n,
                                                 const double *__restrict__ V, double *__restrict__ B) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * kLvCols;
    if (c >= M) return;
    int col[kLvCols];  // workgroup-uniform; columns past n repeat the last one and are not stored
#pragma unroll
    for (int u = 0; u < kLvCols; ++u) col[u] = min(i0 + u, n - 1);
    double acc[kLvCols];
#pragma unroll
    for (int u = 0; u < kLvCols; ++u) acc[u] = 0.0;
    for (int r = 0; r < n; ++r) {
        const double l = L[(int64_t)r * M + c];
        const double *v = V + (int64_t)r * n;
#pragma unroll
        for (int u = 0; u < kLvCols; ++u) acc[u] = __builtin_fma(l, v[col[u]], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < kLvCols; ++u)
        if (i0 + u < n) B[(int64_t)(i0 + u) * M + c] = acc[u];
}

// Q0[(3s+d)*rp + q] = B_set(q)[idx(q)][row_begin + perm[s]] when coordinate(q) == d, else 0
__global__ __launch_bounds__(256) void gpmm_pack_kernel(const double *__restrict__ BA, const double *__restrict__ BB,
                                                        const double *__restrict__ BC, int64_t M_total, int64_t row_begin, int64_t M, int32_t r, int32_t rp,
                                                        const int32_t *__restrict__ perm, const int32_t *__restrict__ qdim,
                                                        const int32_t *__restrict__ qset, const int32_t *__restrict__ qidx,
                                                        double *__restrict__ Q0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * M * rp) return;
    const int64_t row = idx / rp;
    const int32_t q = (int32_t)(idx - row * rp);
    const int64_t s = row / 3;
    const int d = (int)(row - 3 * s);
    double v = 0.0;
    if (q < r && qdim[q] == d) {
        const int64_t c = row_begin + (perm ? perm[s] : s);
        v = (qset[q] == 0 ? BA : (qset[q] == 1 ? BB : BC))[(int64_t)qidx[q] * M_total + c];
    }
    Q0[idx] = v;
}

// stage[k*3M + 3*perm[s] + d] = Q0[(3s+d)*rp + k] / sqrt(variance[k])   (inverse of pack_basis_kernel)
__global__ __launch_bounds__(256) void unpack_basis_kernel(const double *__restrict__ Q0, const double *__restrict__ variance,
                                                           int64_t M, int32_t r, int32_t rp,
                                                           const int32_t *__restrict__ perm, double *__restrict__ stage) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * M * r) return;
    const int64_t row = idx / r;
    const int32_t k = (int32_t)(idx - row * r);
    const int64_t s = row / 3, d = row - 3 * s;
    const double sd = sqrt(variance[k]);
    stage[(int64_t)k * 3 * M + 3 * (int64_t)(perm ? perm[s] : s) + d] = sd > 0.0 ? Q0[row * rp + k] / sd : 0.0;
}

// all-pairs extrema of |p_i - p_j|^2 (unfused), i != j: per-workgroup partials.  The distance of a pair is the same number in both
// orders (the coordinate differences are exact negations of each other, their squares equal), so only the tile pairs on and above the
// diagonal are visited: workgroup (bi, c) takes the 256 points of tile bi against the column tiles bi + c ch ... bi + (c + 1) ch - 1;
// workgroups past the last tile write the neutral pair (0, +inf).  A maximum and a minimum do not depend on the order of the scan: the
// result is the one-row-per-thread scan's of rounds 1-5 bit for bit (that one ran on ceil(n / 256) workgroups -- 196 for 50k points,
// fewer than the device has compute units -- and visited every pair twice: 2.5 ms at 50k).  The `j != i` test only exists in the
// diagonal tile.
constexpr int kExtTile = 256;
__global__ __launch_bounds__(kExtTile) void dist_extrema_kernel(Cloud pts, int ch, double *__restrict__ pmax, double *__restrict__ pmin) {
    __shared__ double tx[kExtTile], ty[kExtTile], tz[kExtTile];
    __shared__ double sh[kExtTile];
    const int t = threadIdx.x;
    const int64_t nt = (pts.n + kExtTile - 1) / kExtTile;
    const int64_t bi = blockIdx.x;
    const int64_t jt0 = bi + (int64_t)blockIdx.y * ch, jt1 = min(nt, jt0 + ch);  // workgroup-uniform
    const int64_t i = bi * kExtTile + t;
    const bool ok = i < pts.n;
    const double x = ok ? pts.x[i] : 0.0, y = ok ? pts.y[i] : 0.0, z = ok ? pts.z[i] : 0.0;
    double mx = 0.0, mn = __builtin_huge_val();
    for (int64_t jt = jt0; jt < jt1; ++jt) {
        const int64_t jb = jt * kExtTile;
        __syncthreads();
        if (jb + t < pts.n) {
            tx[t] = pts.x[jb + t];
            ty[t] = pts.y[jb + t];
            tz[t] = pts.z[jb + t];
        }
        __syncthreads();
        const int cnt = (int)min((int64_t)kExtTile, pts.n - jb);
        if (!ok) continue;
        if (jt == bi) {  // the diagonal tile: every pair of it from both sides, the point itself left out of the minimum
#pragma unroll 4
            for (int jj = 0; jj < cnt; ++jj) {
                const double dx = x - tx[jj], dy = y - ty[jj], dz = z - tz[jj];
                const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                mx = fmax(mx, d2);
                if (jj != t) mn = fmin(mn, d2);
            }
        } else {
#pragma unroll 4
            for (int jj = 0; jj < cnt; ++jj) {
                const double dx = x - tx[jj], dy = y - ty[jj], dz = z - tz[jj];
                const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                mx = fmax(mx, d2);
                mn = fmin(mn, d2);
            }
        }
    }
    const int64_t slot = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    __syncthreads();
    sh[t] = mx;
    __syncthreads();
    for (int off = kExtTile / 2; off > 0; off >>= 1) {
        if (t < off) sh[t] = fmax(sh[t], sh[t + off]);
        __syncthreads();
    }
    if (t == 0) pmax[slot] = sh[0];
    __syncthreads();
    sh[t] = mn;
    __syncthreads();
    for (int off = kExtTile / 2; off > 0; off >>= 1) {
        if (t < off) sh[t] = fmin(sh[t], sh[t + off]);
        __syncthreads();
    }
    if (t == 0) pmin[slot] = sh[0];
}

int check(gingr_ctx *ctx) {
    HIP_TRY(ctx, hipGetLastError());
    return GINGR_OK;
}

}  // namespace

// One decomposition with the two-sided kernel (any n <= 512; the fallback of sym_eig, and its path above kSymEigColsMaxN)
static int jacobi_two_sided(gingr_ctx *ctx, const double *G, int32_t ldg, int32_t n, double *evals, double *Vs, int32_t *sweeps) {
    DevBuf wA, wV;
    HIP_TRY(ctx, wA.alloc((size_t)n * n * sizeof(double)));
    HIP_TRY(ctx, wV.alloc((size_t)n * n * sizeof(double)));
    if (n > 128) {  // the same arithmetic on 64 workgroups (a second buffer for A, a barrier word and a failure flag)
        DevBuf wB, sync;
        HIP_TRY(ctx, wB.alloc((size_t)n * n * sizeof(double)));
        HIP_TRY(ctx, sync.alloc(2 * sizeof(unsigned)));
        HIP_TRY(ctx, hipMemsetAsync(sync.p, 0, 2 * sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(jacobi_eig_grid_kernel, dim3(kJacGridWgs), dim3(kJacGridThreads), 0, ctx->stream, G, ldg, n, wA.as<double>(),
                           wB.as<double>(), wV.as<double>(), evals, Vs, sweeps, sync.as<unsigned>());
        HIP_TRY(ctx, hipGetLastError());
        unsigned h[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(h, sync.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the work buffers go out of scope
        if (h[1])
            return gingr_set_error(ctx, GINGR_ERR_STATE, "sym_eig: the workgroups of the decomposition did not meet within two seconds (n = %d)", (int)n);
        return GINGR_OK;
    }
    hipLaunchKernelGGL(jacobi_eig_kernel, dim3(1), dim3(kJacThreads), 0, ctx->stream, G, ldg, n, wA.as<double>(), wV.as<double>(), evals, Vs,
                       sweeps);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the work buffers go out of scope
    return GINGR_OK;
}

// Eigen-decompositions of up to three symmetric positive semi-definite matrices (leading n[q] x n[q] block of G[q], row stride ldg[q]):
// evals[q] descending, Vs[q][:, k] the matching eigenvectors.  Up to kSymEigColsMaxN the register kernel of eig.hip, all problems in
// one launch; a problem it reports as numerically singular or not converged -- and anything larger -- goes through the two-sided
// kernel.  Synchronises the stream.  GINGR_ERR_NONFINITE when a decomposition does not converge.
static int sym_eig(gingr_ctx *ctx, int count, const double *const *G, const int32_t *ldg, const int32_t *n, double *const *evals,
            double *const *Vs) {
    if (count < 1 || count > 3) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "sym_eig: 1..3 problems");
    bool redo[3] = {false, false, false};
    DevBuf work[3], info;
    HIP_TRY(ctx, info.alloc(3 * 2 * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemsetAsync(info.p, 0, 3 * 2 * sizeof(int32_t), ctx->stream));
    const double *fG[3];
    int32_t fld[3], fn[3];
    double *fw[3], *fe[3], *fv[3];
    int32_t *fi[3];
    int fast = 0, fast_of[3];
    for (int q = 0; q < count; ++q) {
        if (n[q] < 1 || n[q] > kJacMaxN) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "sym_eig: n = %d outside 1..%d", (int)n[q], kJacMaxN);
        if (n[q] > kSymEigColsMaxN) {
            redo[q] = true;
            continue;
        }
        HIP_TRY(ctx, work[q].alloc((size_t)sym_eig_cols_work_doubles(n[q]) * sizeof(double)));
        fG[fast] = G[q], fld[fast] = ldg[q], fn[fast] = n[q], fw[fast] = work[q].as<double>(), fe[fast] = evals[q], fv[fast] = Vs[q];
        fi[fast] = info.as<int32_t>() + 2 * q;
        fast_of[fast++] = q;
    }
    if (fast) {
        launch_sym_eig_cols(ctx, fast, fG, fld, fn, fw, fe, fv, fi);
        HIP_TRY(ctx, hipGetLastError());
        int32_t h[6];
        HIP_TRY(ctx, hipMemcpyAsync(h, info.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int f = 0; f < fast; ++f) {
            const int q = fast_of[f];
            if (h[2 * q] >= 60 || h[2 * q + 1]) redo[q] = true;
        }
    }
    for (int q = 0; q < count; ++q) {
        if (!redo[q]) continue;
        GINGR_TRY(jacobi_two_sided(ctx, G[q], ldg[q], n[q], evals[q], Vs[q], info.as<int32_t>() + 2 * q));
        int32_t sw = 0;
        HIP_TRY(ctx, hipMemcpy(&sw, info.as<int32_t>() + 2 * q, sizeof(sw), hipMemcpyDeviceToHost));
        if (sw >= 60) return gingr_set_error(ctx, GINGR_ERR_NONFINITE, "sym_eig: Jacobi did not converge (n = %d)", (int)n[q]);
    }
    return GINGR_OK;
}

int launch_jacobi_eig(gingr_ctx *ctx, const double *G, int32_t ldg, int32_t n, double *evals, double *Vs) {
    const double *Gs[1] = {G};
    double *es[1] = {evals}, *vs[1] = {Vs};
    return sym_eig(ctx, 1, Gs, &ldg, &n, es, vs);
}

extern "C" {

int gingr_pointset_distance_extrema(gingr_ctx *ctx, const double *xyz, int64_t n, double *max_distance,
                                    double *min_distance) {
    if (!ctx || !xyz || n < 1 || !max_distance || !min_distance)
        return ctx ? gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "distance_extrema: bad argument") : GINGR_ERR_BAD_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // tiles of 256 points; a workgroup takes `ch` column tiles (eight, more for large clouds so that there are at most 64 chunks per tile
    // row): about nt^2 / (2 ch) working workgroups -- 2 500 at 50k points
    const int64_t nt = ceil_div(n, (int64_t)kExtTile);
    const int ch = (int)std::max<int64_t>(8, ceil_div(nt, (int64_t)64));
    const int gy = (int)ceil_div(nt, (int64_t)ch);
    if (nt > 0x7fffffff / 64) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "distance_extrema: %lld points are too many", (long long)n);
    const int64_t nb = nt * gy;
    DevBuf aos, soa, pm;
    HIP_TRY(ctx, aos.alloc((size_t)3 * n * sizeof(double)));
    HIP_TRY(ctx, soa.alloc((size_t)3 * n * sizeof(double)));
    HIP_TRY(ctx, pm.alloc((size_t)2 * nb * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(aos.p, xyz, (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, aos.as<double>(), n, soa.as<double>());
    const double *s = soa.as<double>();
    hipLaunchKernelGGL(dist_extrema_kernel, dim3((unsigned)nt, (unsigned)gy), dim3(kExtTile), 0, ctx->stream, Cloud{s, s + n, s + 2 * n, n}, ch,
                       pm.as<double>(), pm.as<double>() + nb);
    GINGR_TRY(check(ctx));
    std::vector<double> h((size_t)2 * nb);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), pm.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double mx = 0.0, mn = HUGE_VAL;
    for (int64_t b = 0; b < nb; ++b) {
        mx = std::max(mx, h[(size_t)b]);
        mn = std::min(mn, h[(size_t)nb + b]);
    }
    *max_distance = std::sqrt(mx);   // max of sqrt = sqrt of max (sqrt is monotone and correctly rounded)
    *min_distance = std::sqrt(mn);   // +inf for a single point
    return GINGR_OK;
}

// ---- host side of the construction --------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

// The pivoted Cholesky factorisation on the device: over the M points for one scalar kernel, or (gen) over the 3M (point,
// coordinate) entries of a DiagonalKernel whose coordinates have different kernels.
struct Factor {
    gingr_ctx *ctx = nullptr;
    KSpec3 specs{};
    Cloud pts{};
    bool gen = false;
    int64_t n = 0;     // entries
    int nb = 0;
    int32_t kcap = 0;  // columns the buffers can hold
    int32_t ks = 0;    // columns computed
    DevBuf Lb, diag, piv, part, ctl, trace, pivots, swd, swo;
    std::vector<double> htrace;
    double *pmaxv[2]{}, *ptrv[2]{};
    int32_t *pidxv[2]{};

    int init(gingr_ctx *c, const KSpec3 &k, const Cloud &p, bool generic, int32_t cap) {
        ctx = c, specs = k, pts = p, gen = generic, n = generic ? 3 * p.n : p.n, kcap = cap;
        nb = (int)ceil_div(n, kPcBlock);
        HIP_TRY(ctx, Lb.alloc((size_t)kcap * n * sizeof(double)));
        HIP_TRY(ctx, diag.alloc((size_t)n * sizeof(double)));
        HIP_TRY(ctx, piv.alloc((size_t)n * sizeof(int32_t)));
        HIP_TRY(ctx, part.alloc((size_t)2 * nb * (2 * sizeof(double) + sizeof(int32_t)) + 64));
        HIP_TRY(ctx, ctl.alloc(2 * sizeof(int32_t)));
        HIP_TRY(ctx, trace.alloc((size_t)(kcap + 1) * sizeof(double)));
        HIP_TRY(ctx, pivots.alloc((size_t)(kcap + 1) * sizeof(int32_t)));
        HIP_TRY(ctx, swd.alloc((size_t)(kcap + 1) * sizeof(int32_t)));
        HIP_TRY(ctx, swo.alloc((size_t)(kcap + 1) * sizeof(int32_t)));
        pmaxv[0] = part.as<double>(), pmaxv[1] = part.as<double>() + nb;
        ptrv[0] = part.as<double>() + 2 * nb, ptrv[1] = part.as<double>() + 3 * nb;
        pidxv[0] = reinterpret_cast<int32_t *>(part.as<double>() + 4 * nb), pidxv[1] = pidxv[0] + nb;
        if (gen)
            hipLaunchKernelGGL(pc_init_kernel<true>, dim3(nb), dim3(kPcBlock), 0, ctx->stream, pts, specs, diag.as<double>(),
                               piv.as<int32_t>(), pmaxv[0], pidxv[0], ptrv[0], ctl.as<int32_t>());
        else
            hipLaunchKernelGGL(pc_init_kernel<false>, dim3(nb), dim3(kPcBlock), 0, ctx->stream, pts, specs, diag.as<double>(),
                               piv.as<int32_t>(), pmaxv[0], pidxv[0], ptrv[0], ctl.as<int32_t>());
        return check(ctx);
    }

    void launch(int32_t k, double rel_tol) {
        const int in = k & 1, o = in ^ 1;
        if (gen)
            hipLaunchKernelGGL(pc_step_kernel<true>, dim3(nb), dim3(kPcBlock), 0, ctx->stream, pts, specs, k, kcap, rel_tol, nb,
                               Lb.as<double>(), diag.as<double>(), piv.as<int32_t>(), pmaxv[in], pidxv[in], ptrv[in], pmaxv[o], pidxv[o],
                               ptrv[o], ctl.as<int32_t>(), trace.as<double>(), pivots.as<int32_t>(), swd.as<int32_t>(), swo.as<int32_t>());
        else
            hipLaunchKernelGGL(pc_step_kernel<false>, dim3(nb), dim3(kPcBlock), 0, ctx->stream, pts, specs, k, kcap, rel_tol, nb,
                               Lb.as<double>(), diag.as<double>(), piv.as<int32_t>(), pmaxv[in], pidxv[in], ptrv[in], pmaxv[o], pidxv[o],
                               ptrv[o], ctl.as<int32_t>(), trace.as<double>(), pivots.as<int32_t>(), swd.as<int32_t>(), swo.as<int32_t>());
    }

    // the whole factorisation under the device's stopping rule (residual trace below rel_tol * trace, capacity, exhaustion)
    int run_to_tolerance(double rel_tol) {
        int32_t hctl[2] = {0, 0};
        for (int32_t k = 0; k <= kcap; ++k) {
            launch(k, rel_tol);
            if ((k & 15) == 15 || k == kcap) {  // look at the stop flag now and then instead of queueing no-op launches
                GINGR_TRY(check(ctx));
                HIP_TRY(ctx, hipMemcpyAsync(hctl, ctl.p, sizeof(hctl), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (hctl[1]) break;
            }
        }
        if (!hctl[1]) return gingr_set_error(ctx, GINGR_ERR_STATE, "gpmm_build: pivoted Cholesky did not terminate");
        ks = hctl[0];
        htrace.resize((size_t)ks + 1);
        HIP_TRY(ctx, hipMemcpyAsync(htrace.data(), trace.p, htrace.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return GINGR_OK;
    }
};

// Q0[(3s+d)*rp + q] = B[q][3 (row_begin + perm[s]) + d]: the generic factor's columns are already (point, coordinate) interleaved
__global__ __launch_bounds__(256) void gpmm_pack_generic_kernel(const double *__restrict__ B, int64_t M_total, int64_t row_begin,
                                                                int64_t M, int32_t r, int32_t rp, const int32_t *__restrict__ perm,
                                                                double *__restrict__ Q0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * M * rp) return;
    const int64_t row = idx / rp;
    const int32_t q = (int32_t)(idx - row * rp);
    const int64_t s = row / 3;
    const int d = (int)(row - 3 * s);
    Q0[idx] = q < r ? B[(int64_t)q * 3 * M_total + 3 * (row_begin + (perm ? perm[s] : s)) + d] : 0.0;
}

bool same_kernel(const gingr_scalar_kernel *a, const gingr_scalar_kernel *b) {
    if (a == b) return true;
    if (a->kind != b->kind) return false;
    if (a->kind == GINGR_KERNEL_GAUSSIAN_MIXTURE) {
        if (a->n_kernels != b->n_kernels || a->mirror != b->mirror) return false;
        for (int i = 0; i < a->n_kernels; ++i)
            if (a->sigmas[i] != b->sigmas[i] || a->scalings[i] != b->scalings[i]) return false;
        return true;
    }
    return a->scaling == b->scaling && a->lookup == b->lookup;
}

}  // namespace

extern "C" {

int gingr_gpmm_build_diagonal(gingr_ctx *ctx, int64_t M_total, const double *ref, const gingr_scalar_kernel *kx,
                              const gingr_scalar_kernel *ky, const gingr_scalar_kernel *kz, double relative_tolerance,
                              int32_t max_rank, int64_t row_begin, int64_t row_end, gingr_model **out) {
    if (!ctx || !out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (M_total < 1 || !ref || !kx || !ky || !kz) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: need M >= 1, a reference and three kernels");
    if (!(relative_tolerance >= 0.0) || !(relative_tolerance < 1.0))
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: relative tolerance must be in [0, 1)");
    const gingr_scalar_kernel *kd[3] = {kx, ky, kz};
    // distinct kernels -> factor sets
    int dim_set[3], nsets = 0;
    const gingr_scalar_kernel *set_k[3];
    for (int d = 0; d < 3; ++d) {
        int s = -1;
        for (int q = 0; q < nsets; ++q)
            if (same_kernel(set_k[q], kd[d])) s = q;
        if (s < 0) set_k[s = nsets++] = kd[d];
        dim_set[d] = s;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = M_total;
    DevBuf lookups[3];
    KSpec specs[3];
    for (int q = 0; q < nsets; ++q) {
        const gingr_scalar_kernel *k = set_k[q];
        KSpec &sp = specs[q];
        memset(&sp, 0, sizeof(sp));
        sp.kind = k->kind;
        if (k->kind == GINGR_KERNEL_GAUSSIAN_MIXTURE) {
            if (k->n_kernels < 1 || k->n_kernels > kMaxMix || !k->sigmas || !k->scalings)
                return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: need M >= 1 and 1..%d kernels", kMaxMix);
            if (!(k->mirror == 0.0 || k->mirror == 1.0 || k->mirror == -1.0))
                return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: mirror must be 0, +1 or -1");
            sp.mix.n = k->n_kernels;
            sp.mirror = k->mirror;
            for (int i = 0; i < k->n_kernels; ++i) {
                if (!(k->sigmas[i] > 0.0) || !(k->scalings[i] > 0.0) || !std::isfinite(k->sigmas[i]) || !std::isfinite(k->scalings[i]))
                    return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: sigma and scaling must be positive");
                sp.mix.c[i] = -(double)GINGR_EXP_TABLE * 1.4426950408889634074 / (k->sigmas[i] * k->sigmas[i]);
                sp.mix.s[i] = k->scalings[i];
            }
        } else if (k->kind == GINGR_KERNEL_DOT || k->kind == GINGR_KERNEL_LOOKUP) {
            if (!(k->scaling > 0.0) || !std::isfinite(k->scaling))
                return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: scaling must be positive");
            sp.scale = k->scaling;
            if (k->kind == GINGR_KERNEL_LOOKUP) {
                if (!k->lookup) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: lookup kernel without a matrix");
                HIP_TRY(ctx, lookups[q].alloc((size_t)M * M * sizeof(double)));
                HIP_TRY(ctx, hipMemcpyAsync(lookups[q].p, k->lookup, (size_t)M * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
                sp.lookup = lookups[q].as<double>();
            }
        } else {
            return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: unknown kernel kind %d", (int)k->kind);
        }
    }
    if (max_rank <= 0 || max_rank > 512) max_rank = 512;  // the model's rank limit (gingr_model_upload)
    if ((int64_t)max_rank > 3 * M_total) max_rank = (int32_t)(3 * M_total);

    DevBuf aos, soa;
    HIP_TRY(ctx, aos.alloc((size_t)3 * M * sizeof(double)));
    HIP_TRY(ctx, soa.alloc((size_t)3 * M * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(aos.p, ref, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, aos.as<double>(), M, soa.as<double>());
    const double *sp0 = soa.as<double>();
    const Cloud pts{sp0, sp0 + M, sp0 + 2 * M, M};

    if (row_end <= 0) row_end = M_total;
    std::vector<double> zero_mean((size_t)3 * M_total, 0.0);
    KSpec3 sp3;
    for (int d = 0; d < 3; ++d) sp3.k[d] = specs[dim_set[d]];
    if (nsets > 1) {
        // Coordinates with different kernels: scalismo's generic factorisation itself, over the 3M (point, coordinate) entries --
        // argmax of the residual diagonal in the permuted entry order, stop at relTol * trace -- then the eigen-decomposition of
        // L^T L and U sqrt(lambda) = L V, exactly as for any matrix-valued kernel.
        Factor fg;
        GINGR_TRY(fg.init(ctx, sp3, pts, true, (int32_t)std::min<int64_t>(3 * M, max_rank)));
        GINGR_TRY(fg.run_to_tolerance(relative_tolerance));
        const int32_t n = fg.ks;
        if (n < 1) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: empty model (tolerance too large)");
        DevBuf G, ev, V, B;
        HIP_TRY(ctx, G.alloc((size_t)n * n * sizeof(double)));
        HIP_TRY(ctx, ev.alloc((size_t)(n + 1) * sizeof(double)));
        HIP_TRY(ctx, V.alloc((size_t)(n * n + 1) * sizeof(double)));
        HIP_TRY(ctx, B.alloc((size_t)n * 3 * M * sizeof(double)));
        hipLaunchKernelGGL(pc_gram_kernel, dim3((unsigned)ceil_div(n, kPcGramTile), (unsigned)ceil_div(n, kPcGramTile)), dim3(kPcBlock), 0, ctx->stream, fg.Lb.as<double>(), 3 * M, n, G.as<double>());
        {
            const double *Gs[1] = {G.as<double>()};
            double *es[1] = {ev.as<double>()}, *vs[1] = {V.as<double>()};
            const int32_t ns[1] = {n};
            GINGR_TRY(sym_eig(ctx, 1, Gs, ns, ns, es, vs));
        }
        hipLaunchKernelGGL(lv_kernel, dim3((unsigned)ceil_div(3 * M, 256), (unsigned)ceil_div(n, kLvCols)), dim3(256), 0, ctx->stream, fg.Lb.as<double>(), 3 * M, n,
                           V.as<double>(), B.as<double>());
        GINGR_TRY(check(ctx));
        std::vector<double> hev((size_t)n);
        HIP_TRY(ctx, hipMemcpyAsync(hev.data(), ev.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (auto &v : hev) v = v > 0.0 ? v : 0.0;
        auto fill = [&](gingr_model *m) -> int {
            const int64_t total = 3 * m->M * m->rp;
            hipLaunchKernelGGL(gpmm_pack_generic_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, B.as<double>(),
                               M_total, m->row_begin, m->M, m->r, m->rp, m->perm, m->Q0);
            return check(ctx);
        };
        return model_create_impl(ctx, M_total, n, ref, zero_mean.data(), hev.data(), row_begin, row_end, fill, out);
    }

    // One kernel for all coordinates: the generic pivot sequence is (P0,x),(P0,y),(P0,z),(P1,x),... with P0,P1,.. the scalar
    // pivots (the three coordinates swap in triplets, so "first in the permuted order" among the entries of one coordinate is
    // the scalar factorisation's own permuted order); after n = 3j + e pivots the residual trace is (3-e) tr_s(j) + e tr_s(j+1).
    Factor fac[1];
    int32_t cnt[3] = {0, 0, 0};  // columns per coordinate
    int32_t n = 0;               // generic pivots = rank
    {
        const int32_t kmax = (int32_t)std::min<int64_t>(M, (max_rank + 2) / 3);  // scalar columns that can ever be needed
        GINGR_TRY(fac[0].init(ctx, sp3, pts, false, kmax));
        GINGR_TRY(fac[0].run_to_tolerance(relative_tolerance));
        const int32_t ks = fac[0].ks;
        if (ks < 1) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: empty model (tolerance too large)");
        const std::vector<double> &htr = fac[0].htrace;
        const double tol_g = relative_tolerance * (3.0 * htr[0]);
        const int32_t nmax = std::min<int32_t>(max_rank, 3 * ks);
        while (n < nmax) {
            const int32_t j = n / 3, e = n % 3;
            const double tr = e == 0 ? 3.0 * htr[(size_t)j] : (3 - e) * htr[(size_t)j] + e * htr[(size_t)j + 1];
            if (!(tr >= tol_g)) break;
            ++n;
        }
        for (int d = 0; d < 3; ++d) cnt[d] = n / 3 + (d < n % 3 ? 1 : 0);
    }
    if (n < 1) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: empty model (tolerance too large)");

    // blocks = distinct column counts among the coordinates; eigen-decomposition of L_s[:, :count]^T L_s[:, :count]
    int dim_block[3] = {-1, -1, -1}, nblocks = 0, block_set[3] = {0, 0, 0};
    int32_t block_n[3] = {0, 0, 0};
    for (int d = 0; d < 3; ++d) {
        if (cnt[d] == 0) continue;
        int b = -1;
        for (int q = 0; q < nblocks; ++q)
            if (block_n[q] == cnt[d]) b = q;
        if (b < 0) b = nblocks++, block_n[b] = cnt[d];
        dim_block[d] = b;
    }
    DevBuf G[3], ev[3], V[3], B[3];
    int32_t kkmax = 0;
    for (int b = 0; b < nblocks; ++b) kkmax = std::max(kkmax, block_n[b]);
    int32_t set_kk[3] = {kkmax, 0, 0};
    HIP_TRY(ctx, G[0].alloc((size_t)kkmax * kkmax * sizeof(double)));
    hipLaunchKernelGGL(pc_gram_kernel, dim3((unsigned)ceil_div(kkmax, kPcGramTile), (unsigned)ceil_div(kkmax, kPcGramTile)), dim3(kPcBlock), 0, ctx->stream, fac[0].Lb.as<double>(), M, kkmax,
                       G[0].as<double>());
    std::vector<double> hev[3];
    {
        const double *Gs[3];
        double *es[3], *vs[3];
        int32_t lds[3], ns[3];
        for (int b = 0; b < nblocks; ++b) {
            const int32_t nb_ = block_n[b];
            const int q = block_set[b];
            HIP_TRY(ctx, ev[b].alloc((size_t)(nb_ + 1) * sizeof(double)));
            HIP_TRY(ctx, V[b].alloc((size_t)(nb_ * nb_ + 1) * sizeof(double)));
            HIP_TRY(ctx, B[b].alloc(((size_t)nb_ * M + 1) * sizeof(double)));
            Gs[b] = G[q].as<double>(), lds[b] = set_kk[q], ns[b] = nb_, es[b] = ev[b].as<double>(), vs[b] = V[b].as<double>();
            hev[b].resize((size_t)nb_);
        }
        GINGR_TRY(sym_eig(ctx, nblocks, Gs, lds, ns, es, vs));  // the blocks side by side, one workgroup each
    }
    for (int b = 0; b < nblocks; ++b)
        hipLaunchKernelGGL(lv_kernel, dim3((unsigned)ceil_div(M, 256), (unsigned)ceil_div(block_n[b], kLvCols)), dim3(256), 0, ctx->stream,
                           fac[block_set[b]].Lb.as<double>(), M, block_n[b], V[b].as<double>(), B[b].as<double>());
    GINGR_TRY(check(ctx));
    for (int b = 0; b < nblocks; ++b)
        HIP_TRY(ctx, hipMemcpyAsync(hev[b].data(), ev[b].p, hev[b].size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

    // merged eigenpairs, descending; equal eigenvalues keep coordinate order (x, y, z)
    struct Col {
        double lam;
        int32_t dim, set, idx;
    };
    std::vector<Col> cols;
    for (int d = 0; d < 3; ++d)
        for (int32_t i = 0; i < cnt[d]; ++i) cols.push_back(Col{hev[dim_block[d]][(size_t)i], d, dim_block[d], i});
    std::stable_sort(cols.begin(), cols.end(), [](const Col &x, const Col &y) {
        if (x.lam != y.lam) return x.lam > y.lam;
        if (x.idx != y.idx) return x.idx < y.idx;
        return x.dim < y.dim;
    });
    const int32_t rank = (int32_t)cols.size();  // == n
    std::vector<double> variance((size_t)rank);
    std::vector<int32_t> qmap((size_t)3 * rank);
    for (int32_t q = 0; q < rank; ++q) {
        variance[(size_t)q] = cols[(size_t)q].lam > 0.0 ? cols[(size_t)q].lam : 0.0;
        qmap[(size_t)q] = cols[(size_t)q].dim;
        qmap[(size_t)rank + q] = cols[(size_t)q].set;
        qmap[(size_t)2 * rank + q] = cols[(size_t)q].idx;
    }
    DevBuf dq;
    HIP_TRY(ctx, dq.alloc(qmap.size() * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpy(dq.p, qmap.data(), qmap.size() * sizeof(int32_t), hipMemcpyHostToDevice));

    auto fill = [&](gingr_model *m) -> int {
        const int64_t total = 3 * m->M * m->rp;
        hipLaunchKernelGGL(gpmm_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, B[0].as<double>(),
                           B[1].as<double>(), B[2].as<double>(), M_total, m->row_begin, m->M, m->r, m->rp, m->perm, dq.as<int32_t>(),
                           dq.as<int32_t>() + rank, dq.as<int32_t>() + 2 * rank, m->Q0);
        return check(ctx);
    };
    return model_create_impl(ctx, M_total, rank, ref, zero_mean.data(), variance.data(), row_begin, row_end, fill, out);
}

int gingr_gpmm_build_gaussian(gingr_ctx *ctx, int64_t M_total, const double *ref, int32_t n_kernels, const double *sigmas,
                              const double *scalings, double relative_tolerance, int32_t max_rank, int64_t row_begin,
                              int64_t row_end, gingr_model **out) {
    if (!ctx || !out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (M_total < 1 || !ref || n_kernels < 1 || n_kernels > kMaxMix || !sigmas || !scalings)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: need M >= 1 and 1..%d kernels", kMaxMix);
    gingr_scalar_kernel k;
    memset(&k, 0, sizeof(k));
    k.kind = GINGR_KERNEL_GAUSSIAN_MIXTURE, k.n_kernels = n_kernels, k.sigmas = sigmas, k.scalings = scalings;
    return gingr_gpmm_build_diagonal(ctx, M_total, ref, &k, &k, &k, relative_tolerance, max_rank, row_begin, row_end, out);
}

int gingr_model_download(gingr_ctx *ctx, const gingr_model *m, double *ref, double *mean, double *basis_colmajor,
                         double *variance) {
    if (!ctx || !m) return GINGR_ERR_BAD_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = m->M;
    DevBuf aos, var, stage;
    HIP_TRY(ctx, aos.alloc((size_t)3 * M * sizeof(double)));
    if (ref) {
        launch_soa_to_aos(ctx, m->ref, M, aos.as<double>(), m->perm);
        HIP_TRY(ctx, hipMemcpyAsync(ref, aos.p, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (mean) {
        launch_soa_to_aos(ctx, m->mean, M, aos.as<double>(), m->perm);
        HIP_TRY(ctx, hipMemcpyAsync(mean, aos.p, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (variance) memcpy(variance, m->variance.data(), (size_t)m->r * sizeof(double));
    if (basis_colmajor) {
        HIP_TRY(ctx, var.alloc((size_t)m->r * sizeof(double)));
        HIP_TRY(ctx, stage.alloc((size_t)3 * M * m->r * sizeof(double)));
        HIP_TRY(ctx, hipMemcpyAsync(var.p, m->variance.data(), (size_t)m->r * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        const int64_t total = 3 * M * m->r;
        hipLaunchKernelGGL(unpack_basis_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, m->Q0,
                           var.as<double>(), M, m->r, m->rp, m->perm, stage.as<double>());
        GINGR_TRY(check(ctx));
        HIP_TRY(ctx, hipMemcpyAsync(basis_colmajor, stage.p, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return GINGR_OK;
}

}  // extern "C"

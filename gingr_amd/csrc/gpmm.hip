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

struct Best {
    double v;
    int32_t i;
};

__device__ __forceinline__ Best better(Best a, Best b) {  // larger value; ties -> lower index; NaN never wins
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// block-wide (max, argmax, sum) with a fixed tree: identical in every workgroup that reduces the same data
__device__ __forceinline__ void block_best_sum(Best &b, double &sum, Best *shb, double *shs) {
    const int t = threadIdx.x;
    shb[t] = b;
    shs[t] = sum;
    __syncthreads();
    for (int off = kPcBlock / 2; off > 0; off >>= 1) {
        if (t < off) {
            shb[t] = better(shb[t], shb[t + off]);
            shs[t] += shs[t + off];
        }
        __syncthreads();
    }
    b = shb[0];
    sum = shs[0];
    __syncthreads();
}

__global__ __launch_bounds__(kPcBlock) void pc_init_kernel(int64_t M, Mixture mix, double *__restrict__ diag,
                                                           int32_t *__restrict__ pivoted, double *__restrict__ pmax,
                                                           int32_t *__restrict__ pidx, double *__restrict__ ptr,
                                                           int32_t *__restrict__ ctl) {
    __shared__ Best shb[kPcBlock];
    __shared__ double shs[kPcBlock];
    double d0 = 0.0;
    for (int i = 0; i < mix.n; ++i) d0 = i == 0 ? mix.s[i] : d0 + mix.s[i];  // k(x,x) = sum scaling_i * exp(0)
    const int64_t c = (int64_t)blockIdx.x * kPcBlock + threadIdx.x;
    Best b{0.0, -1};
    double tr = 0.0;
    if (c < M) {
        diag[c] = d0;
        pivoted[c] = 0;
        b = Best{d0, (int32_t)c};
        tr = d0;
    }
    block_best_sum(b, tr, shb, shs);
    if (threadIdx.x == 0) {
        pmax[blockIdx.x] = b.v;
        pidx[blockIdx.x] = b.i;
        ptr[blockIdx.x] = tr;
        if (blockIdx.x == 0) ctl[0] = ctl[1] = 0;
    }
}

// One pivot step.  Every workgroup first reduces the previous step's block partials (same data, same tree => same pivot
// everywhere, no grid synchronisation), then fills its rows of column k.
__global__ __launch_bounds__(kPcBlock) void pc_step_kernel(Cloud pts, Mixture mix, int32_t k, int32_t kmax, double rel_tol,
                                                           int32_t nblocks, double *__restrict__ L /* [kmax][M] */,
                                                           double *__restrict__ diag, int32_t *__restrict__ pivoted,
                                                           const double *__restrict__ pmax_in,
                                                           const int32_t *__restrict__ pidx_in,
                                                           const double *__restrict__ ptr_in, double *__restrict__ pmax,
                                                           int32_t *__restrict__ pidx, double *__restrict__ ptr,
                                                           int32_t *__restrict__ ctl, double *__restrict__ trace,
                                                           int32_t *__restrict__ pivots) {
    __shared__ double T[GINGR_EXP_TABLE];
    __shared__ Best shb[kPcBlock];
    __shared__ double shs[kPcBlock];
    __shared__ double Lp[512];
    __shared__ int32_t finished;
    const int t = threadIdx.x;
    if (t == 0) finished = ctl[1];  // set by an earlier launch (or, harmlessly, by workgroup 0 of this one)
    __syncthreads();
    if (finished) return;
    fastexp_table_init(T);
    Best g{0.0, -1};
    double gtr = 0.0;
    for (int b = t; b < nblocks; b += kPcBlock) {
        g = better(g, Best{pmax_in[b], pidx_in[b]});
        gtr += ptr_in[b];
    }
    block_best_sum(g, gtr, shb, shs);
    const double tol = rel_tol * trace[0];  // trace[0] was written by step 0 (k == 0 uses gtr itself)
    const bool stop = k >= kmax || g.i < 0 || !(gtr >= (k == 0 ? rel_tol * gtr : tol)) || !(g.v > 0.0);
    if (blockIdx.x == 0 && t == 0) {
        trace[k] = gtr;
        if (stop) {
            ctl[0] = k;
            ctl[1] = 1;
        } else {
            pivots[k] = g.i;
        }
    }
    if (stop) return;
    const int64_t M = pts.n;
    const int64_t p = g.i;
    const double lpk = sqrt(g.v);
    for (int r = t; r < k; r += kPcBlock) Lp[r] = L[(int64_t)r * M + p];
    __syncthreads();
    const double px = pts.x[p], py = pts.y[p], pz = pts.z[p];
    const int64_t c = (int64_t)blockIdx.x * kPcBlock + t;
    Best b{0.0, -1};
    double tr = 0.0;
    if (c < M) {
        double l;
        if (c == p) {
            l = lpk;
            pivoted[c] = 1;
        } else if (pivoted[c]) {
            l = 0.0;
        } else {
            double S = 0.0;
            for (int r = 0; r < k; ++r) S = __dadd_rn(S, __dmul_rn(L[(int64_t)r * M + c], Lp[r]));
            const double dx = pts.x[c] - px, dy = pts.y[c] - py, dz = pts.z[c] - pz;
            const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
            l = (mixture_value(mix, d2, T) - S) / lpk;
            const double dc = __dadd_rn(diag[c], -__dmul_rn(l, l));
            diag[c] = dc;
            b = Best{dc, (int32_t)c};
            tr = dc;
        }
        L[(int64_t)k * M + c] = l;
    }
    block_best_sum(b, tr, shb, shs);
    if (t == 0) {
        pmax[blockIdx.x] = b.v;
        pidx[blockIdx.x] = b.i;
        ptr[blockIdx.x] = tr;
    }
}

// G[a][b] = sum_c L[a][c] L[b][c]   (a <= b computed, mirrored); grid (k, k)
__global__ __launch_bounds__(kPcBlock) void pc_gram_kernel(const double *__restrict__ L, int64_t M, int32_t k,
                                                           double *__restrict__ G) {
    __shared__ double shs[kPcBlock];
    const int a = blockIdx.x, b = blockIdx.y;
    if (a > b) return;
    const double *la = L + (int64_t)a * M, *lb = L + (int64_t)b * M;
    double s = 0.0;
    for (int64_t c = threadIdx.x; c < M; c += kPcBlock) s = __builtin_fma(la[c], lb[c], s);
    shs[threadIdx.x] = s;
    __syncthreads();
    for (int off = kPcBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) shs[threadIdx.x] += shs[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        G[(int64_t)a * k + b] = shs[0];
        G[(int64_t)b * k + a] = shs[0];
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
                const int m = idx / n, i = idx - m * n;
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

// B[i][c] = sum_r L[r][c] V[r][i]   (c over ALL points, i < n): column i of  U sqrt(lambda) = L V
__global__ __launch_bounds__(256) void lv_kernel(const double *__restrict__ L, int64_t M, int32_t n,
                                                 const double *__restrict__ V, double *__restrict__ B) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (c >= M) return;
    double acc = 0.0;
    for (int r = 0; r < n; ++r) acc = __builtin_fma(L[(int64_t)r * M + c], V[r * n + i], acc);
    B[(int64_t)i * M + c] = acc;
}

// Q0[(3s+d)*rp + q] = B_set(q)[idx(q)][row_begin + perm[s]] when coordinate(q) == d, else 0
__global__ __launch_bounds__(256) void gpmm_pack_kernel(const double *__restrict__ BA, const double *__restrict__ BB,
                                                        int64_t M_total, int64_t row_begin, int64_t M, int32_t r, int32_t rp,
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
        v = (qset[q] == 0 ? BA : BB)[(int64_t)qidx[q] * M_total + c];
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

// all-pairs extrema of |p_i - p_j|^2 (unfused), i != j: per-block partials
__global__ __launch_bounds__(256) void dist_extrema_kernel(Cloud pts, double *__restrict__ pmax, double *__restrict__ pmin) {
    __shared__ double tx[256], ty[256], tz[256];
    __shared__ double sh[256];
    const int t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 256 + t;
    const bool ok = i < pts.n;
    const double x = ok ? pts.x[i] : 0.0, y = ok ? pts.y[i] : 0.0, z = ok ? pts.z[i] : 0.0;
    double mx = 0.0, mn = __builtin_huge_val();
    for (int64_t jb = 0; jb < pts.n; jb += 256) {
        __syncthreads();
        if (jb + t < pts.n) {
            tx[t] = pts.x[jb + t];
            ty[t] = pts.y[jb + t];
            tz[t] = pts.z[jb + t];
        }
        __syncthreads();
        const int cnt = (int)min((int64_t)256, pts.n - jb);
        if (ok)
            for (int jj = 0; jj < cnt; ++jj) {
                const double dx = x - tx[jj], dy = y - ty[jj], dz = z - tz[jj];
                const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                mx = fmax(mx, d2);
                if (jb + jj != i) mn = fmin(mn, d2);
            }
    }
    sh[t] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) sh[t] = fmax(sh[t], sh[t + off]);
        __syncthreads();
    }
    if (t == 0) pmax[blockIdx.x] = sh[0];
    __syncthreads();
    sh[t] = mn;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) sh[t] = fmin(sh[t], sh[t + off]);
        __syncthreads();
    }
    if (t == 0) pmin[blockIdx.x] = sh[0];
}

int check(gingr_ctx *ctx) {
    HIP_TRY(ctx, hipGetLastError());
    return GINGR_OK;
}

}  // namespace

extern "C" {

int gingr_pointset_distance_extrema(gingr_ctx *ctx, const double *xyz, int64_t n, double *max_distance,
                                    double *min_distance) {
    if (!ctx || !xyz || n < 1 || !max_distance || !min_distance)
        return ctx ? gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "distance_extrema: bad argument") : GINGR_ERR_BAD_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int nb = (int)ceil_div(n, 256);
    DevBuf aos, soa, pm;
    HIP_TRY(ctx, aos.alloc((size_t)3 * n * sizeof(double)));
    HIP_TRY(ctx, soa.alloc((size_t)3 * n * sizeof(double)));
    HIP_TRY(ctx, pm.alloc((size_t)2 * nb * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(aos.p, xyz, (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, aos.as<double>(), n, soa.as<double>());
    const double *s = soa.as<double>();
    hipLaunchKernelGGL(dist_extrema_kernel, dim3(nb), dim3(256), 0, ctx->stream, Cloud{s, s + n, s + 2 * n, n},
                       pm.as<double>(), pm.as<double>() + nb);
    GINGR_TRY(check(ctx));
    std::vector<double> h((size_t)2 * nb);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), pm.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double mx = 0.0, mn = HUGE_VAL;
    for (int b = 0; b < nb; ++b) {
        mx = std::max(mx, h[(size_t)b]);
        mn = std::min(mn, h[(size_t)nb + b]);
    }
    *max_distance = std::sqrt(mx);   // max of sqrt = sqrt of max (sqrt is monotone and correctly rounded)
    *min_distance = std::sqrt(mn);   // +inf for a single point
    return GINGR_OK;
}

int gingr_gpmm_build_gaussian(gingr_ctx *ctx, int64_t M_total, const double *ref, int32_t n_kernels, const double *sigmas,
                              const double *scalings, double relative_tolerance, int32_t max_rank, int64_t row_begin,
                              int64_t row_end, gingr_model **out) {
    if (!ctx || !out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (M_total < 1 || !ref || n_kernels < 1 || n_kernels > kMaxMix || !sigmas || !scalings)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: need M >= 1 and 1..%d kernels", kMaxMix);
    if (!(relative_tolerance >= 0.0) || !(relative_tolerance < 1.0))
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: relative tolerance must be in [0, 1)");
    Mixture mix;
    memset(&mix, 0, sizeof(mix));
    mix.n = n_kernels;
    for (int i = 0; i < n_kernels; ++i) {
        if (!(sigmas[i] > 0.0) || !(scalings[i] > 0.0) || !std::isfinite(sigmas[i]) || !std::isfinite(scalings[i]))
            return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: sigma and scaling must be positive");
        mix.c[i] = -(double)GINGR_EXP_TABLE * 1.4426950408889634074 / (sigmas[i] * sigmas[i]);
        mix.s[i] = scalings[i];
    }
    if (max_rank <= 0 || max_rank > 512) max_rank = 512;  // the model's rank limit (gingr_model_upload)
    if ((int64_t)max_rank > 3 * M_total) max_rank = (int32_t)(3 * M_total);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = M_total;
    const int32_t kmax = (int32_t)std::min<int64_t>(M, (max_rank + 2) / 3);  // scalar columns that can ever be needed
    const int nb = (int)ceil_div(M, kPcBlock);

    DevBuf aos, soa, Lb, diag, piv, part, ctl, trace, pivots;
    HIP_TRY(ctx, aos.alloc((size_t)3 * M * sizeof(double)));
    HIP_TRY(ctx, soa.alloc((size_t)3 * M * sizeof(double)));
    HIP_TRY(ctx, Lb.alloc((size_t)kmax * M * sizeof(double)));
    HIP_TRY(ctx, diag.alloc((size_t)M * sizeof(double)));
    HIP_TRY(ctx, piv.alloc((size_t)M * sizeof(int32_t)));
    HIP_TRY(ctx, part.alloc((size_t)2 * nb * (2 * sizeof(double) + sizeof(int32_t)) + 64));
    HIP_TRY(ctx, ctl.alloc(2 * sizeof(int32_t)));
    HIP_TRY(ctx, trace.alloc((size_t)(kmax + 1) * sizeof(double)));
    HIP_TRY(ctx, pivots.alloc((size_t)(kmax + 1) * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpyAsync(aos.p, ref, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, aos.as<double>(), M, soa.as<double>());
    const double *sp = soa.as<double>();
    const Cloud pts{sp, sp + M, sp + 2 * M, M};
    double *pmaxv[2] = {part.as<double>(), part.as<double>() + nb};
    double *ptrv[2] = {part.as<double>() + 2 * nb, part.as<double>() + 3 * nb};
    int32_t *pidxv[2] = {reinterpret_cast<int32_t *>(part.as<double>() + 4 * nb),
                         reinterpret_cast<int32_t *>(part.as<double>() + 4 * nb) + nb};
    hipLaunchKernelGGL(pc_init_kernel, dim3(nb), dim3(kPcBlock), 0, ctx->stream, M, mix, diag.as<double>(), piv.as<int32_t>(),
                       pmaxv[0], pidxv[0], ptrv[0], ctl.as<int32_t>());
    int32_t hctl[2] = {0, 0};
    for (int32_t k = 0; k <= kmax; ++k) {
        const int in = k & 1, o = in ^ 1;
        hipLaunchKernelGGL(pc_step_kernel, dim3(nb), dim3(kPcBlock), 0, ctx->stream, pts, mix, k, kmax, relative_tolerance, nb,
                           Lb.as<double>(), diag.as<double>(), piv.as<int32_t>(), pmaxv[in], pidxv[in], ptrv[in], pmaxv[o],
                           pidxv[o], ptrv[o], ctl.as<int32_t>(), trace.as<double>(), pivots.as<int32_t>());
        if ((k & 15) == 15 || k == kmax) {  // look at the stop flag now and then instead of queueing no-op launches
            GINGR_TRY(check(ctx));
            HIP_TRY(ctx, hipMemcpyAsync(hctl, ctl.p, sizeof(hctl), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (hctl[1]) break;
        }
    }
    if (!hctl[1]) return gingr_set_error(ctx, GINGR_ERR_STATE, "gpmm_build: pivoted Cholesky did not terminate");
    const int32_t ks = hctl[0];
    if (ks < 1) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: empty model (tolerance too large)");
    std::vector<double> htr((size_t)ks + 1);
    HIP_TRY(ctx, hipMemcpy(htr.data(), trace.p, htr.size() * sizeof(double), hipMemcpyDeviceToHost));
    // generic (3M-index) stopping rule on the scalar traces: after n = 3j + e pivots tr = (3-e) tr_s(j) + e tr_s(j+1)
    const double tol_g = relative_tolerance * (3.0 * htr[0]);
    int32_t n = 0;
    const int32_t nmax = std::min<int32_t>(max_rank, 3 * ks);
    while (n < nmax) {
        const int32_t j = n / 3, e = n % 3;
        const double tr = e == 0 ? 3.0 * htr[(size_t)j] : (3 - e) * htr[(size_t)j] + e * htr[(size_t)j + 1];
        if (!(tr >= tol_g)) break;
        ++n;
    }
    if (n < 1) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gpmm_build: empty model (tolerance too large)");
    const int32_t j = n / 3, e = n % 3;
    const int32_t nA = e ? j + 1 : 0, nB = j;  // coordinates d < e own nA scalar columns, the others nB
    const int32_t kk = std::max(nA, nB);

    // Gram of the scalar factor and the eigen-decompositions of its leading blocks
    DevBuf G, wA, wV, evA, evB, VA, VB, BA, BB, sw;
    HIP_TRY(ctx, G.alloc((size_t)kk * kk * sizeof(double)));
    HIP_TRY(ctx, wA.alloc((size_t)kk * kk * sizeof(double)));
    HIP_TRY(ctx, wV.alloc((size_t)kk * kk * sizeof(double)));
    HIP_TRY(ctx, evA.alloc((size_t)(nA + 1) * sizeof(double)));
    HIP_TRY(ctx, evB.alloc((size_t)(nB + 1) * sizeof(double)));
    HIP_TRY(ctx, VA.alloc((size_t)(nA * nA + 1) * sizeof(double)));
    HIP_TRY(ctx, VB.alloc((size_t)(nB * nB + 1) * sizeof(double)));
    HIP_TRY(ctx, BA.alloc(((size_t)nA * M + 1) * sizeof(double)));
    HIP_TRY(ctx, BB.alloc(((size_t)nB * M + 1) * sizeof(double)));
    HIP_TRY(ctx, sw.alloc(2 * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemsetAsync(sw.p, 0, 2 * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(pc_gram_kernel, dim3(kk, kk), dim3(kPcBlock), 0, ctx->stream, Lb.as<double>(), M, kk, G.as<double>());
    if (nA) {
        hipLaunchKernelGGL(jacobi_eig_kernel, dim3(1), dim3(kJacThreads), 0, ctx->stream, G.as<double>(), kk, nA, wA.as<double>(),
                           wV.as<double>(), evA.as<double>(), VA.as<double>(), sw.as<int32_t>());
        hipLaunchKernelGGL(lv_kernel, dim3((unsigned)ceil_div(M, 256), nA), dim3(256), 0, ctx->stream, Lb.as<double>(), M, nA,
                           VA.as<double>(), BA.as<double>());
    }
    if (nB) {
        hipLaunchKernelGGL(jacobi_eig_kernel, dim3(1), dim3(kJacThreads), 0, ctx->stream, G.as<double>(), kk, nB, wA.as<double>(),
                           wV.as<double>(), evB.as<double>(), VB.as<double>(), sw.as<int32_t>() + 1);
        hipLaunchKernelGGL(lv_kernel, dim3((unsigned)ceil_div(M, 256), nB), dim3(256), 0, ctx->stream, Lb.as<double>(), M, nB,
                           VB.as<double>(), BB.as<double>());
    }
    GINGR_TRY(check(ctx));
    std::vector<double> hA((size_t)nA + 1), hB((size_t)nB + 1);
    int32_t hsw[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(hA.data(), evA.p, (size_t)nA * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hB.data(), evB.p, (size_t)nB * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hsw, sw.p, sizeof(hsw), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (hsw[0] >= 60 || hsw[1] >= 60) return gingr_set_error(ctx, GINGR_ERR_NONFINITE, "gpmm_build: Jacobi did not converge");

    // merged eigenpairs, descending; equal eigenvalues keep coordinate order (x, y, z)
    struct Col {
        double lam;
        int32_t dim, set, idx;
    };
    std::vector<Col> cols;
    for (int d = 0; d < 3; ++d) {
        const bool a = d < e;
        const int32_t cnt = a ? nA : nB;
        for (int32_t i = 0; i < cnt; ++i) cols.push_back(Col{a ? hA[(size_t)i] : hB[(size_t)i], d, a ? 0 : 1, i});
    }
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

    if (row_end <= 0) row_end = M_total;
    std::vector<double> zero_mean((size_t)3 * M_total, 0.0);
    auto fill = [&](gingr_model *m) -> int {
        const int64_t total = 3 * m->M * m->rp;
        hipLaunchKernelGGL(gpmm_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, BA.as<double>(),
                           BB.as<double>(), M_total, m->row_begin, m->M, m->r, m->rp, m->perm, dq.as<int32_t>(),
                           dq.as<int32_t>() + rank, dq.as<int32_t>() + 2 * rank, m->Q0);
        return check(ctx);
    };
    return model_create_impl(ctx, M_total, rank, ref, zero_mean.data(), variance.data(), row_begin, row_end, fill, out);
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

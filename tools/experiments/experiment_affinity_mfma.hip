// CPD affinity passes with the scaled squared distances produced by the f64 matrix pipe (gfx950).
//
// RECORD OF A CLOSED EXPERIMENT -- not built, not part of libgingr_hip.so (rounds 1-2 shipped it behind GINGR_AFFINITY=mfma; round 3
// moved it here, the launch hooks in affinity.hip / common.h are gone).  Parity-green but NOT faster than affinity.hip: measured on gfx950, f64 MFMA and
// f64 VALU instructions do not overlap (they share the double-precision hardware; profiles/r01_ubench_mfma_valu_overlap.txt),
// so moving d2 to the matrix pipe only moves the time.  Kept as the evidence for DESIGN.md section 4.
//
// The VALU formulation (affinity.hip) spends 6 of its 15 / 19 f64 instructions per pair on d2 = |x - y|^2.  The idea was
// to produce the exponent argument on the matrix pipe:
//     t_ij = c*|x_j - y_i|^2 = c|x_j|^2 + c|y_i|^2 - 2c x_j.y_i        (c = -2048 log2(e) / (2 sigma2))
// is one v_mfma_f64_16x16x4_f64 per 16x16 pair tile: A row = (-2c y, c|y|^2) of the streamed point, B column =
// (x, 1) of the owned point, C = c|x|^2 (a per-lane constant).  The VALU is left with the exponential only:
// 9 instructions per pair in pass 1 (was 15) and 13 in pass 2 (was 21).
//
// Cancellation: the expansion is evaluated on coordinates centred on the target centroid; the absolute error of t is
// ~4 ulp(|c| R^2) (R = cloud radius), i.e. a relative error of K_ij of about 4.4e-16 * R^2 / (2 sigma2)  (2e-12 for the
// benchmark clouds at sigma2 = 4; 1e-10 for a 200 mm mesh at sigma2 = 0.1) against the 1e-5 budget on vertex positions.
// STATUS: opt-in experiment (GINGR_AFFINITY=mfma), parity-tested like the default.  It is NOT faster: on gfx950 the f64
// MFMA and the f64 VALU share the double-precision hardware (4 MFMA + 32 v_fma_f64 per iteration take the SUM of their
// separate times, profiles/r01_ubench_mfma_valu_overlap.txt), so the 6 VALU instructions saved per pair are paid back
// as matrix-pipe cycles (measured: pass 1 1.43 ms vs 1.45 ms, pass 2 2.31 ms vs 1.85 ms at 50k x 50k).
//
// Mapping (D = A B + C, 16x16x4): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it receives
// D[i = (l>>4) + 4 reg][j = l&15], reg = 0..3.  The OWNED points are the columns j (a per-lane constant, so the running
// sums live in that lane's registers); the STREAMED points are the rows i, staged through LDS as four planes.
#include "common.h"
#include "fastexp.h"

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTileM = 256;       // streamed points per LDS tile
constexpr int kPlanePad = 16;     // plane stride 272 doubles: the four k-planes of one read hit distinct banks
constexpr int kOwnedPerWave = 64;  // 4 column groups of 16
constexpr int kOwnedPerBlock = 256;
constexpr double kTInvalid = -4.6e6;  // exponent argument of a padding row: 2^(-2246) underflows to exactly +0
constexpr double kTLimit = -2.3e6;    // clamp (only when the range check asks for it)

struct __attribute__((aligned(32))) P4 {
    double x, y, z, w;
};

template <bool CLAMP>
__device__ __forceinline__ double exp_from_t(double t, const double *T) {
    if (CLAMP) t = fmax(t, kTLimit);
    const double tm = t + GINGR_EXP_MAGIC;
    const double kf = tm - GINGR_EXP_MAGIC;
    const double f = t - kf;  // exact
    return fastexp2_core(tm, f, T);
}

// ------------------------------------------------------------------------------------------------ pass 1
template <bool CLAMP>
__device__ __forceinline__ void colsum_steps(const double (*A)[kTileM + kPlanePad], int nsteps, int kq, int cl,
                                             const double (&b)[4], const v4f64 (&cinit)[4], double (&acc)[4], const double *T) {
    for (int s = 0; s < nsteps; ++s) {
        const double a = A[kq][s * 16 + cl];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4f64 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[g], cinit[g], 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[g] += exp_from_t<CLAMP>(d[reg], T);
        }
    }
}

__global__ __launch_bounds__(kThreads) void cpd_colsum_mfma_kernel(Cloud fit, Cloud tgt, const double *__restrict__ sigma2,
                                                                   const double *__restrict__ aux, int64_t rows_per_chunk,
                                                                   double *__restrict__ partial) {
    __shared__ double T[GINGR_EXP_TABLE];
    __shared__ double A[4][kTileM + kPlanePad];
    fastexp_table_init(T);
    const double c = fastexp_scale_for_variance(2.0 * sigma2[0]);
    const double am = aux[0] + aux[1];
    const bool clamp = fastexp_needs_clamp(3.0 * am * am, c);  // wave-uniform
    const double cx = aux[2], cy = aux[3], cz = aux[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, cl = lane & 15;
    double b[4], acc[4];
    v4f64 cinit[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t j = (int64_t)blockIdx.x * kOwnedPerBlock + wave * kOwnedPerWave + g * 16 + cl;
        const bool ok = j < tgt.n;
        const double x = ok ? tgt.x[j] - cx : 0.0, y = ok ? tgt.y[j] - cy : 0.0, z = ok ? tgt.z[j] - cz : 0.0;
        b[g] = kq == 0 ? x : (kq == 1 ? y : (kq == 2 ? z : 1.0));
        const double n2 = c * __builtin_fma(z, z, __builtin_fma(y, y, x * x));
        cinit[g] = v4f64{n2, n2, n2, n2};
        acc[g] = 0.0;
    }
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t i1 = min(fit.n, i0 + rows_per_chunk);
    const double m2c = -2.0 * c;
    for (int64_t ib = i0; ib < i1; ib += kTileM) {
        __syncthreads();
        {
            const int64_t i = ib + tid;
            if (i < i1) {
                const double x = fit.x[i] - cx, y = fit.y[i] - cy, z = fit.z[i] - cz;
                A[0][tid] = m2c * x;
                A[1][tid] = m2c * y;
                A[2][tid] = m2c * z;
                A[3][tid] = c * __builtin_fma(z, z, __builtin_fma(y, y, x * x));
            } else {
                A[0][tid] = 0.0;
                A[1][tid] = 0.0;
                A[2][tid] = 0.0;
                A[3][tid] = kTInvalid;
            }
        }
        __syncthreads();
        const int cnt = (int)min((int64_t)kTileM, i1 - ib);
        const int nsteps = (cnt + 15) >> 4;
        if (clamp)
            colsum_steps<true>(A, nsteps, kq, cl, b, cinit, acc, T);
        else
            colsum_steps<false>(A, nsteps, kq, cl, b, cinit, acc, T);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        double v = acc[g];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        const int64_t j = (int64_t)blockIdx.x * kOwnedPerBlock + wave * kOwnedPerWave + g * 16 + cl;
        if (kq == 0 && j < tgt.n) partial[(int64_t)blockIdx.y * tgt.n + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------ pass 2
template <bool CLAMP>
__device__ __forceinline__ void rowstats_steps(const double (*A)[kTileM + kPlanePad], const P4 *E, int nsteps, int kq, int cl,
                                               const double (&b)[4], const v4f64 (&cinit)[4], double (&a1)[4], double (&ax)[4],
                                               double (&ay)[4], double (&az)[4], const double *T) {
    for (int s = 0; s < nsteps; ++s) {
        const double a = A[kq][s * 16 + cl];
        P4 e[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) e[reg] = E[s * 16 + kq + 4 * reg];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4f64 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[g], cinit[g], 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double p = exp_from_t<CLAMP>(d[reg], T) * e[reg].w;
                a1[g] += p;
                ax[g] = __builtin_fma(p, e[reg].x, ax[g]);
                ay[g] = __builtin_fma(p, e[reg].y, ay[g]);
                az[g] = __builtin_fma(p, e[reg].z, az[g]);
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void cpd_rowstats_mfma_kernel(Cloud fit, Cloud tgt, const double *__restrict__ sigma2,
                                                                     const double *__restrict__ aux,
                                                                     const double *__restrict__ inv_den, int64_t cols_per_chunk,
                                                                     double *__restrict__ partial) {
    __shared__ double T[GINGR_EXP_TABLE];
    __shared__ double A[4][kTileM + kPlanePad];
    __shared__ P4 E[kTileM];
    fastexp_table_init(T);
    const double c = fastexp_scale_for_variance(2.0 * sigma2[0]);
    const double am = aux[0] + aux[1];
    const bool clamp = fastexp_needs_clamp(3.0 * am * am, c);  // wave-uniform
    const double cx = aux[2], cy = aux[3], cz = aux[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, cl = lane & 15;
    double b[4], a1[4], ax[4], ay[4], az[4];
    v4f64 cinit[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t i = (int64_t)blockIdx.x * kOwnedPerBlock + wave * kOwnedPerWave + g * 16 + cl;
        const bool ok = i < fit.n;
        const double x = ok ? fit.x[i] - cx : 0.0, y = ok ? fit.y[i] - cy : 0.0, z = ok ? fit.z[i] - cz : 0.0;
        b[g] = kq == 0 ? x : (kq == 1 ? y : (kq == 2 ? z : 1.0));
        const double n2 = c * __builtin_fma(z, z, __builtin_fma(y, y, x * x));
        cinit[g] = v4f64{n2, n2, n2, n2};
        a1[g] = ax[g] = ay[g] = az[g] = 0.0;
    }
    const int64_t j0 = (int64_t)blockIdx.y * cols_per_chunk;
    const int64_t j1 = min(tgt.n, j0 + cols_per_chunk);
    const double m2c = -2.0 * c;
    for (int64_t jb = j0; jb < j1; jb += kTileM) {
        __syncthreads();
        {
            const int64_t j = jb + tid;
            if (j < j1) {
                const double x = tgt.x[j] - cx, y = tgt.y[j] - cy, z = tgt.z[j] - cz;
                A[0][tid] = m2c * x;
                A[1][tid] = m2c * y;
                A[2][tid] = m2c * z;
                A[3][tid] = c * __builtin_fma(z, z, __builtin_fma(y, y, x * x));
                E[tid] = P4{x, y, z, inv_den[j]};
            } else {
                A[0][tid] = 0.0;
                A[1][tid] = 0.0;
                A[2][tid] = 0.0;
                A[3][tid] = kTInvalid;
                E[tid] = P4{0.0, 0.0, 0.0, 0.0};
            }
        }
        __syncthreads();
        const int cnt = (int)min((int64_t)kTileM, j1 - jb);
        const int nsteps = (cnt + 15) >> 4;
        if (clamp)
            rowstats_steps<true>(A, E, nsteps, kq, cl, b, cinit, a1, ax, ay, az, T);
        else
            rowstats_steps<false>(A, E, nsteps, kq, cl, b, cinit, a1, ax, ay, az, T);
    }
    const int64_t M = fit.n;
    double *base = partial + (int64_t)blockIdx.y * 4 * M;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        double v1 = a1[g], vx = ax[g], vy = ay[g], vz = az[g];
        v1 += __shfl_xor(v1, 16);
        vx += __shfl_xor(vx, 16);
        vy += __shfl_xor(vy, 16);
        vz += __shfl_xor(vz, 16);
        v1 += __shfl_xor(v1, 32);
        vx += __shfl_xor(vx, 32);
        vy += __shfl_xor(vy, 32);
        vz += __shfl_xor(vz, 32);
        const int64_t i = (int64_t)blockIdx.x * kOwnedPerBlock + wave * kOwnedPerWave + g * 16 + cl;
        if (kq == 0 && i < M) {
            // sums were taken over centred targets: sum_j p (x_j - ctr) + ctr * sum_j p
            base[i] = v1;
            base[M + i] = __builtin_fma(cx, v1, vx);
            base[2 * M + i] = __builtin_fma(cy, v1, vy);
            base[3 * M + i] = __builtin_fma(cz, v1, vz);
        }
    }
}

// centroid of a cloud into out[0..2] (single workgroup, fixed order)
__global__ __launch_bounds__(1024) void cloud_centroid_kernel(Cloud c, double *__restrict__ out) {
    __shared__ double sh[3][1024];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int64_t i = threadIdx.x; i < c.n; i += 1024) {
        sx += c.x[i];
        sy += c.y[i];
        sz += c.z[i];
    }
    sh[0][threadIdx.x] = sx;
    sh[1][threadIdx.x] = sy;
    sh[2][threadIdx.x] = sz;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int d = 0; d < 3; ++d) sh[d][threadIdx.x] += sh[d][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        const double v = sh[threadIdx.x][0] / (double)c.n;
        out[threadIdx.x] = (v == v && fabs(v) < 1e300) ? v : 0.0;  // a non-finite centroid would poison every pair
    }
}

constexpr int kTargetBlocks = 1536;  // ~6 workgroups per CU

inline void plan_chunks_mfma(int64_t owned, int64_t stream_len, int *nchunks, int64_t *chunk_len) {
    const int64_t bx = ceil_div(owned, kOwnedPerBlock);
    int64_t want = ceil_div(kTargetBlocks, bx > 0 ? bx : 1);
    const int64_t max_chunks = ceil_div(stream_len, kTileM);
    if (want > max_chunks) want = max_chunks;
    if (want < 1) want = 1;
    int64_t len = round_up(ceil_div(stream_len, want), kTileM);
    if (len < kTileM) len = kTileM;
    *chunk_len = len;
    *nchunks = (int)ceil_div(stream_len > 0 ? stream_len : 1, len);
}

}  // namespace

int64_t cpd_colsum_mfma_ws_doubles(int64_t M, int64_t N) {
    int nch;
    int64_t len;
    plan_chunks_mfma(N, M, &nch, &len);
    return (int64_t)nch * N;
}

int64_t cpd_rowstats_mfma_ws_doubles(int64_t M, int64_t N) {
    int nch;
    int64_t len;
    plan_chunks_mfma(M, N, &nch, &len);
    return (int64_t)nch * 4 * M;
}

void launch_cloud_centroid(gingr_ctx *ctx, Cloud c, double *out3) {
    hipLaunchKernelGGL(cloud_centroid_kernel, dim3(1), dim3(1024), 0, ctx->stream, c, out3);
}

int launch_cpd_colsum_mfma(gingr_ctx *ctx, Cloud fit, Cloud target, const double *sigma2_dev, const double *aux, double *ws,
                           int *nchunks_out) {
    int nch;
    int64_t len;
    plan_chunks_mfma(target.n, fit.n, &nch, &len);
    dim3 grid((unsigned)ceil_div(target.n, kOwnedPerBlock), (unsigned)nch);
    hipLaunchKernelGGL(cpd_colsum_mfma_kernel, grid, dim3(kThreads), 0, ctx->stream, fit, target, sigma2_dev, aux, len, ws);
    *nchunks_out = nch;
    return 0;
}

int launch_cpd_rowstats_mfma(gingr_ctx *ctx, Cloud fit, Cloud target, const double *sigma2_dev, const double *aux,
                             const double *inv_den, double *ws, int *nchunks_out) {
    int nch;
    int64_t len;
    plan_chunks_mfma(fit.n, target.n, &nch, &len);
    dim3 grid((unsigned)ceil_div(fit.n, kOwnedPerBlock), (unsigned)nch);
    hipLaunchKernelGGL(cpd_rowstats_mfma_kernel, grid, dim3(kThreads), 0, ctx->stream, fit, target, sigma2_dev, aux, inv_den,
                       len, ws);
    *nchunks_out = nch;
    return 0;
}

// Classic Coherent Point Drift (the reference's `other/` family, Myronenko & Song 2010) on the device: a second consumer of the
// streaming affinity statistics (affinity.hip) and of the Gaussian kernel block.
//
//   CPDFactory(templatePoints, lambda, beta, w)         G/other/algorithms/cpd/CPDFactory.scala:28-80   (G = exp(-|a-b|^2 / 2 beta^2))
//   RigidCPD.Expectation / Maximization / Registration  G/other/algorithms/cpd/RigidCPD.scala:59-139
//   AffineCPD.Maximization                              G/other/algorithms/cpd/AffineCPD.scala:35-61
//   NonRigidCPD.Maximization                            G/other/algorithms/cpd/NonRigidCPD.scala:45-88
//
// Expectation never forms the M x N matrix P: the Maximizations only need P1 = P 1, Pt1 = P^T 1, P X and Np, which the two
// streaming passes of the GiNGR path already produce.  Rigid / affine: two O(M + N) reductions (weighted means, then the centred
// 3x3 moments) and one thread of 3x3 algebra.  Non-rigid: the M x M system (G + lambda sigma2 diag(1/P1)) W = diag(1/P1) P X - Y is
// symmetric positive definite (the reference solves it with LU, `A \ B`); it is factored here by a blocked right-looking Cholesky
// with 64-wide panels -- diagonal block in LDS (one workgroup, gp.hip:chol_block64_kernel), panel and trailing update as 64 x 64 MFMA tiles
// (v_mfma_f64_16x16x4) over the whole chip -- with the three right-hand sides riding along as extra rows of the matrix (the forward
// substitution is a by-product), a blocked backward substitution, and TY = Y + G W as one pass over G.
#include "common.h"
#include "svd3.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

constexpr int kNBc = 64;        // Cholesky panel width
constexpr int kRedBlocks = 64;  // workgroups of the O(M + N) reductions

typedef double v4f64 __attribute__((ext_vector_type(4)));

Cloud cloud_at(const double *soa, int64_t n) { return Cloud{soa, soa + n, soa + 2 * n, n}; }

__device__ __forceinline__ double block_sum256(double v, double *sh) {
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    return sh[0];
}

// partial[b][0..2] = sum_j Pt1_j x_j, [3..5] = sum_i P1_i y_i
__global__ __launch_bounds__(256) void means_kernel(Cloud X, const double *__restrict__ Pt1, Cloud Y, const double *__restrict__ P1,
                                                    double *__restrict__ partial) {
    __shared__ double sh[256];
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < X.n; j += (int64_t)kRedBlocks * 256) {
        const double p = Pt1[j];
        a[0] += p * X.x[j];
        a[1] += p * X.y[j];
        a[2] += p * X.z[j];
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < Y.n; i += (int64_t)kRedBlocks * 256) {
        const double p = P1[i];
        a[3] += p * Y.x[i];
        a[4] += p * Y.y[i];
        a[5] += p * Y.z[i];
    }
    for (int q = 0; q < 6; ++q) {
        const double t = block_sum256(a[q], sh);
        if (threadIdx.x == 0) partial[blockIdx.x * 6 + q] = t;
    }
}

// mu[0..2] = muX, mu[3..5] = muY   (RigidCPD.scala:117-118)
__global__ void means_finish_kernel(const double *__restrict__ partial, const double *__restrict__ scalars, double *__restrict__ mu) {
    if (threadIdx.x >= 6) return;
    double v = 0.0;
    for (int b = 0; b < kRedBlocks; ++b) v += partial[b * 6 + threadIdx.x];
    mu[threadIdx.x] = v / scalars[0];
}

// partial[b][0..8] = A = Xhat^T P^T Yhat (row-major; from P X: sum_i (PX_i - P1_i muX)(y_i - muY)^T), [9..17] = Yhat^T diag(P1) Yhat,
// [18] = trace(Xhat^T diag(Pt1) Xhat)
__global__ __launch_bounds__(256) void moments_kernel(Cloud X, const double *__restrict__ Pt1, Cloud Y, const double *__restrict__ P1,
                                                      const double *__restrict__ PX, const double *__restrict__ mu,
                                                      double *__restrict__ partial) {
    __shared__ double sh[256];
    double a[19];
    for (int q = 0; q < 19; ++q) a[q] = 0.0;
    const double mx[3] = {mu[0], mu[1], mu[2]}, my[3] = {mu[3], mu[4], mu[5]};
    const int64_t M = Y.n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)kRedBlocks * 256) {
        const double p = P1[i];
        const double u[3] = {PX[i] - p * mx[0], PX[M + i] - p * mx[1], PX[2 * M + i] - p * mx[2]};
        const double yh[3] = {Y.x[i] - my[0], Y.y[i] - my[1], Y.z[i] - my[2]};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                a[r * 3 + c] += u[r] * yh[c];
                a[9 + r * 3 + c] += p * yh[r] * yh[c];
            }
    }
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < X.n; j += (int64_t)kRedBlocks * 256) {
        const double dx = X.x[j] - mx[0], dy = X.y[j] - mx[1], dz = X.z[j] - mx[2];
        a[18] += Pt1[j] * ((dx * dx + dy * dy) + dz * dz);
    }
    for (int q = 0; q < 19; ++q) {
        const double t = block_sum256(a[q], sh);
        if (threadIdx.x == 0) partial[blockIdx.x * 19 + q] = t;
    }
}

__device__ double det3(const double *m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// params[0] = s (affine: 1), [1..9] = R or B (row-major), [10..12] = t; scalars[8] = the new sigma2
__global__ void transform_finish_kernel(const double *__restrict__ partial, const double *__restrict__ mu, double *__restrict__ scalars,
                                        int affine, double *__restrict__ params, int32_t *__restrict__ flag) {
    if (threadIdx.x != 0) return;
    double m[19];
    for (int q = 0; q < 19; ++q) {
        double v = 0.0;
        for (int b = 0; b < kRedBlocks; ++b) v += partial[b * 19 + q];
        m[q] = v;
    }
    const double *A = m, *YPY = m + 9, s1 = m[18], Np = scalars[0];
    double L[9], sc = 1.0, s2;
    if (!affine) {
        // svd(A) = U S V^T; C = diag(1, 1, det(U V^T)); R = U C V^T; s = tr(A^T R) / tr(Yhat^T P1 Yhat)     (RigidCPD.scala:124-131)
        double U[9], S[3], V[9], UVt[9];
        svd3(A, U, S, V);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) UVt[r * 3 + c] = U[r * 3] * V[c * 3] + U[r * 3 + 1] * V[c * 3 + 1] + U[r * 3 + 2] * V[c * 3 + 2];
        const double d = det3(UVt);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) L[r * 3 + c] = U[r * 3] * V[c * 3] + U[r * 3 + 1] * V[c * 3 + 1] + d * U[r * 3 + 2] * V[c * 3 + 2];
        double trAR = 0.0;
        for (int q = 0; q < 9; ++q) trAR += A[q] * L[q];
        sc = trAR / (YPY[0] + YPY[4] + YPY[8]);
        s2 = sc * trAR;
    } else {
        // B = Xhat^T P^T Yhat (Yhat^T P1 Yhat)^-1; s2 = tr(Xhat^T P^T Yhat B^T)                              (AffineCPD.scala:51-56)
        const double det = det3(YPY);
        double inv[9];
        inv[0] = (YPY[4] * YPY[8] - YPY[5] * YPY[7]) / det;
        inv[1] = (YPY[2] * YPY[7] - YPY[1] * YPY[8]) / det;
        inv[2] = (YPY[1] * YPY[5] - YPY[2] * YPY[4]) / det;
        inv[3] = (YPY[5] * YPY[6] - YPY[3] * YPY[8]) / det;
        inv[4] = (YPY[0] * YPY[8] - YPY[2] * YPY[6]) / det;
        inv[5] = (YPY[2] * YPY[3] - YPY[0] * YPY[5]) / det;
        inv[6] = (YPY[3] * YPY[7] - YPY[4] * YPY[6]) / det;
        inv[7] = (YPY[1] * YPY[6] - YPY[0] * YPY[7]) / det;
        inv[8] = (YPY[0] * YPY[4] - YPY[1] * YPY[3]) / det;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) L[r * 3 + c] = A[r * 3] * inv[c] + A[r * 3 + 1] * inv[3 + c] + A[r * 3 + 2] * inv[6 + c];
        s2 = 0.0;
        for (int q = 0; q < 9; ++q) s2 += A[q] * L[q];
    }
    params[0] = sc;
    bool fin = true;
    for (int q = 0; q < 9; ++q) {
        params[1 + q] = L[q];
        fin = fin && fabs(L[q]) <= 1.79769313486231570815e308;
    }
    for (int r = 0; r < 3; ++r) params[10 + r] = mu[r] - sc * (L[r * 3] * mu[3] + L[r * 3 + 1] * mu[4] + L[r * 3 + 2] * mu[5]);
    const double ns = (s1 - s2) / (Np * 3.0);
    scalars[8] = ns;
    if (!fin || !(fabs(ns) <= 1.79769313486231570815e308)) *flag = GINGR_ERR_NONFINITE;
}

// TY = s Y R^T + 1 t^T  (in place)
__global__ void apply_transform_kernel(int64_t M, double *__restrict__ ty, const double *__restrict__ params) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double s = params[0], *L = params + 1, *t = params + 10;
    const double x = ty[i], y = ty[M + i], z = ty[2 * M + i];
    ty[i] = s * (L[0] * x + L[1] * y + L[2] * z) + t[0];
    ty[M + i] = s * (L[3] * x + L[4] * y + L[5] * z) + t[1];
    ty[2 * M + i] = s * (L[6] * x + L[7] * y + L[8] * z) + t[2];
}

// ------------------------------------------------------------------------------------------------ non-rigid: the M x M system
// Aw: (Mp + 64) x Mp row-major (ld = Mp, Mp = M rounded up to 64).  Rows < M: lower triangle of G + lambda sigma2 diag(1/P1); rows
// M..Mp-1: identity (padding); rows Mp + d, d < 3: B[:, d]^T with B = diag(1/P1) P X - Y; the other border rows are zero.
__global__ void build_system_kernel(int64_t M, int64_t Mp, const double *__restrict__ G, const double *__restrict__ P1,
                                    const double *__restrict__ PX, const double *__restrict__ Y, const double *__restrict__ scalars,
                                    double lambda, double *__restrict__ Aw) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (c >= Mp) return;
    double v = 0.0;
    if (r < M) {
        if (c <= r) v = G[r * M + c];
        if (c == r) v += lambda * scalars[8] * (1.0 / P1[r]);
    } else if (r < Mp) {
        v = c == r ? 1.0 : 0.0;
    } else if (r < Mp + 3 && c < M) {
        const int64_t d = r - Mp;
        v = PX[d * M + c] * (1.0 / P1[c]) - Y[d * M + c];
    }
    Aw[r * Mp + c] = v;
}

// one 64 x 64 tile on the matrix pipe: D = (accumulate ? C : 0) + beta * A B^T with A = 64 rows of Ap, B = 64 rows of Bp, K = 64.
// Both operands are staged through LDS in two halves of 32 columns (coalesced 256-byte row segments in, conflict-free fragment
// reads out; a panel tile that overwrites its own A operand is safe because A is consumed from LDS before C is written).
// Fragment layout of v_mfma_f64_16x16x4: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15] and holds
// D[i = (l >> 4) + 4 reg][j = l & 15]; wave w owns the output rows 16 w .. 16 w + 15.
__device__ __forceinline__ void tile_abt(const double *Ap, int64_t lda, const double *Bp, int64_t ldb, double *Cp, int64_t ldc,
                                         bool accumulate, double beta) {
    __shared__ double As[kNBc][33], Bs[kNBc][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    v4f64 acc[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        const double *pc = Cp + (int64_t)(16 * wave + l4) * ldc + 16 * tj + l15;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[tj][g] = accumulate ? pc[(int64_t)4 * g * ldc] : 0.0;
    }
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        double va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u, r = e >> 5, c = e & 31;
            va[u] = Ap[(int64_t)r * lda + 32 * half + c];
            vb[u] = Bp[(int64_t)r * ldb + 32 * half + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 256 * u, r = e >> 5, c = e & 31;
            As[r][c] = beta * va[u];
            Bs[r][c] = vb[u];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double a = As[16 * wave + l15][4 * q + l4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[16 * tj + l15][4 * q + l4], acc[tj], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        double *pc = Cp + (int64_t)(16 * wave + l4) * ldc + 16 * tj + l15;
#pragma unroll
        for (int g = 0; g < 4; ++g) pc[(int64_t)4 * g * ldc] = acc[tj][g];
    }
}

// panel solve: L_ik = A_ik L_kk^-T for the row blocks i > k (the last one is the border with the right-hand sides)
// (inverse: the border is an identity block of nb block rows -- dense_spd_inverse -- whose block row b is still zero left of column b)
__global__ __launch_bounds__(256) void chol_panel_kernel(double *__restrict__ Aw, int64_t ld, int k, const double *__restrict__ Linv, int nb,
                                                         bool inverse) {
    const int64_t i = k + 1 + blockIdx.x;
    if (inverse && i >= nb && i - nb > k) return;
    double *aik = Aw + i * kNBc * ld + (int64_t)k * kNBc;
    tile_abt(aik, ld, Linv + (int64_t)k * kNBc * kNBc, kNBc, aik, ld, false, 1.0);
}

// trailing update: A_ij -= L_ik L_jk^T for k < j <= i (j a matrix block, i up to the border block)
__global__ __launch_bounds__(256) void chol_trailing_kernel(double *__restrict__ Aw, int64_t ld, int k, int nb, bool inverse) {
    const int64_t i = k + 1 + blockIdx.y, j = k + 1 + blockIdx.x;
    if (j > i || j >= nb) return;
    if (inverse && i >= nb && i - nb > k) return;
    tile_abt(Aw + i * kNBc * ld + (int64_t)k * kNBc, ld, Aw + j * kNBc * ld + (int64_t)k * kNBc, ld, Aw + i * kNBc * ld + j * kNBc, ld, true,
             -1.0);
}

// backward substitution L^T W = Z, step k (from the last panel to the first): every workgroup forms W_k = L_kk^-T Z_k from the border
// rows; workgroup j < k then applies Z_j -= L_kj^T W_k, workgroup k stores W_k (planes of stride Mp)
__global__ __launch_bounds__(256) void chol_backward_kernel(double *__restrict__ Aw, int64_t ld, int64_t Mp, int k,
                                                            const double *__restrict__ Linv, double *__restrict__ W) {
    __shared__ double Z[3][kNBc], Wk[3][kNBc];
    const int tid = threadIdx.x, c = tid & 63, d = tid >> 6;
    const int64_t kb = (int64_t)k * kNBc;
    if (d < 3) Z[d][c] = Aw[(Mp + d) * ld + kb + c];
    __syncthreads();
    if (d < 3) {
        const double *li = Linv + (int64_t)k * kNBc * kNBc;
        double v = 0.0;
#pragma unroll 16
        for (int p = 0; p < kNBc; ++p) v += li[p * kNBc + c] * Z[d][p];  // Linv is lower triangular: the entries p < c are zero
        Wk[d][c] = v;
    }
    __syncthreads();
    if (d >= 3) return;
    const int j = blockIdx.x;
    if (j == k) {
        W[(int64_t)d * Mp + kb + c] = Wk[d][c];
        return;
    }
    const int64_t jb = (int64_t)j * kNBc;
    double v = 0.0;
    const double *lkj = Aw + kb * ld + jb + c;
#pragma unroll 16
    for (int p = 0; p < kNBc; ++p) v += lkj[(int64_t)p * ld] * Wk[d][p];
    Aw[(Mp + d) * ld + jb + c] -= v;
}

// TY = Y + G W: 16 lanes per row of G
__global__ __launch_bounds__(256) void deform_kernel(int64_t M, int64_t Mp, const double *__restrict__ G, const double *__restrict__ W,
                                                     double *__restrict__ ty) {
    const int lane16 = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    double a[3] = {0, 0, 0};
    if (i < M)
        for (int64_t m = lane16; m < M; m += 16) {
            const double g = G[i * M + m];
            a[0] = __builtin_fma(g, W[m], a[0]);
            a[1] = __builtin_fma(g, W[Mp + m], a[1]);
            a[2] = __builtin_fma(g, W[2 * Mp + m], a[2]);
        }
    for (int q = 0; q < 3; ++q) {
        a[q] += __shfl_xor(a[q], 8);
        a[q] += __shfl_xor(a[q], 4);
        a[q] += __shfl_xor(a[q], 2);
        a[q] += __shfl_xor(a[q], 1);
    }
    if (i < M && lane16 == 0) {
        ty[i] += a[0];
        ty[M + i] += a[1];
        ty[2 * M + i] += a[2];
    }
}

// partial[b][0] = sum_i P1_i |TY_i|^2, [1] = sum_i TY_i . PX_i      (NonRigidCPD.scala:74-77)
__global__ __launch_bounds__(256) void nonrigid_sums_kernel(int64_t M, const double *__restrict__ ty, const double *__restrict__ P1,
                                                            const double *__restrict__ PX, double *__restrict__ partial) {
    __shared__ double sh[256];
    double a = 0.0, b = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)kRedBlocks * 256) {
        const double x = ty[i], y = ty[M + i], z = ty[2 * M + i];
        a += P1[i] * ((x * x + y * y) + z * z);
        b += (x * PX[i] + y * PX[M + i]) + z * PX[2 * M + i];
    }
    const double ta = block_sum256(a, sh);
    const double tb = block_sum256(b, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = ta;
        partial[blockIdx.x * 2 + 1] = tb;
    }
}

__global__ void nonrigid_finish_kernel(const double *__restrict__ partial, double *__restrict__ scalars, int32_t *__restrict__ flag) {
    if (threadIdx.x != 0) return;
    double ypy = 0.0, tr = 0.0;
    for (int b = 0; b < kRedBlocks; ++b) {
        ypy += partial[b * 2];
        tr += partial[b * 2 + 1];
    }
    const double ns = (scalars[1] - 2 * tr + ypy) / (scalars[0] * 3.0);
    scalars[8] = ns;
    if (!(fabs(ns) <= 1.79769313486231570815e308)) *flag = GINGR_ERR_NONFINITE;
}

}  // namespace

// Aw ((Mp + 64) x Mp, lower triangle of an SPD matrix + three right-hand sides in the border rows Mp .. Mp + 2, see
// build_system_kernel) -> W (three planes of stride Mp): blocked right-looking Cholesky, then the blocked backward substitution.
// *flag receives GINGR_ERR_NOT_SPD when a diagonal block fails.  (Also the posterior solve above rank 256: gp.hip, common.h.)
static void blocked_cholesky(gingr_ctx *ctx, double *Aw, int64_t Mp, double *Linv, int32_t *flag, bool inverse) {
    const int nb = (int)(Mp / kNBc);
    const int nrows = inverse ? 2 * nb : nb + 1;  // block rows: the matrix, then the border
    for (int k = 0; k < nb; ++k) {
        launch_chol_block64(ctx, Aw, Mp, k, Linv, flag);
        const int below = inverse ? nb + k + 1 : nrows;  // (inverse: border block rows past k are still zero)
        hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)(below - k - 1)), dim3(256), 0, ctx->stream, Aw, Mp, k, Linv, nb, inverse);
        if (k + 1 < nb)
            hipLaunchKernelGGL(chol_trailing_kernel, dim3((unsigned)(nb - k - 1), (unsigned)(below - k - 1)), dim3(256), 0, ctx->stream, Aw, Mp,
                               k, nb, inverse);
    }
}

void dense_spd_solve3(gingr_ctx *ctx, double *Aw, int64_t Mp, double *Linv, double *W, int32_t *flag) {
    blocked_cholesky(ctx, Aw, Mp, Linv, flag, false);
    const int nb = (int)(Mp / kNBc);
    if (!W) return;  // (the factor alone: the lower triangle of Aw then holds L)
    for (int k = nb - 1; k >= 0; --k)
        hipLaunchKernelGGL(chol_backward_kernel, dim3((unsigned)(k + 1)), dim3(256), 0, ctx->stream, Aw, Mp, Mp, k, Linv, W);
}

namespace {
// C tile (a, b) = sum over the column blocks kb >= max(a, b) of X_a,kb X_b,kb^T, X = L^-T (upper triangular, row stride ld)
__global__ __launch_bounds__(256) void inverse_product_kernel(const double *__restrict__ X, int64_t ld, int nb, double *__restrict__ C) {
    const int a = blockIdx.y, b = blockIdx.x;
    double *c = C + (int64_t)a * kNBc * ld + (int64_t)b * kNBc;
    bool first = true;
    for (int kb = a > b ? a : b; kb < nb; ++kb) {
        tile_abt(X + (int64_t)a * kNBc * ld + (int64_t)kb * kNBc, ld, X + (int64_t)b * kNBc * ld + (int64_t)kb * kNBc, ld, c, ld, !first, 1.0);
        first = false;
        __syncthreads();
    }
}
}  // namespace

// The inverse of an SPD matrix on the matrix pipe.  Aw: (2 Mp) x Mp, Mp a multiple of 64 -- the lower triangle of A on top of an
// IDENTITY: the blocked Cholesky takes the identity along as border rows, which leaves L^-T there (the rows of a border X become
// X L^-T), and A^-1 = L^-T L^-1 is one product of that triangle with itself.  C: Mp x Mp, all of it written (exactly symmetric).
// *flag receives GINGR_ERR_NOT_SPD when a diagonal block fails.  Linv: (Mp / 64) blocks of 64 x 64.
void dense_spd_inverse(gingr_ctx *ctx, double *Aw, int64_t Mp, double *Linv, double *C, int32_t *flag) {
    blocked_cholesky(ctx, Aw, Mp, Linv, flag, true);
    const int nb = (int)(Mp / kNBc);
    hipLaunchKernelGGL(inverse_product_kernel, dim3((unsigned)nb, (unsigned)nb), dim3(256), 0, ctx->stream, Aw + Mp * Mp, Mp, nb, C);
}

namespace {

// ------------------------------------------------------------------------------------------------ optimal-step non-rigid ICP
// Normal equations of the stacked least-squares systems of NonRigidOptimalStepICP.scala (the reference solves `A \ B` on the sparse
// stack; A has full column rank, so the solution is the same).  With Lg = M^T M the graph Laplacian of the template's edges:
//   N-ICP-T (:151-190)   unknown X (n x 3):   (alpha^2 Lg + W^2 + E_L) X = W^2 (U - V) + E_L^T beta (UL - VL)
//                        E_L: the reference's A3 has its ones at (i, i), i < L -- the first L columns, unscaled by beta
//   N-ICP-A (:241-283)   unknown X (4n x 3):  (alpha^2 Lg (x) G^2 + D^T W^2 D + beta^2 DL^T DL) X = D^T W^2 U + beta^2 DL^T UL
//                        D row i = [p_i, 1] in the columns 4i .. 4i+3; W zeroed at the landmark vertices; G = diag(1, 1, 1, gamma)
// One thread per entry of the lower triangle / the border rows; the matrix is dense on the device (n of a registration template:
// thousands) and goes through the blocked MFMA Cholesky of the non-rigid CPD.
struct ScratchPart {  // a slice of the context's grow-only scratch
    void *p = nullptr;
    template <typename T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

struct NicpArgs {
    int64_t n, dim, Mp, n_edges;
    int kind;  // 0 T, 1 A
    const double *v;        // template, SoA [3][n]
    const double *u;        // closest points, SoA
    const double *w;        // weights (A: already zero at the landmark vertices)
    const int32_t *deg;     // edges per vertex
    const int32_t *lmcount; // landmarks mapped to the vertex (A) / 1 for the first L vertices (T)
    const double *lmsum;    // SoA [3][n]: A: sum of the landmark targets of the vertex; T: beta (UL_i - VL_i) for i < L
    double alpha2, beta2, gamma2;
};

__global__ __launch_bounds__(256) void nicp_fill_kernel(NicpArgs a, double *__restrict__ Aw) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.Mp) return;
    for (int64_t r = blockIdx.y; r < a.Mp + kNBc; r += gridDim.y) {
    double val = 0.0;
    if (r < a.dim) {
        if (a.kind == 0) {
            if (c == r) val = a.alpha2 * (double)a.deg[r] + a.w[r] * a.w[r] + (double)a.lmcount[r];
        } else if (c <= r && (c >> 2) == (r >> 2)) {  // the 4 x 4 block of vertex i
            const int64_t i = r >> 2;
            const int ra = (int)(r & 3), ca = (int)(c & 3);
            const double qr = ra < 3 ? a.v[ra * a.n + i] : 1.0, qc = ca < 3 ? a.v[ca * a.n + i] : 1.0;
            val = (a.w[i] * a.w[i] + a.beta2 * (double)a.lmcount[i]) * qr * qc;
            if (ra == ca) val += a.alpha2 * (double)a.deg[i] * (ra < 3 ? 1.0 : a.gamma2);
        }
    } else if (r < a.Mp) {
        val = c == r ? 1.0 : 0.0;  // padding
    } else if (r < a.Mp + 3 && c < a.dim) {
        const int d = (int)(r - a.Mp);
        if (a.kind == 0) {
            val = a.w[c] * a.w[c] * (a.u[d * a.n + c] - a.v[d * a.n + c]) + a.lmsum[d * a.n + c];
        } else {
            const int64_t i = c >> 2;
            const int ca = (int)(c & 3);
            const double qc = ca < 3 ? a.v[ca * a.n + i] : 1.0;
            val = qc * (a.w[i] * a.w[i] * a.u[d * a.n + i] + a.beta2 * a.lmsum[d * a.n + i]);
        }
    }
    Aw[r * a.Mp + c] = val;
    }
}

// the off-diagonal entries of alpha^2 Lg (x) G^2: -alpha^2 g_a^2 at ((p2, a), (p1, a)) for every edge p1 < p2
__global__ __launch_bounds__(256) void nicp_edges_kernel(NicpArgs a, const int32_t *__restrict__ edges, double *__restrict__ Aw) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n_edges) return;
    const int64_t p1 = edges[2 * e], p2 = edges[2 * e + 1];
    if (a.kind == 0) {
        Aw[p2 * a.Mp + p1] = -a.alpha2;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) Aw[(4 * p2 + k) * a.Mp + 4 * p1 + k] = -a.alpha2 * (k < 3 ? 1.0 : a.gamma2);
    }
}

// T: out = V + X;  A: out_i = [p_i, 1] X_i  (D X)
__global__ __launch_bounds__(256) void nicp_apply_kernel(NicpArgs a, const double *__restrict__ W, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        double r;
        if (a.kind == 0) {
            r = a.v[d * a.n + i] + W[d * a.Mp + i];
        } else {
            const double *x = W + d * a.Mp + 4 * i;
            r = ((a.v[i] * x[0] + a.v[a.n + i] * x[1]) + a.v[2 * a.n + i] * x[2]) + x[3];
        }
        out[d * a.n + i] = r;
    }
}

}  // namespace

struct gingr_classic_cpd {
    gingr_ctx *ctx = nullptr;
    int32_t kind = 0;
    int64_t M = 0, N = 0, Mp = 0;
    double lambda = 2.0, beta = 2.0, w = 0.0;
    DevBuf X, TY, den, inv_den, Pt1, P1, PX, ws, part, sc, aux, red, mu, params, flag, G, Aw, Linv, W, stage;
};

extern "C" {

void gingr_classic_cpd_destroy(gingr_classic_cpd *h) {
    if (!h) return;
    (void)hipSetDevice(h->ctx->device);
    delete h;
}

int gingr_classic_cpd_create(gingr_ctx *ctx, int32_t kind, int64_t M, const double *template_xyz, int64_t N, const double *target_xyz,
                             double lambda, double beta, double w, gingr_classic_cpd **out) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if (!out || !template_xyz || !target_xyz || M < 1 || N < 1 || kind < 0 || kind > 2)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "classic_cpd_create: bad argument");
    // CPDFactory's requirements (CPDFactory.scala:43-45); w = 1 divides by zero in the outlier constant
    if (!(w >= 0.0 && w < 1.0) || !(beta > 0.0) || !(lambda > 0.0))
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "classic_cpd_create: need 0 <= w < 1, beta > 0, lambda > 0");
    if (kind == 2 && M > 46000) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "classic_cpd_create: non-rigid template too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gingr_classic_cpd *h = new gingr_classic_cpd;
    h->ctx = ctx;
    h->kind = kind;
    h->M = M;
    h->N = N;
    h->Mp = round_up(M, kNBc);
    h->lambda = lambda;
    h->beta = beta;
    h->w = w;
    auto fail = [&](int rc) {
        gingr_classic_cpd_destroy(h);
        return rc;
    };
#define CC_TRY(expr)                                                                         \
    do {                                                                                     \
        if ((expr) != hipSuccess) return fail(gingr_set_error(ctx, GINGR_ERR_HIP, #expr));   \
    } while (0)
    const int64_t big = M > N ? M : N;
    CC_TRY(h->X.alloc((size_t)3 * N * sizeof(double)));
    CC_TRY(h->TY.alloc((size_t)3 * M * sizeof(double)));
    CC_TRY(h->stage.alloc((size_t)3 * big * sizeof(double)));
    CC_TRY(h->den.alloc((size_t)N * sizeof(double)));
    CC_TRY(h->inv_den.alloc((size_t)N * sizeof(double)));
    CC_TRY(h->Pt1.alloc((size_t)N * sizeof(double)));
    CC_TRY(h->P1.alloc((size_t)M * sizeof(double)));
    CC_TRY(h->PX.alloc((size_t)3 * M * sizeof(double)));
    const int64_t ws1 = cpd_colsum_ws_doubles(M, N), ws2 = cpd_rowstats_ws_doubles(M, N);
    CC_TRY(h->ws.alloc((size_t)std::max(std::max(ws1, ws2), sumsq_pairs_ws_doubles(M)) * sizeof(double)));
    CC_TRY(h->part.alloc(GINGR_SCALAR_PART * sizeof(double)));
    CC_TRY(h->sc.alloc(16 * sizeof(double)));
    CC_TRY(h->aux.alloc(GINGR_AUX * sizeof(double)));
    CC_TRY(h->red.alloc((size_t)kRedBlocks * 19 * sizeof(double)));
    CC_TRY(h->mu.alloc(6 * sizeof(double)));
    CC_TRY(h->params.alloc(16 * sizeof(double)));
    CC_TRY(h->flag.alloc(sizeof(int32_t)));
    CC_TRY(hipMemsetAsync(h->sc.p, 0, 16 * sizeof(double), ctx->stream));
    CC_TRY(hipMemsetAsync(h->params.p, 0, 16 * sizeof(double), ctx->stream));
    CC_TRY(hipMemsetAsync(h->flag.p, 0, sizeof(int32_t), ctx->stream));
    CC_TRY(hipMemcpyAsync(h->stage.p, target_xyz, (size_t)3 * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, h->stage.as<double>(), N, h->X.as<double>());
    CC_TRY(hipStreamSynchronize(ctx->stream));  // the staging buffer is reused
    CC_TRY(hipMemcpyAsync(h->stage.p, template_xyz, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, h->stage.as<double>(), M, h->TY.as<double>());
    const Cloud X = cloud_at(h->X.as<double>(), N), Y = cloud_at(h->TY.as<double>(), M);
    double *aux = h->aux.as<double>();
    launch_cloud_centroid(ctx, X, aux + 2);
    launch_cloud_absmax(ctx, X, aux + 2, aux);
    // sigma2_0 = sum |y_m - x_n|^2 / (dim N M)                                    (RigidCPD.initializeGaussianKernel, :46-57)
    launch_sumsq_pairs(ctx, Y, X, h->ws.as<double>(), h->sc.as<double>() + 9);
    double tot = 0.0;
    CC_TRY(hipMemcpyAsync(&tot, h->sc.as<double>() + 9, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CC_TRY(hipStreamSynchronize(ctx->stream));
    const double s0 = tot / (3.0 * (double)N * (double)M);
    CC_TRY(hipMemcpyAsync(h->sc.as<double>() + 8, &s0, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (kind == 2) {
        const int64_t Mp = h->Mp, nb = Mp / kNBc;
        CC_TRY(h->G.alloc((size_t)M * M * sizeof(double)));
        CC_TRY(h->Aw.alloc((size_t)(Mp + kNBc) * Mp * sizeof(double)));
        CC_TRY(h->Linv.alloc((size_t)nb * kNBc * kNBc * sizeof(double)));
        CC_TRY(h->W.alloc((size_t)3 * Mp * sizeof(double)));
        // G = exp(-|y_i - y_j|^2 / (2 beta^2)) over the TEMPLATE points                    (CPDFactory.initializeKernelMatrixG, :54-66)
        launch_gauss_block(ctx, Y, Y, sqrt(2.0) * beta, 1.0, h->G.as<double>());
    }
    CC_TRY(hipGetLastError());
    CC_TRY(hipStreamSynchronize(ctx->stream));
#undef CC_TRY
    *out = h;
    return GINGR_OK;
}

int gingr_classic_cpd_iterate(gingr_classic_cpd *h, int32_t n_iterations) {
    if (!h) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = h->ctx;
    if (n_iterations < 0) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "classic_cpd_iterate: negative count");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = h->M, N = h->N, Mp = h->Mp;
    const Cloud X = cloud_at(h->X.as<double>(), N), Y = cloud_at(h->TY.as<double>(), M);
    double *sc = h->sc.as<double>(), *s2 = sc + 8, *aux = h->aux.as<double>(), *ty = h->TY.as<double>();
    int32_t *flag = h->flag.as<int32_t>();
    for (int32_t it = 0; it < n_iterations; ++it) {
        // Expectation (RigidCPD.scala:90-105) as streaming statistics: P1, Pt1, P X, Np, xPx
        launch_cloud_absmax(ctx, Y, aux + 2, aux + 1);
        launch_cpd_colsum(ctx, Y, X, s2, aux, nullptr, h->ws.as<double>(), h->den.as<double>());
        launch_cpd_den_finalize(ctx, X, s2, h->w, M, h->den.as<double>(), h->inv_den.as<double>(), h->Pt1.as<double>(), nullptr,
                                h->part.as<double>(), sc);
        launch_cpd_rowstats(ctx, Y, X, s2, aux, h->inv_den.as<double>(), nullptr, nullptr, h->ws.as<double>(), h->P1.as<double>(),
                            h->PX.as<double>(), h->part.as<double>(), sc);
        if (h->kind != 2) {
            hipLaunchKernelGGL(means_kernel, dim3(kRedBlocks), dim3(256), 0, ctx->stream, X, h->Pt1.as<double>(), Y, h->P1.as<double>(),
                               h->red.as<double>());
            hipLaunchKernelGGL(means_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, h->red.as<double>(), sc, h->mu.as<double>());
            hipLaunchKernelGGL(moments_kernel, dim3(kRedBlocks), dim3(256), 0, ctx->stream, X, h->Pt1.as<double>(), Y, h->P1.as<double>(),
                               h->PX.as<double>(), h->mu.as<double>(), h->red.as<double>());
            hipLaunchKernelGGL(transform_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, h->red.as<double>(), h->mu.as<double>(), sc,
                               h->kind == 1 ? 1 : 0, h->params.as<double>(), flag);
            hipLaunchKernelGGL(apply_transform_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, M, ty,
                               h->params.as<double>());
        } else {
            double *Aw = h->Aw.as<double>(), *Linv = h->Linv.as<double>();
            hipLaunchKernelGGL(build_system_kernel, dim3((unsigned)ceil_div(Mp, 256), (unsigned)(Mp + kNBc)), dim3(256), 0, ctx->stream, M, Mp,
                               h->G.as<double>(), h->P1.as<double>(), h->PX.as<double>(), ty, sc, h->lambda, Aw);
            dense_spd_solve3(ctx, Aw, Mp, Linv, h->W.as<double>(), flag);
            hipLaunchKernelGGL(deform_kernel, dim3((unsigned)ceil_div(M, 16)), dim3(256), 0, ctx->stream, M, Mp, h->G.as<double>(),
                               h->W.as<double>(), ty);
            hipLaunchKernelGGL(nonrigid_sums_kernel, dim3(kRedBlocks), dim3(256), 0, ctx->stream, M, ty, h->P1.as<double>(),
                               h->PX.as<double>(), h->red.as<double>());
            hipLaunchKernelGGL(nonrigid_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, h->red.as<double>(), sc, flag);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    int32_t err = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&err, flag, sizeof(err), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (err) {
        HIP_TRY(ctx, hipMemsetAsync(flag, 0, sizeof(int32_t), ctx->stream));
        return gingr_set_error(ctx, err, "classic_cpd_iterate: %s", err == GINGR_ERR_NOT_SPD ? "system not positive definite" : "non-finite result");
    }
    return GINGR_OK;
}

int gingr_classic_cpd_get(gingr_classic_cpd *h, double *ty_xyz, double *sigma2, double *transform13, double *w_xyz) {
    if (!h) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = h->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = h->M, Mp = h->Mp;
    if (ty_xyz) {
        launch_soa_to_aos(ctx, h->TY.as<double>(), M, h->stage.as<double>());
        HIP_TRY(ctx, hipMemcpyAsync(ty_xyz, h->stage.p, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (sigma2) HIP_TRY(ctx, hipMemcpyAsync(sigma2, h->sc.as<double>() + 8, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (transform13) HIP_TRY(ctx, hipMemcpyAsync(transform13, h->params.p, 13 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (w_xyz) {
        if (h->kind != 2) return gingr_set_error(ctx, GINGR_ERR_STATE, "classic_cpd_get: W exists for the non-rigid kind only");
        std::vector<double> hw((size_t)3 * Mp);
        HIP_TRY(ctx, hipMemcpyAsync(hw.data(), h->W.p, hw.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int64_t i = 0; i < M; ++i)
            for (int d = 0; d < 3; ++d) w_xyz[3 * i + d] = hw[(size_t)d * Mp + i];
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GINGR_OK;
}

int gingr_classic_cpd_set(gingr_classic_cpd *h, const double *ty_xyz, double sigma2) {
    if (!h) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = h->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ty_xyz) {
        HIP_TRY(ctx, hipMemcpyAsync(h->stage.p, ty_xyz, (size_t)3 * h->M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        launch_aos_to_soa(ctx, h->stage.as<double>(), h->M, h->TY.as<double>());
    }
    HIP_TRY(ctx, hipMemcpyAsync(h->sc.as<double>() + 8, &sigma2, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GINGR_OK;
}

int gingr_nicp_solve(gingr_ctx *ctx, int32_t kind, int64_t n, const double *moving_xyz, int64_t n_edges, const int32_t *edges,
                     const double *w, const double *cp_xyz, int32_t n_lm, const int32_t *lm_ids, const double *lm_target_xyz, double alpha,
                     double beta, double gamma, double *out_xyz, double *out_lm_xyz) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if ((kind != 0 && kind != 1) || n < 1 || n_edges < 0 || n_lm < 0 || !moving_xyz || !w || !cp_xyz || !out_xyz || (n_edges > 0 && !edges) ||
        (n_lm > 0 && (!lm_ids || !lm_target_xyz)) || !(alpha >= 0.0) || !(beta >= 0.0) || !(gamma >= 0.0) || n_lm > n)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "nicp_solve: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t dim = kind == 0 ? n : 4 * n, Mp = round_up(dim, kNBc), nb = Mp / kNBc;
    // host side: SoA clouds, vertex degrees, the landmark terms per vertex
    std::vector<double> hv((size_t)3 * n), hu((size_t)3 * n), hw((size_t)n), hs((size_t)3 * n, 0.0);
    std::vector<int32_t> hdeg((size_t)n, 0), hcnt((size_t)n, 0);
    for (int64_t i = 0; i < n; ++i) {
        for (int d = 0; d < 3; ++d) {
            hv[(size_t)(d * n + i)] = moving_xyz[3 * i + d];
            hu[(size_t)(d * n + i)] = cp_xyz[3 * i + d];
        }
        hw[(size_t)i] = w[i];
    }
    for (int64_t e = 0; e < n_edges; ++e) {
        const int32_t p1 = edges[2 * e], p2 = edges[2 * e + 1];
        if (p1 < 0 || p2 <= p1 || p2 >= n) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "nicp_solve: edge %lld is not p1 < p2 < n", (long long)e);
        ++hdeg[(size_t)p1];
        ++hdeg[(size_t)p2];
    }
    for (int32_t l = 0; l < n_lm; ++l) {
        const int32_t id = lm_ids[l];
        if (id < 0 || id >= n) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "nicp_solve: landmark id out of range");
        if (kind == 0) {  // row l of A3 has its one in COLUMN l; B3 row l = beta (UL_l - V[id_l])
            hcnt[(size_t)l] = 1;
            for (int d = 0; d < 3; ++d) hs[(size_t)(d * n + l)] = beta * (lm_target_xyz[3 * l + d] - moving_xyz[3 * (int64_t)id + d]);
        } else {
            hw[(size_t)id] = 0.0;  // W(i, i) = 0 at the landmark vertices (:251-253)
            ++hcnt[(size_t)id];
            for (int d = 0; d < 3; ++d) hs[(size_t)(d * n + id)] += lm_target_xyz[3 * l + d];
        }
    }
    // one grow-only scratch of the context, carved up (an N-ICP run calls this once per iteration with the same sizes)
    ScratchPart dv, du, dw, ds, ddeg, dcnt, dedges, Aw, Linv, W, flag, dout;
    {
        const size_t sizes[12] = {hv.size() * 8, hu.size() * 8, hw.size() * 8, hs.size() * 8, hdeg.size() * 4, hcnt.size() * 4,
                                  (size_t)(n_edges > 0 ? n_edges : 1) * 8, (size_t)(Mp + kNBc) * Mp * sizeof(double),
                                  (size_t)nb * kNBc * kNBc * sizeof(double), (size_t)3 * Mp * sizeof(double), sizeof(int32_t),
                                  (size_t)3 * n * sizeof(double)};
        ScratchPart *parts[12] = {&dv, &du, &dw, &ds, &ddeg, &dcnt, &dedges, &Aw, &Linv, &W, &flag, &dout};
        size_t total = 0;
        for (int q = 0; q < 12; ++q) total += (sizes[q] + 255) & ~(size_t)255;
        if (ctx->scratch_bytes < total) {
            if (ctx->scratch) (void)hipFree(ctx->scratch);
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
            if (hipMalloc(&ctx->scratch, total) != hipSuccess) {
                (void)hipGetLastError();
                ctx->scratch = nullptr;
                return gingr_set_error(ctx, GINGR_ERR_HIP, "nicp_solve: out of device memory (the system is dense: %lld x %lld doubles)",
                                       (long long)Mp, (long long)Mp);
            }
            ctx->scratch_bytes = total;
        }
        size_t off = 0;
        for (int q = 0; q < 12; ++q) {
            parts[q]->p = static_cast<char *>(ctx->scratch) + off;
            off += (sizes[q] + 255) & ~(size_t)255;
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(dv.p, hv.data(), hv.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(du.p, hu.data(), hu.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dw.p, hw.data(), hw.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ds.p, hs.data(), hs.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ddeg.p, hdeg.data(), hdeg.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dcnt.p, hcnt.data(), hcnt.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n_edges > 0) HIP_TRY(ctx, hipMemcpyAsync(dedges.p, edges, (size_t)n_edges * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(flag.p, 0, sizeof(int32_t), ctx->stream));
    NicpArgs a{n, dim, Mp, n_edges, kind, dv.as<double>(), du.as<double>(), dw.as<double>(), ddeg.as<int32_t>(), dcnt.as<int32_t>(),
               ds.as<double>(), alpha * alpha, beta * beta, gamma * gamma};
    hipLaunchKernelGGL(nicp_fill_kernel, dim3((unsigned)ceil_div(Mp, 256), (unsigned)std::min<int64_t>(Mp + kNBc, 65535)), dim3(256), 0, ctx->stream, a, Aw.as<double>());
    if (n_edges > 0)
        hipLaunchKernelGGL(nicp_edges_kernel, dim3((unsigned)ceil_div(n_edges, 256)), dim3(256), 0, ctx->stream, a, dedges.as<int32_t>(),
                           Aw.as<double>());
    dense_spd_solve3(ctx, Aw.as<double>(), Mp, Linv.as<double>(), W.as<double>(), flag.as<int32_t>());
    hipLaunchKernelGGL(nicp_apply_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, a, W.as<double>(), dout.as<double>());
    HIP_TRY(ctx, hipGetLastError());
    std::vector<double> ho((size_t)3 * n);
    int32_t err = 0;
    HIP_TRY(ctx, hipMemcpyAsync(ho.data(), dout.p, ho.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&err, flag.p, sizeof(err), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (err) return gingr_set_error(ctx, GINGR_ERR_NOT_SPD, "nicp_solve: the normal equations are not positive definite (a mesh component without any weighted vertex or landmark)");
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            const double val = ho[(size_t)(d * n + i)];
            if (!std::isfinite(val)) return gingr_set_error(ctx, GINGR_ERR_NONFINITE, "nicp_solve: non-finite result");
            out_xyz[3 * i + d] = val;
        }
    if (out_lm_xyz)  // N-ICP-A: DL X = the moved landmark vertices (:278-282); N-ICP-T has no such output: the moved vertices too
        for (int32_t l = 0; l < n_lm; ++l)
            for (int d = 0; d < 3; ++d) out_lm_xyz[3 * l + d] = out_xyz[3 * (int64_t)lm_ids[l] + d];
    return GINGR_OK;
}

}  // extern "C"

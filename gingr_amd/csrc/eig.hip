// Eigen-decomposition of a small symmetric positive semi-definite matrix (n <= 192) with the matrix in REGISTERS.
//
// What it replaces: the breeze eigSym(L^T L) inside scalismo's LowRankGaussianProcess.approximateGPCholesky (the Karhunen-Loeve
// basis of a kernel model: G/api/registration/utils/GPMMHelper.scala:39-69, E/CreateBunnyGPMM.scala), and the eigenbasis of the
// model's moment S_tot = Q^T Q the uniform-weight posterior uses (gp.hip posterior_solve_eig_kernel).
//
// The two-sided cyclic Jacobi it takes over from (gpmm.hip jacobi_eig_kernel, still the path above n = 192 and for numerically
// singular matrices) keeps A and V in global memory and spends a round trip to L2 per load-store of every rotation round: 40 us a
// round at n = 171, 61 ms a decomposition.  Here:
//   * G = C C^T first (right-looking Cholesky, a column at a time through LDS), then the ONE-SIDED (Hestenes) Jacobi on the columns
//     of C: rotations from the right make the columns of C V mutually orthogonal, so they are the eigenvectors of G scaled by the
//     roots of the eigenvalues -- eigenvalue = squared length, eigenvector = normalised column, no V to accumulate, and the
//     accuracy is governed by the condition number of C (the root of G's).  A failed pivot leaves the columns of G itself to rotate
//     (eigenvalue = length), and info[1] reports the matrix as numerically singular: the callers then take the two-sided kernel,
//     whose V starts from the identity and is complete whatever the spectrum.
//   * a column lives in the registers of ONE wave, E values per lane (row = e * 64 + lane); a wave owns P pairs of columns;
//   * a round = ONE dot product per pair (the squared norms travel with the columns, updated by the rotation formulas and
//     recomputed once a sweep), summed for all P pairs at once by v_permlane32_swap / v_permlane16_swap + DPP so that the lanes
//     8 j .. 8 j + 7 hold pair j's; these lanes work out the rotation (two v_rsq_f64 with Newton steps instead of three divisions
//     and three roots), which reaches the wave through scalar registers; then the round-robin tournament moves every column one
//     place: inside a wave a register rename, between neighbouring waves one column each way through LDS -- one barrier per round,
//     two LDS buffers;
//   * 16 waves x 6 pairs x 2 columns = 192 columns of 192 rows at most; up to three problems per launch, one workgroup each.
// Rotation threshold |p.q| > sqrt(n) eps |p| |q| (LAPACK dgesvj's), at most 60 sweeps (8 at n = 171; the two-sided kernel needs 10).
// n = 34 / 100 / 171: 0.26 / 1.3 / 3.2 ms (tools/ubench_sym_eig.hip, profiles/r06_ubench_sym_eig.txt).
#include "gp.h"

namespace {

constexpr int kEigSlots = 192;  // columns a workgroup holds at most (2 x waves x pairs per wave)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// sum over the 64 lanes, the same bits in every lane (every step adds a value to its mirror image: a + b == b + a)
__device__ __forceinline__ double wave_allsum(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}


__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// x[l] + x[l ^ 32] of `lo` in the lanes 0 .. 31, of `hi` in the lanes 32 .. 63 (v_permlane32_swap: no LDS round trip)
__device__ __forceinline__ double halves_sum32(double lo, double hi) {
    const auto a = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(lo), (unsigned)__double2loint(hi), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(lo), (unsigned)__double2hiint(hi), false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// x[l] + x[l ^ 16] of `lo` in the rows 0 and 2 (lanes with bit 4 clear), of `hi` in the rows 1 and 3
__device__ __forceinline__ double halves_sum16(double lo, double hi) {
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(lo), (unsigned)__double2loint(hi), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(lo), (unsigned)__double2hiint(hi), false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// v[j], j < P <= 8: one value per pair and lane.  Returns, in the lanes 8 j .. 8 j + 7, the sum of v[j] over the wave (the same bits
// in all eight): halving exchanges over the lane bits 5, 4, 3 (a lane keeps the half of the values its bit selects), then an
// all-reduce over the eight lanes by DPP.
template <int P>
__device__ __forceinline__ double pair_sums(const double (&v)[8], int lane) {
    const bool b3 = lane & 8;
    double u4[4], u2[2], u1;
#pragma unroll
    for (int k = 0; k < 4; ++k) u4[k] = P > 4 ? halves_sum32(v[k], v[4 + k]) : halves_sum32(v[k], v[k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) u2[k] = P > 2 ? halves_sum16(u4[k], u4[2 + k]) : halves_sum16(u4[k], u4[k]);
    {
        const double mine = b3 ? u2[1] : u2[0], theirs = b3 ? u2[0] : u2[1];
        u1 = mine + dpp_f64<0x128>(theirs);  // row_ror:8
    }
    u1 += dpp_f64<0xB1>(u1);   // quad_perm [1,0,3,2]
    u1 += dpp_f64<0x4E>(u1);   // quad_perm [2,3,0,1]
    u1 += dpp_f64<0x141>(u1);  // row_half_mirror
    return u1;
}

// 1 / sqrt(x), x > 0: v_rsq_f64 and two Newton steps (a few ulp; the rotation only needs c^2 + s^2 = 1 to rounding)
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-(h * y), y, 0.5);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

struct EigProblem {
    const double *G;  // n x n, row stride ldg (symmetric: column c is read as row c)
    int32_t ldg, n;
    double *H;       // work: [2 K][n] final columns, K = waves * P positions
    double *norms;   // work: [2 K]
    double *evals;   // out: [n] descending
    double *Vs;      // out: [n][n], eigenvector k in column k
    int32_t *info;   // out: [0] sweeps used, [1] 1 when numerically singular
};
struct EigBatch {
    EigProblem p[3];
};

template <int E, int P, int W>
__global__ __launch_bounds__(W * 64) void sym_eig_cols_kernel(EigBatch batch) {
    __shared__ double xT[2][W][E * 64], xB[2][W][E * 64];
    __shared__ double nrm[2][2][W * P];  // squared column norms by slot: [.][0] top row, [.][1] bottom row
    __shared__ int32_t rank_of[2 * W * P];
    const EigProblem pr = batch.p[blockIdx.x];
    const int n = pr.n;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // w: scalar
    const int positions = (n + 1) / 2;
    const int waves = (positions + P - 1) / P;  // waves in the ring
    const int K = waves * P;
    const bool in_ring = w < waves;
    const bool last = w == waves - 1;
    const int grp = lane >> 3;                   // the pair whose rotation this lane works out
    const int g = w * P + (grp < P ? grp : 0);   // its slot in the ring
    double top[P][E], bot[P][E];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int ct = w * P + j, cb = K + w * P + j;  // column of G that starts in the slot
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int row = e * 64 + lane;
            top[j][e] = (in_ring && ct < n && row < n) ? pr.G[(int64_t)ct * pr.ldg + row] : 0.0;
            bot[j][e] = (in_ring && cb < n && row < n) ? pr.G[(int64_t)cb * pr.ldg + row] : 0.0;
        }
    }
    // G = C C^T first (right-looking, a column at a time through LDS): the rotations then work on the columns of C, whose
    // condition number is the square root of G's; a pivot that fails (G numerically singular) leaves the columns of G to rotate
    __shared__ double colbuf[2][E * 64];
    __shared__ int32_t chol_failed;
    if (threadIdx.x == 0) chol_failed = 0;
    __syncthreads();
    auto pivot_column = [&](double (&col)[E], int k) {  // column k: scaled by 1 / sqrt(pivot), rows above the pivot zeroed, published
        double dsel = col[0];
#pragma unroll
        for (int e = 1; e < E; ++e)
            if ((k >> 6) == e) dsel = col[e];
        const double d = readlane_f64(dsel, k & 63);
        const bool ok = d > 0.0 && d <= 1.79769313486231570815e308;
        const double rs = ok ? 1.0 / sqrt(d) : 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int row = e * 64 + lane;
            col[e] = row >= k ? col[e] * rs : 0.0;
            colbuf[k & 1][row] = col[e];
        }
        if (!ok && lane == 0) chol_failed = 1;
    };
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if (k < K && w * P + j == k) pivot_column(top[j], k);
            if (k >= K && K + w * P + j == k) pivot_column(bot[j], k);
        }
        __syncthreads();
        if (in_ring) {
            double cb[E];
#pragma unroll
            for (int e = 0; e < E; ++e) cb[e] = colbuf[k & 1][e * 64 + lane];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int ct = w * P + j, cbt = K + w * P + j;
                if (ct > k && ct < n) {
                    const double f = colbuf[k & 1][ct];
#pragma unroll
                    for (int e = 0; e < E; ++e) top[j][e] = __builtin_fma(-cb[e], f, top[j][e]);
                }
                if (cbt > k && cbt < n) {
                    const double f = colbuf[k & 1][cbt];
#pragma unroll
                    for (int e = 0; e < E; ++e) bot[j][e] = __builtin_fma(-cb[e], f, bot[j][e]);
                }
            }
        }
    }
    __syncthreads();
    const bool factored = __builtin_amdgcn_readfirstlane(chol_failed) == 0;
    if (factored) {  // the strict upper triangle never held the factor
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int row = e * 64 + lane;
                if (row < w * P + j) top[j][e] = 0.0;
                if (row < K + w * P + j) bot[j][e] = 0.0;
            }
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int ct = w * P + j, cbt = K + w * P + j;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int row = e * 64 + lane;
                top[j][e] = (in_ring && ct < n && row < n) ? pr.G[(int64_t)ct * pr.ldg + row] : 0.0;
                bot[j][e] = (in_ring && cbt < n && row < n) ? pr.G[(int64_t)cbt * pr.ldg + row] : 0.0;
            }
        }
    }
    const double tol2 = (double)n * 2.220446049250313e-16 * 2.220446049250313e-16;
    const int rounds = 2 * K - 1;
    int par = 0, sweep = 0;
    double app = 0.0, aqq = 0.0;  // squared norms of the lane's pair: exact at the start of a sweep, updated by the rotations
    for (; sweep < 60 && n > 1; ++sweep) {
        int rotated = 0;
        for (int rd = 0; rd < rounds; ++rd) {
            if (in_ring) {
                // p . q of every pair, summed so that the lanes 8 j .. 8 j + 7 hold that of pair j; these lanes work out the
                // rotation (all P of them at once), which then reaches the whole wave through scalar registers
                double vpq[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    vpq[j] = 0.0;
                    if (j < P) {
#pragma unroll
                        for (int e = 0; e < E; ++e) vpq[j] = __builtin_fma(top[j][e], bot[j][e], vpq[j]);
                    }
                }
                const double apq = pair_sums<P>(vpq, lane);
                if (rd == 0) {
#pragma unroll 1
                    for (int pass = 0; pass < 2; ++pass) {  // one row of the ring at a time: 16 registers instead of 32
                        double vxx[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            vxx[j] = 0.0;
                            if (j < P) {
#pragma unroll
                                for (int e = 0; e < E; ++e) {
                                    const double x = pass ? bot[j][e] : top[j][e];
                                    vxx[j] = __builtin_fma(x, x, vxx[j]);
                                }
                            }
                        }
                        const double sum = pair_sums<P>(vxx, lane);
                        if (pass)
                            aqq = sum;
                        else
                            app = sum;
                    }
                }
                // tan 2 theta = 2 apq / (aqq - app), |theta| <= pi / 4:  cos 2 theta = |alpha| / r,  sin 2 theta = sign(alpha) beta / r
                double c = 1.0, s = 0.0;
                const double alpha = aqq - app, beta = 2.0 * apq;
                const double r2 = __builtin_fma(alpha, alpha, beta * beta);
                if (apq * apq > tol2 * (app * aqq) && r2 > 0.0) {
                    const double ir = rsqrt_nr(r2);
                    const double c2 = __builtin_fma(0.5 * fabs(alpha), ir, 0.5);
                    const double ic = rsqrt_nr(c2);
                    c = c2 * ic;
                    s = (alpha >= 0.0 ? 0.5 : -0.5) * beta * ir * ic;
                    const double t = s * ic;
                    app = __builtin_fma(-t, apq, app);
                    aqq = __builtin_fma(t, apq, aqq);
                }
                if ((lane & 7) == 0 && grp < P) {
                    nrm[par][0][g] = app;
                    nrm[par][1][g] = aqq;
                }
                // the two columns that leave the wave first
#pragma unroll
                for (int jj = 0; jj < P; ++jj) {
                    const int j = jj == 0 ? P - 1 : jj - 1;
                    const double cj = readlane_f64(c, 8 * j), sj = readlane_f64(s, 8 * j);
                    if (sj != 0.0) {  // wave uniform
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const double p = top[j][e], q = bot[j][e];
                            top[j][e] = __builtin_fma(cj, p, -(sj * q));
                            bot[j][e] = __builtin_fma(sj, p, cj * q);
                        }
                        rotated = 1;
                    }
                    if (jj == 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e) xT[par][w][e * 64 + lane] = top[P - 1][e];
                    }
                    if (j == 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e) xB[par][w][e * 64 + lane] = bot[0][e];
                    }
                }
            }
            // tournament: slot 0 of the top row stays, everything else moves one place round the ring
            __syncthreads();
            if (in_ring) {
                double inT[E], inB[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    // top slot w P: from the top row's slot w P - 1 (the neighbour's last)
                    inT[e] = w == 0 ? 0.0 : xT[par][w - 1][e * 64 + lane];
                    // bottom slot w P + P - 1: from the bottom row's next slot; the ring's last takes the top row's last
                    inB[e] = last ? top[P - 1][e] : xB[par][w + 1][e * 64 + lane];
                }
#pragma unroll
                for (int j = P - 1; j >= 1; --j)
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool from_bottom = w == 0 && j == 1;  // slot 1 of the ring takes the bottom row's slot 0
                        top[j][e] = from_bottom ? bot[0][e] : top[j - 1][e];
                    }
                if (w != 0) {
#pragma unroll
                    for (int e = 0; e < E; ++e) top[0][e] = inT[e];
                }
#pragma unroll
                for (int j = 0; j < P - 1; ++j)
#pragma unroll
                    for (int e = 0; e < E; ++e) bot[j][e] = bot[j + 1][e];
#pragma unroll
                for (int e = 0; e < E; ++e) bot[P - 1][e] = inB[e];
                // the norms follow their columns
                app = g == 0 ? nrm[par][0][0] : (g == 1 ? nrm[par][1][0] : nrm[par][0][g - 1]);
                aqq = g == K - 1 ? nrm[par][0][K - 1] : nrm[par][1][g + 1];
            }
            par ^= 1;
        }
        if (!__syncthreads_or(rotated)) {
            ++sweep;
            break;
        }
    }
    // columns and their norms to the work area; rank sort (descending, ties in slot order); normalised columns out
    if (in_ring) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int st = w * P + j, sb = K + w * P + j;
            double nt = 0.0, nb = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                nt = __builtin_fma(top[j][e], top[j][e], nt);
                nb = __builtin_fma(bot[j][e], bot[j][e], nb);
                const int row = e * 64 + lane;
                if (row < n) {
                    pr.H[(int64_t)st * n + row] = top[j][e];
                    pr.H[(int64_t)sb * n + row] = bot[j][e];
                }
            }
            nt = wave_allsum(nt);
            nb = wave_allsum(nb);
            if (lane == 0) {  // eigenvalue: the squared length of a column of C V, the length of one of G V
                pr.norms[st] = factored ? nt : sqrt(nt);
                pr.norms[sb] = factored ? nb : sqrt(nb);
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    const int slots = 2 * K;
    for (int i = threadIdx.x; i < slots; i += W * 64) {
        const double li = pr.norms[i];
        int rank = 0;
        for (int j = 0; j < slots; ++j) {
            const double lj = pr.norms[j];
            rank += (lj > li || (lj == li && j < i)) ? 1 : 0;
        }
        rank_of[i] = rank;
        if (rank < n) pr.evals[rank] = li;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double mx = 0.0, mn = __builtin_huge_val();
        for (int i = 0; i < slots; ++i)
            if (rank_of[i] < n) {
                mx = fmax(mx, pr.norms[i]);
                mn = fmin(mn, pr.norms[i]);
            }
        pr.info[0] = sweep;
        pr.info[1] = (!factored || !(mn > (double)n * 2.220446049250313e-16 * mx)) ? 1 : 0;
    }
    for (int idx = threadIdx.x; idx < slots * n; idx += W * 64) {
        const int i = idx / n, row = idx - i * n;
        const int rank = rank_of[i];
        if (rank >= n) continue;
        const double len = factored ? sqrt(pr.norms[i]) : pr.norms[i];
        pr.Vs[(int64_t)row * n + rank] = len > 0.0 ? pr.H[idx] / len : 0.0;
    }
}

}  // namespace

int64_t sym_eig_cols_work_doubles(int32_t n) {
    return (int64_t)kEigSlots * n + kEigSlots;
}

// up to three decompositions side by side (one workgroup each); every n in 1 .. kSymEigColsMaxN.  work[q]: sym_eig_cols_work_doubles(n[q])
// doubles; info[q]: two int32 (sweeps, singular flag).
void launch_sym_eig_cols(gingr_ctx *ctx, int count, const double *const *G, const int32_t *ldg, const int32_t *n, double *const *work,
                         double *const *evals, double *const *Vs, int32_t *const *info) {
    EigBatch b;
    int32_t nmax = 0;
    for (int q = 0; q < count; ++q) {
        b.p[q] = EigProblem{G[q], ldg[q], n[q], work[q], work[q] + (int64_t)kEigSlots * n[q], evals[q], Vs[q], info[q]};
        nmax = n[q] > nmax ? n[q] : nmax;
    }
    for (int q = count; q < 3; ++q) b.p[q] = b.p[0];
    if (nmax <= 64)
        hipLaunchKernelGGL((sym_eig_cols_kernel<1, 2, 16>), dim3(count), dim3(16 * 64), 0, ctx->stream, b);
    else if (nmax <= 128)
        hipLaunchKernelGGL((sym_eig_cols_kernel<2, 4, 16>), dim3(count), dim3(16 * 64), 0, ctx->stream, b);
    else
        hipLaunchKernelGGL((sym_eig_cols_kernel<3, 6, 16>), dim3(count), dim3(16 * 64), 0, ctx->stream, b);
}

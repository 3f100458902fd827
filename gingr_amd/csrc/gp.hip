// Low-rank Gaussian-process kernels of the GiNGR update for gfx950 (MI355X).
//
// What they replace (the arithmetic lives in scalismo 1.0-RC1, reached from G/api/GingrAlgorithm.scala):
//   gram_kernel       Q^T L Q of DiscreteLowRankGaussianProcess.regression (GingrAlgorithm.scala:300) -- the only
//                     GEMM-shaped op of the path; float64 MFMA (v_mfma_f64_16x16x4_f64), split over row slabs
//   sweep_kernel      every pass over the 3M x r basis: Q^T L (y - m) (:300), coefficients (:215,236),
//                     instance (:222,224, ModelFittingParameters.scala:134), Umeyama partial sums (:260-279);
//                     HBM-bound streaming of Q0, fused with the pose / projection epilogues
//   posterior_solve   pinv(QtL Q + I) * QtL (y - m)  -> Cholesky solve of the SPD matrix I + G
//   post_solve_kernel LandmarkRegistration.rigid3D/similarity3DLandmarkRegistration (3x3 SVD + Euler round trip) from the moments,
//                     second projection, the state hand-over of update (:239-246) incl. the Try-failure paths and the retry
//                     counter of the probabilistic proposal (:194-210,248,251)
//
// Layout: Q0 is row-major [3M][rp]: the 3 x rp block of one point is contiguous (2.7 KB at r = 100), so one point's
// observation weight, rotation and epilogue touch one contiguous block; rp = rank rounded up to 16 (MFMA tile).
// All reductions across workgroups go through per-block partials combined in a fixed order (bitwise reproducible).
#include "gp.h"
#include "svd3.h"

#include <algorithm>
#include <type_traits>

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// rotation conventions (euler_to_rot / rot_to_euler): svd3.h

__device__ __forceinline__ bool finite_d(double v) { return fabs(v) <= 1.79769313486231570815e308; }

// Basis of a model on a NEW reference whose every point takes a fixed convex combination of three source points (nearest
// neighbour: weights (1,0,0); triangle-mesh interpolation: barycentric weights of the closest surface point):
//   Q0_new[(3 s + d) rp + q] = sum_k w[3 o + k] Q0_src[(3 inv_src[ids[3 o + k]] + d) rp + q],   o = row_begin + perm_new[s]
__global__ __launch_bounds__(256) void interp_pack_kernel(const double *__restrict__ Qs, int32_t rp, const int32_t *__restrict__ inv_src,
                                                          const int32_t *__restrict__ ids, const double *__restrict__ w,
                                                          const int32_t *__restrict__ perm_new, int64_t row_begin, int64_t M,
                                                          double *__restrict__ Q0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * M * rp) return;
    const int64_t row = idx / rp;
    const int32_t q = (int32_t)(idx - row * rp);
    const int64_t s = row / 3;
    const int d = (int)(row - 3 * s);
    const int64_t o = row_begin + perm_new[s];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double wk = w[3 * o + k];
        if (wk != 0.0) acc += wk * Qs[((int64_t)3 * inv_src[ids[3 * o + k]] + d) * rp + q];
    }
    Q0[idx] = acc;
}

// ------------------------------------------------------------------------------------------------- basis packing
__global__ void pack_basis_kernel(const double *__restrict__ stage, const double *__restrict__ variance, int64_t rows,
                                  int32_t r, int32_t rp, const int32_t *__restrict__ perm, double *__restrict__ Q0) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * rp) return;
    const int64_t row = idx / rp;
    const int32_t k = (int32_t)(idx - row * rp);
    const int64_t pt = row / 3, d = row - 3 * pt;
    const int64_t src = 3 * (int64_t)(perm ? perm[pt] : pt) + d;
    // Q(i, j) = eigenvector * sqrt(eigenvalue)   (scalismo genericRegressionComputations)
    Q0[idx] = k < r ? stage[(int64_t)k * rows + src] * sqrt(variance[k]) : 0.0;
}

// ------------------------------------------------------------------------------------------------- basis sweeps
constexpr int kSweepThreads = 256;
constexpr int kGroups = kSweepThreads / 16;  // points per block step
constexpr int kSweepMaxBlocks = 1024;

// sum over the caller's 16-lane group, the same bits in every lane: the butterfly 8, 4, 2, 1.  Round 4: on the DPP crossbar (row
// rotations) instead of __shfl_xor, which goes through ds_bpermute (~100 cycles per step; twelve of these sums sit on the critical
// path of a one-wave-per-SIMD launch like the fit pass of a small model).  Bit-identical to the shuffle butterfly: after the step
// with distance 2d every lane holds the same bits as the lane 2d away, so the lane d "behind" (what a rotation delivers) holds
// exactly what the xor partner holds, and a + b = b + a.
__device__ __forceinline__ double group16_sum(double v) {
    auto step = [&](auto ctrl) {
        constexpr int c = decltype(ctrl)::value;
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, c, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), c, 0xf, 0xf, false);
        v += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    };
    step(std::integral_constant<int, 0x128>{});  // row_ror:8
    step(std::integral_constant<int, 0x124>{});  // row_ror:4
    step(std::integral_constant<int, 0x122>{});  // row_ror:2
    step(std::integral_constant<int, 0x121>{});  // row_ror:1
    return v;
}

template <int MODE, int KMAX>
__global__ __launch_bounds__(kSweepThreads) void sweep_kernel(SweepArgs a) {
    constexpr bool FWD = (MODE == SWEEP_PROJ1 || MODE == SWEEP_SHAPES || MODE == SWEEP_FIT || MODE == SWEEP_POSED);
    constexpr bool FWD2 = (MODE == SWEEP_SHAPES);
    constexpr bool TRANS = (MODE == SWEEP_RHS || MODE == SWEEP_PROJ1 || MODE == SWEEP_PROJ2 || MODE == SWEEP_RHS_ICP);
    extern __shared__ double lds[];  // [2*rp] coefficients, then [kGroups*rp] reduction scratch
    const int tid = threadIdx.x, lane16 = tid & 15, grp = tid >> 4;
    const int rp = a.rp, km = rp >> 4;
    const int64_t M = a.M;
    double *coef = lds;
    double *red = lds + 2 * rp;
    if (MODE == SWEEP_RHS && !gate_open(a.gate)) return;  // (workgroup-uniform)
    if (a.zero_slot && blockIdx.x == 0 && tid == 0) *a.zero_slot = 0.0;  // e.g. the |coordinate| maximum of the fit this pass rewrites
    if (FWD) {
        for (int k = tid; k < rp; k += kSweepThreads) {
            coef[k] = a.coef0[k];
            if (FWD2) coef[rp + k] = a.coef1[k];
        }
        __syncthreads();
    }
    // pose scalars (wave-uniform loads)
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tr[3] = {0, 0, 0}, cen[3] = {0, 0, 0}, scale = 1.0;
    if (MODE == SWEEP_SHAPES || MODE == SWEEP_FIT || MODE == SWEEP_POSED || MODE == SWEEP_RHS_ICP) {
        for (int q = 0; q < 9; ++q) R[q] = a.state->R[q];
        for (int q = 0; q < 3; ++q) {
            tr[q] = a.state->t[q];
            cen[q] = a.state->center[q];
        }
        scale = a.state->scale;
    } else if (MODE == SWEEP_PROJ2) {
        if (a.frame) {  // the rigid part of a device state directly (scale 1): no pose object has to be filled first
            for (int q = 0; q < 9; ++q) R[q] = a.frame->R[q];
            for (int q = 0; q < 3; ++q) {
                tr[q] = a.frame->t[q];
                cen[q] = a.frame->center[q];
            }
        } else {
            for (int q = 0; q < 9; ++q) R[q] = a.pose->R[q];
            for (int q = 0; q < 3; ++q) {
                tr[q] = a.pose->t[q];
                cen[q] = a.pose->center[q];
            }
        }
    }
    double acc[KMAX];
#pragma unroll
    for (int m = 0; m < KMAX; ++m) acc[m] = 0.0;
    double us[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) us[s] = 0.0;

    for (int64_t base = (int64_t)blockIdx.x * kGroups; base < M; base += (int64_t)gridDim.x * kGroups) {
        const int64_t p = base + grp;
        const bool valid = p < M;
        const int64_t pc = valid ? p : 0;
        const double *q0 = a.Q0 + (3 * pc) * rp + lane16;
        const double *q1 = q0 + rp;
        const double *q2 = q1 + rp;
        double f0[3] = {0, 0, 0}, f1[3] = {0, 0, 0};
        if (FWD) {
#pragma unroll 4
            for (int m = 0; m < km; ++m) {
                const int k = m * 16;
                const double c0 = coef[k + lane16];
                const double u0 = q0[k], u1 = q1[k], u2 = q2[k];
                f0[0] = __builtin_fma(u0, c0, f0[0]);
                f0[1] = __builtin_fma(u1, c0, f0[1]);
                f0[2] = __builtin_fma(u2, c0, f0[2]);
                if (FWD2) {
                    const double c1 = coef[rp + k + lane16];
                    f1[0] = __builtin_fma(u0, c1, f1[0]);
                    f1[1] = __builtin_fma(u1, c1, f1[1]);
                    f1[2] = __builtin_fma(u2, c1, f1[2]);
                }
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                f0[d] = group16_sum(f0[d]);
                if (FWD2) f1[d] = group16_sum(f1[d]);
            }
        }
        double e[3] = {0, 0, 0};
        const double rx = a.ref[pc], ry = a.ref[M + pc], rz = a.ref[2 * M + pc];
        const double mx = a.mean[pc], my = a.mean[M + pc], mz = a.mean[2 * M + pc];
        if (MODE == SWEEP_RHS) {
            e[0] = a.evec[pc];
            e[1] = a.evec[M + pc];
            e[2] = a.evec[2 * M + pc];
        } else if (MODE == SWEEP_RHS_ICP) {
            // the observation of obs_points_kernel (ICP branch), formed here: one launch and one round trip of e less per iteration.
            // Same expressions, so the same e; weight and e are still written out for whoever reads them later.
            const int32_t j = a.icp_idx[pc];
            const bool inside = j >= 0 && (int64_t)j < a.n_targets;  // -1: no finite distance (non-finite fit)
            const int32_t jc = inside ? j : 0;
            const double nanv = __builtin_nan("");
            const double ox = inside ? a.tx[jc] : nanv, oy = inside ? a.ty[jc] : nanv, oz = inside ? a.tz[jc] : nanv;
            const double w = 1.0 / a.state->sigma2;
            const bool off = (a.lm_mask && a.lm_mask[pc]) || w == 0.0;
            const double dx = ox - cen[0] - tr[0], dy = oy - cen[1] - tr[1], dz = oz - cen[2] - tr[2];
            const double ex = R[0] * dx + R[3] * dy + R[6] * dz - (rx - cen[0]) - mx;
            const double ey = R[1] * dx + R[4] * dy + R[7] * dz - (ry - cen[1]) - my;
            const double ez = R[2] * dx + R[5] * dy + R[8] * dz - (rz - cen[2]) - mz;
            e[0] = off ? 0.0 : w * ex;
            e[1] = off ? 0.0 : w * ey;
            e[2] = off ? 0.0 : w * ez;
            if (valid && lane16 == 0) {
                a.weight_out[p] = off ? 0.0 : w;
                a.evec_out[p] = e[0];
                a.evec_out[M + p] = e[1];
                a.evec_out[2 * M + p] = e[2];
            }
        } else if (MODE == SWEEP_PROJ1) {
            // shape - ref' - mean' = R (Q0_i a); projecting back multiplies by R^T: e = Q0_i a
            e[0] = f0[0];
            e[1] = f0[1];
            e[2] = f0[2];
        } else if (MODE == SWEEP_SHAPES) {
            // newshape = R (ref + mean + Q0_i alpha_c - c) + c + t      (transformedModelInit.instance, :222)
            // cur0     = ref + mean + Q0_i alpha                        (model.instance, :224)
            const double ix = rx + mx + f0[0] - cen[0], iy = ry + my + f0[1] - cen[1], iz = rz + mz + f0[2] - cen[2];
            const double nx = R[0] * ix + R[1] * iy + R[2] * iz + cen[0] + tr[0];
            const double ny = R[3] * ix + R[4] * iy + R[5] * iz + cen[1] + tr[1];
            const double nz = R[6] * ix + R[7] * iy + R[8] * iz + cen[2] + tr[2];
            if (valid && lane16 == 0) {
                a.shape_out[p] = nx;
                a.shape_out[M + p] = ny;
                a.shape_out[2 * M + p] = nz;
                const double x0 = rx + mx + f1[0] - a.c0[0], x1 = ry + my + f1[1] - a.c0[1], x2 = rz + mz + f1[2] - a.c0[2];
                const double y0 = nx - a.c0[0], y1 = ny - a.c0[1], y2 = nz - a.c0[2];
                us[0] += x0; us[1] += x1; us[2] += x2;
                us[3] += y0; us[4] += y1; us[5] += y2;
                us[6] += y0 * x0; us[7] += y0 * x1; us[8] += y0 * x2;
                us[9] += y1 * x0; us[10] += y1 * x1; us[11] += y1 * x2;
                us[12] += y2 * x0; us[13] += y2 * x1; us[14] += y2 * x2;
                us[15] += x0 * x0 + x1 * x1 + x2 * x2;
            }
        } else if (MODE == SWEEP_PROJ2) {
            // newshape - (R2 ref + t2) - R2 mean, rotated back by R2^T      (transformedModel.coefficients, :234-237)
            const double sx = a.shape_in[pc] - cen[0] - tr[0], sy = a.shape_in[M + pc] - cen[1] - tr[1],
                         sz = a.shape_in[2 * M + pc] - cen[2] - tr[2];
            e[0] = R[0] * sx + R[3] * sy + R[6] * sz - (rx - cen[0]) - mx;
            e[1] = R[1] * sx + R[4] * sy + R[7] * sz - (ry - cen[1]) - my;
            e[2] = R[2] * sx + R[5] * sy + R[8] * sz - (rz - cen[2]) - mz;
        } else if (MODE == SWEEP_FIT || MODE == SWEEP_POSED) {
            // fit = s * (R (inst - c) + c + t)       ModelFittingParameters.scala:130-143
            const double ix = rx + mx + f0[0] - cen[0], iy = ry + my + f0[1] - cen[1], iz = rz + mz + f0[2] - cen[2];
            const double nx = R[0] * ix + R[1] * iy + R[2] * iz + cen[0] + tr[0];
            const double ny = R[3] * ix + R[4] * iy + R[5] * iz + cen[1] + tr[1];
            const double nz = R[6] * ix + R[7] * iy + R[8] * iz + cen[2] + tr[2];
            if (valid && lane16 == 0) {
                const double s = (MODE == SWEEP_FIT) ? scale : 1.0;
                a.shape_out[p] = s * nx;
                a.shape_out[M + p] = s * ny;
                a.shape_out[2 * M + p] = s * nz;
            }
        }
        if (TRANS) {
            if (!valid) e[0] = e[1] = e[2] = 0.0;
#pragma unroll
            for (int m = 0; m < KMAX; ++m) {
                if (m < km) {
                    const int k = m * 16;
                    acc[m] = __builtin_fma(q0[k], e[0], __builtin_fma(q1[k], e[1], __builtin_fma(q2[k], e[2], acc[m])));
                }
            }
        }
    }
    if (TRANS) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < KMAX; ++m)
            if (m < km) red[grp * rp + m * 16 + lane16] = acc[m];
        __syncthreads();
        for (int k = tid; k < rp; k += kSweepThreads) {
            double s = 0.0;
            for (int g = 0; g < kGroups; ++g) s += red[g * rp + k];
            a.partial[(int64_t)blockIdx.x * rp + k] = s;
        }
    }
    if (MODE == SWEEP_SHAPES) {
        __syncthreads();
        if (lane16 == 0)
            for (int s = 0; s < 16; ++s) red[grp * 16 + s] = us[s];
        __syncthreads();
        if (tid < 24) {
            double s = 0.0;
            if (tid < 16)
                for (int g = 0; g < kGroups; ++g) s += red[g * 16 + tid];
            a.partial[(int64_t)blockIdx.x * 24 + tid] = s;
        }
    }
}

// out[k] = sum over blocks of partial[b][k]: one workgroup per k, fixed summation tree (bitwise reproducible)
__global__ __launch_bounds__(256) void block_partials_reduce_kernel(const double *__restrict__ partial, int nblocks, int width,
                                                                    double *__restrict__ out) {
    __shared__ double sh[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * width + k];
    sh[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = sh[0];
}

// SWEEP_FIT with quarter boxes (round 4): fit = s (R (ref + mean + Q0 alpha - c) + c + t) as in sweep_kernel<SWEEP_FIT>, but a
// workgroup owns whole 64-point QUARTERS of the fit -- sixteen 16-lane groups x four points each, the basis rows of all four
// requested before the first is used -- and leaves the bounding box of each quarter behind, plus the largest |coordinate - centre|
// of the cloud: what tile_bbox_kernel computed for the column-sum pass of the next iteration in a launch of its own (5 us + a kernel
// boundary per iteration; exact minima / maxima, so the same bits).  KM >= rp / 16.
// S points per 16-lane group in flight (4 for KM <= 8: the whole quarter at once; 1 above: the basis rows of a point are 3 KM values
// per lane, requested eight column blocks at a time -- ranks above 128 keep the quarter-box form with the registers of three waves per
// SIMD instead of falling back to the generic pass + a box launch of its own).
template <int KM, int S>
__global__ __launch_bounds__(kSweepThreads) void sweep_fit_boxes_kernel(SweepArgs a) {
    constexpr int CH = KM < 8 ? KM : 8;  // column blocks per request
    static_assert(4 % S == 0, "points per group and round");
    extern __shared__ double lds[];  // [rp] coefficients, then [16][8] box scratch + 8
    const int tid = threadIdx.x, lane16 = tid & 15, grp = tid >> 4;
    const int rp = a.rp, km = rp >> 4;
    const int64_t M = a.M;
    double *coef = lds, *red = lds + rp;
    for (int k = tid; k < rp; k += kSweepThreads) coef[k] = a.coef0[k];
    double R[9], tr[3], cen[3];
    for (int q = 0; q < 9; ++q) R[q] = a.state->R[q];
    for (int q = 0; q < 3; ++q) {
        tr[q] = a.state->t[q];
        cen[q] = a.state->center[q];
    }
    const double scale = a.state->scale;
    double cf[KM];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < KM; ++m) cf[m] = m < km ? coef[m * 16 + lane16] : 0.0;
    double amax = 0.0;
    for (int64_t qd = blockIdx.x; qd * 64 < M; qd += gridDim.x) {
        double lo[3] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val()};
        double hi[3] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
#pragma unroll
        for (int s0 = 0; s0 < 4; s0 += S) {
            double facc[S][3], rm[S][3];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int d = 0; d < 3; ++d) facc[s][d] = 0.0;
#pragma unroll
            for (int mc = 0; mc < KM; mc += CH) {
                double u[S][3][CH];
#pragma unroll
                for (int s = 0; s < S; ++s) {  // every load of the round's column blocks in flight
                    const int64_t p = qd * 64 + (s0 + s) * kGroups + grp;
                    const int64_t pc = p < M ? p : 0;
                    const double *q0 = a.Q0 + (3 * pc) * rp + lane16;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
#pragma unroll
                        for (int m = 0; m < CH; ++m) u[s][d][m] = (mc + m < KM && mc + m < km) ? q0[d * rp + (mc + m) * 16] : 0.0;
                        if (mc == 0) rm[s][d] = a.ref[d * M + pc] + a.mean[d * M + pc];
                    }
                }
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int m = 0; m < CH; ++m)
                            if (mc + m < KM) facc[s][d] = __builtin_fma(u[s][d][m], cf[mc + m], facc[s][d]);  // (m >= km: 0 * 0)
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int64_t p = qd * 64 + (s0 + s) * kGroups + grp;
                double f[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) f[d] = group16_sum(facc[s][d]);
                // fit = s * (R (inst - c) + c + t)       ModelFittingParameters.scala:130-143
                const double ix = rm[s][0] + f[0] - cen[0], iy = rm[s][1] + f[1] - cen[1], iz = rm[s][2] + f[2] - cen[2];
                const double nx = scale * (R[0] * ix + R[1] * iy + R[2] * iz + cen[0] + tr[0]);
                const double ny = scale * (R[3] * ix + R[4] * iy + R[5] * iz + cen[1] + tr[1]);
                const double nz = scale * (R[6] * ix + R[7] * iy + R[8] * iz + cen[2] + tr[2]);
                if (p < M) {
                    if (lane16 == 0) {
                        a.shape_out[p] = nx;
                        a.shape_out[M + p] = ny;
                        a.shape_out[2 * M + p] = nz;
                    }
                    // (fmin / fmax skip a NaN coordinate: it never widens a box, as in tile_bbox_kernel)
                    lo[0] = fmin(lo[0], nx), lo[1] = fmin(lo[1], ny), lo[2] = fmin(lo[2], nz);
                    hi[0] = fmax(hi[0], nx), hi[1] = fmax(hi[1], ny), hi[2] = fmax(hi[2], nz);
                }
            }
        }
        if (lane16 == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                red[grp * 8 + d] = lo[d];
                red[grp * 8 + 3 + d] = hi[d];
            }
        }
        __syncthreads();
        if (tid < 6) {
            double v = red[tid];
            for (int g = 1; g < kGroups; ++g) v = tid < 3 ? fmin(v, red[g * 8 + tid]) : fmax(v, red[g * 8 + tid]);
            a.qboxes[qd * 6 + tid] = v;
            red[kGroups * 8 + tid] = fabs(v - a.box_centre[tid % 3]);
        }
        __syncthreads();
        if (tid == 0)
            for (int q = 0; q < 6; ++q) amax = fmax(amax, red[kGroups * 8 + q]);
        __syncthreads();  // red is rewritten by the next quarter
    }
    if (tid == 0 && a.absmax_slot) {
        // non-negative doubles order like their bit patterns; only a value above what is already there needs the atomic (the slot was
        // cleared by an EARLIER launch on the stream: post_solve_kernel / state_init_kernel)
        const unsigned long long mb = __builtin_bit_cast(unsigned long long, amax);
        if (mb > __hip_atomic_load(reinterpret_cast<unsigned long long *>(a.absmax_slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned long long *>(a.absmax_slot), mb);
    }
}

// ------------------------------------------------------------------------------------------------- weighted Gram
// A workgroup of 4 waves computes one 64x64 patch (4x4 MFMA tiles) of G over one slab of rows; the waves interleave
// the 4-row steps of the slab and are summed through LDS in a fixed order.
//   D(16x16) += A(16x4) B(4x16),  A[i][k] = w_row * Q0[row0+k][a0+i],  B[k][j] = Q0[row0+k][b0+j]
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]: both are Q0[row0 + (l>>4)][col0 + (l&15)], i.e. four
// 128-byte row segments per load instruction.  D: lane holds col j = l&15, rows i = (l>>4) + 4*reg.
// The next step's fragments are loaded before the current step's 16 MFMAs are issued (software prefetch).
// Generalised for the one-off moment Grams S[d][e] = sum_i Q0[3i+d]^T Q0[3i+e]: logical row L maps to the physical rows
// L*row_stride + offA (A side) and L*row_stride + offB (B side); `full` enumerates all patches instead of pa <= pb.
__global__ __launch_bounds__(256) void gram_kernel(const double *__restrict__ Q0, int64_t rows, int rp,
                                                   const double *__restrict__ weight, int64_t rows_per_slab, int nbp,
                                                   int row_stride, int offA, int offB, int full,
                                                   double *__restrict__ partial) {
    __shared__ double red[16 * 4 * 64];
    int pa = 0, pb = 0;
    if (full) {
        pa = blockIdx.y / nbp;
        pb = blockIdx.y - pa * nbp;
    } else {  // triangular patch index -> (pa <= pb)
        int t = blockIdx.y;
        for (pa = 0; pa < nbp; ++pa) {
            const int cnt = nbp - pa;
            if (t < cnt) {
                pb = pa + t;
                break;
            }
            t -= cnt;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, cl = lane & 15;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab;
    const int64_t r1 = min(rows, r0 + rows_per_slab);
    v4f64 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
    // Column interleave: tile t of the patch holds the columns {p*64 + 4*lane16 + t}, so the four fragment values a lane
    // needs per side are 32 contiguous bytes (two 16-byte loads) instead of four 8-byte loads 128 bytes apart.
    const bool vla = pa * 64 + 4 * cl < rp, vlb = pb * 64 + 4 * cl < rp;  // rp is a multiple of 16: all-or-nothing per lane
    const bool diag = pa == pb && offA == offB;
    typedef double d4 __attribute__((ext_vector_type(4)));
    double ca[4], cb[4];
    auto load = [&](int64_t row, double fa[4], double fb[4]) {
        const int64_t rr = row + kq;
        const bool valid = rr < r1;
        const int64_t rc = valid ? rr : r0;
        const double wv = valid ? (weight ? weight[row_stride == 1 ? rc / 3 : rc] : 1.0) : 0.0;
        d4 vb4 = d4{0, 0, 0, 0};
        if (vlb && valid) vb4 = *reinterpret_cast<const d4 *>(Q0 + (rc * row_stride + offB) * rp + pb * 64 + 4 * cl);
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = vb4[t];
        if (diag) {
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = fb[t] * wv;
        } else {
            d4 va4 = d4{0, 0, 0, 0};
            if (vla && valid) va4 = *reinterpret_cast<const d4 *>(Q0 + (rc * row_stride + offA) * rp + pa * 64 + 4 * cl);
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = va4[t] * wv;
        }
    };
    int64_t row = r0 + 4 * wave;
    if (row < r1) load(row, ca, cb);
    for (; row < r1; row += 16) {
        double na[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
        if (row + 16 < r1) load(row + 16, na, nb);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[i], cb[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ca[t] = na[t];
            cb[t] = nb[t];
        }
    }
    // waves 1..3 are added into wave 0 in order
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) red[((i * 4 + j) * 4 + reg) * 64 + lane] = acc[i][j][reg];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[i][j][reg] += red[((i * 4 + j) * 4 + reg) * 64 + lane];
        }
    }
    if (wave != 0) return;
    double *out = partial + (int64_t)blockIdx.x * rp * rp;
    // D[i_row][j_col]: i_row = kq + 4*reg is the A-side lane index, j_col = cl the B-side lane index
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = pa * 64 + 4 * (kq + 4 * reg) + i;
                const int gj = pb * 64 + 4 * cl + j;
                if (gi < rp && gj < rp) out[(int64_t)gi * rp + gj] = acc[i][j][reg];
            }
        }
}

// Weighted Gram for rp <= 112 (NT = rp / 16 <= 7 column tiles): a PAIR of waves keeps the whole upper triangle of G -- the
// NT (NT + 1) / 2 accumulator tiles are dealt alternately to the two waves (14 + 14 at NT = 7) -- so a 4-row step needs NT fragment
// loads per wave for half of NT (NT + 1) / 2 MFMAs (7 : 14 instead of 8 : 16 for the 64 x 64 patches, and twice per pair), no
// padded tile is ever multiplied and the symmetric half is never computed.  A = w * fragment, B = fragment: one load serves both
// operands.  Eight waves per workgroup = four K groups (they interleave the 4-row steps of the slab) x two tile halves, two waves
// per SIMD: 112 accumulator registers per wave stay in VGPRs.  (One wave holding all 28 tiles needs the AGPR half of the file and
// the compiler then copies all 224 accumulator registers to and from it in every step: 85 us at 50k points instead of 59 us.)
// The K groups are summed through LDS in a fixed order.  Same fragment layout and output layout as gram_kernel.
// compile-time loop: f(std::integral_constant<int, I>) for I = BEGIN .. END-1 (DPP controls must be immediates)
template <int BEGIN, int END, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (BEGIN < END) {
        f(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

// upper-triangle tile q (row-major over t <= u) -> (t, u)
template <int NT>
__host__ __device__ constexpr int tri_row(int q) {
    int t = 0;
    while (q >= NT - t) q -= NT - t, ++t;
    return t;
}
template <int NT>
__host__ __device__ constexpr int tri_col(int q) {
    int t = 0;
    while (q >= NT - t) q -= NT - t, ++t;
    return t + q;
}

template <int NT, int HALF, bool FULL>
__device__ __forceinline__ void gram_tri_half(const double *__restrict__ Q0, int rp, const double *__restrict__ weight, int64_t r0,
                                              int64_t r1, int kgroup, int kq, int cl, int lane, double *red, double *xchg,
                                              double *__restrict__ out, const double *__restrict__ evec, int64_t npts,
                                              double *__restrict__ rhs_out, double *rsh) {
    constexpr int kTiles = NT * (NT + 1) / 2;
    constexpr int kMine = HALF == 0 ? (kTiles + 1) / 2 : kTiles / 2;
    constexpr int kM = kMine > 0 ? kMine : 1;
    v4f64 acc[kM];
#pragma unroll
    for (int q = 0; q < kMine; ++q) acc[q] = v4f64{0, 0, 0, 0};
    // The two waves of a K group need the SAME NT fragments of every 4-row step.  Each loads only every other one (HALF 0: tiles
    // 0, 2, 4, ...; HALF 1: 1, 3, 5, ...) and the pair exchanges them through LDS -- loading all of them in both waves fetched every
    // basis row twice from the fabric (PMC FETCH_SIZE 221 MB for 134 MB of basis at 50k points, rank 100: the second request for
    // a line arrives while the first is still in flight and is not merged).
    //
    // What bounds this loop (tools/ubench_mfma_f64_fill.hip, profiles/r03_ubench_mfma_f64_fill.txt): while a float64 MFMA runs, its
    // SIMD issues NO other vector instruction -- integer, move or float64, from either wave; each one adds its full issue time to
    // the MFMA stream (2.3-5.2 ns), whereas LDS traffic, the barrier and most of a global load's issue are free beside it.  So the
    // time of a step is (28 MFMAs of the SIMD's two waves) + (every VALU instruction of both waves), wherever those are placed,
    // and the loop is built to issue as few as possible:
    //   * the PRODUCER of a fragment scales it (w * fragment, the A operand) and adds it to the right-hand side; both forms go
    //     through LDS, the consumers read 2 NT values and multiply nothing;
    //   * addresses advance incrementally (16 rows per step: pointer += 16 rp, point index += 5 or 6); the from-scratch form (64-bit
    //     multiplies, a division by 3) cost ~45 instructions per step and wave;
    //   * rows past the slab are handled on a wave-uniform slow path (last step of the last slab, prefetches past the end), the
    //     fast path has no selects;
    //   * two operand sets (cur, a) and two prefetch slots alternate, the loop is unrolled by two instead of moving registers.
    // The hand-over of step s+1 is issued between the MFMAs of step s (its latencies hide there); global loads run two steps ahead.
    constexpr int kOwn = HALF == 0 ? (NT + 1) / 2 : NT / 2;  // fragments this wave loads
    constexpr int kO = kOwn > 0 ? kOwn : 1;
    // evec != nullptr: the right-hand side Q0^T evec rides along (each wave for the fragments it loads).  evec: SoA planes [3][npts].
    const bool with_rhs = evec != nullptr;
    const bool with_w = weight != nullptr;
    const int kgu = __builtin_amdgcn_readfirstlane(kgroup);
    const int64_t ubase = r0 + 4 * kgu;       // wave-uniform: row of lane group kq = 0 in step 0; 16 rows further per step
    const int64_t first = ubase + kq;
    const double *pclamp = Q0 + r0 * rp + cl;  // rows past the slab read row r0 (finite); the consumer sets their w and e to 0
    const double *pnext = Q0 + first * rp + cl;
    const int64_t pstep = 16 * (int64_t)rp;
    int64_t left_u = r1 - ubase;  // wave-uniform: rows from the first row of the next step to load to the slab's end
    // row -> (point, coordinate) = (row / 3, row % 3); 16 rows further: (point + 5, coordinate + 1) or (point + 6, coordinate - 2).
    // `third` = coordinate * ceil(2^32 / 3): adding ceil(2^32 / 3) carries exactly when the coordinate wraps (the excess of 2 per
    // wrap stays below the margin for 7e8 wraps).  evec is SoA [3][npts]: entry (coordinate, point).
    const int64_t pt_first = first / 3;
    const int rem_first = (int)(first - 3 * pt_first);
    constexpr unsigned kThird = 0x55555556u;
    unsigned third = (unsigned)rem_first * kThird;
    const double *wclamp = with_w ? weight + r0 / 3 : pclamp;
    const double *eclamp = with_rhs ? evec + r0 / 3 : pclamp;
    const double *wp = with_w ? weight + pt_first : pclamp;  // without weights / evec: any readable address, the value is not used
    const double *ep = with_rhs ? evec + rem_first * npts + pt_first : pclamp;
    const int64_t estep = with_rhs ? 5 + npts : 0, ewrap = with_rhs ? 6 - 2 * npts : 0;
    double fn[2][kO], wn[2], en[2];
    int vrows[2];  // wave-uniform, per prefetch slot: how many of the step's four rows are inside the slab (4 = all)
    // Both paths issue the same loads in the same order, and nothing touches the loaded values here: the waits at the consumer stay
    // counted (vmcnt(n) leaves the younger slot in flight).
    auto load_next = [&](auto slot) __attribute__((always_inline)) {
        constexpr int L = decltype(slot)::value;
        if (left_u >= 4) {  // wave-uniform: all four rows of the step inside the slab
            vrows[L] = 4;
#pragma unroll
            for (int k = 0; k < kOwn; ++k) fn[L][k] = pnext[16 * (2 * k + HALF)];
            wn[L] = *wp;
            en[L] = *ep;
        } else {  // rows past the slab (last step of the last slab, prefetches past the end): clamped addresses
            vrows[L] = (int)max((int64_t)0, left_u);
            const bool valid = left_u > kq;
            const double *p = valid ? pnext : pclamp;
#pragma unroll
            for (int k = 0; k < kOwn; ++k) fn[L][k] = p[16 * (2 * k + HALF)];
            wn[L] = *(valid ? wp : wclamp);
            en[L] = *(valid ? ep : eclamp);
        }
        left_u -= 16;
        pnext += pstep;
        const unsigned t2 = third + kThird;
        const bool wrap = t2 < third;
        third = t2;
        wp += wrap ? 6 : 5;
        ep += wrap ? ewrap : estep;
    };
    double racc[kO];
#pragma unroll
    for (int k = 0; k < kOwn; ++k) racc[k] = 0.0;
    // every wave of the workgroup runs the same number of steps (the barrier inside is workgroup wide): the K group with the
    // most rows sets it; steps past a wave's own rows multiply zeros (w = 0, clamped addresses)
    const int64_t nsteps = (r1 - r0 + 15) / 16;
    constexpr int kBufStride = 4 * 2 * NT * 64;  // xchg: [2 buffers][4 K groups][plain, scaled][NT][64 lanes]
    double *xbuf = xchg + (size_t)kgu * 2 * NT * 64 + lane;
    double cur[2][NT], a[2][NT];
    auto hand_over_write = [&](auto slot) __attribute__((always_inline)) {  // fragments of slot L go to buffer L (step parity = slot = buffer)
        constexpr int L = decltype(slot)::value;
        double w = wn[L], e = en[L];
        if constexpr (!FULL) {  // without weights / evec the loads above read a placeholder
            w = with_w ? w : 1.0;
            e = with_rhs ? e : 0.0;
        }
        if (vrows[L] < 4) {  // wave-uniform
            asm volatile("; rows past the slab");  // (keeps this a scalar branch: as selects it is 5 VALU in every step)
            const bool valid = vrows[L] > kq;
            w = valid ? w : 0.0;
            e = valid ? e : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            xbuf[L * kBufStride + (2 * k + HALF) * 64] = fn[L][k];
            xbuf[L * kBufStride + (NT + 2 * k + HALF) * 64] = fn[L][k] * w;
            racc[k] = __builtin_fma(fn[L][k], e, racc[k]);  // e = 0 without evec
        }
    };
    auto hand_over_read = [&](auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            cur[S][t] = xbuf[S * kBufStride + t * 64];
            a[S][t] = xbuf[S * kBufStride + (NT + t) * 64];
        }
    };
    auto mfmas = [&](auto set, auto begin, auto endq) __attribute__((always_inline)) {  // tiles [begin, end) of this wave's share, operands of `set`
        constexpr int S = decltype(set)::value;
        static_for<decltype(begin)::value, decltype(endq)::value>([&](auto m) {
            constexpr int mi = decltype(m)::value, q = 2 * mi + HALF, tr = tri_row<NT>(q), tc = tri_col<NT>(q);
            acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[S][tr], cur[S][tc], acc[mi], 0, 0, 0);
        });
    };
    using std::integral_constant;
    constexpr integral_constant<int, 0> c0{};
    constexpr integral_constant<int, 1> c1{};
    constexpr int kQ1 = kMine / 4, kQ2 = kMine / 2, kQ3 = (3 * kMine) / 4;
    // prologue: steps 0 and 1 requested, step 0 through LDS into set 0, step 2 requested
    load_next(c0);
    load_next(c1);
    hand_over_write(c0);
    __syncthreads();
    load_next(c0);
    hand_over_read(c0);
    // one step: the MFMAs of set S with the hand-over of the next step (set T = 1 - S) slotted between them
    auto step = [&](auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value, T = 1 - S;
        constexpr integral_constant<int, T> other{};
        __builtin_amdgcn_sched_barrier(0);
        mfmas(set, integral_constant<int, 0>{}, integral_constant<int, kQ1>{});
        __builtin_amdgcn_sched_barrier(0);
        hand_over_write(other);  // own fragments of the next step (requested two steps ago)
        __builtin_amdgcn_sched_barrier(0);
        mfmas(set, integral_constant<int, kQ1>{}, integral_constant<int, kQ2>{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        load_next(other);  // the step after the next two, into the slot just written out
        hand_over_read(other);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(set, integral_constant<int, kQ2>{}, integral_constant<int, kQ3>{});
        __builtin_amdgcn_sched_barrier(0);
        mfmas(set, integral_constant<int, kQ3>{}, integral_constant<int, kMine>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    // (the step after the last one is handed over too and never multiplied: w = 0 rows, one barrier more, no branch in the loop)
    int64_t st = 0;
    for (; st + 1 < nsteps; st += 2) {
        step(c0);
        step(c1);
    }
    if (st < nsteps) step(c0);
    __syncthreads();  // the exchange buffers are free: slot 1 of the K-group reduction below reuses them
    if (evec) {  // workgroup-uniform.  Right-hand side: lanes of a column (the four kq) first, then the K groups 0..3 in order
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            racc[k] += __shfl_xor(racc[k], 16);
            racc[k] += __shfl_xor(racc[k], 32);
        }
        if (kq == 0)
#pragma unroll
            for (int k = 0; k < kOwn; ++k) rsh[kgroup * (NT * 16) + (2 * k + HALF) * 16 + cl] = racc[k];
        __syncthreads();
        if (HALF == 0 && kgroup == 0 && kq == 0)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = t * 16 + cl;
                rhs_out[c] = ((rsh[c] + rsh[NT * 16 + c]) + rsh[2 * NT * 16 + c]) + rsh[3 * NT * 16 + c];
            }
    }
    // K groups: (0 + 2) + (1 + 3), two rounds through LDS (both halves at once, disjoint parts of a slot); slot 1 is the exchange area
    // (1 024 NT doubles >= the 128 NT (NT + 1) (+ 256) of a slot for NT <= 7)
    static_assert(((kTiles + 1) / 2) * 2 * 256 <= 2 * kBufStride, "the exchange area must hold one slot of the K-group reduction");
    double *slot0 = red + HALF * ((kTiles + 1) / 2) * 256, *slot1 = xchg + HALF * ((kTiles + 1) / 2) * 256;
    auto put = [&](double *slot) {
#pragma unroll
        for (int q = 0; q < kMine; ++q)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) slot[(q * 4 + reg) * 64 + lane] = acc[q][reg];
    };
    auto add = [&](const double *slot) {
#pragma unroll
        for (int q = 0; q < kMine; ++q)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[q][reg] += slot[(q * 4 + reg) * 64 + lane];
    };
    if (kgroup >= 2) put(kgroup == 2 ? slot0 : slot1);
    __syncthreads();
    if (kgroup < 2) add(kgroup == 0 ? slot0 : slot1);
    __syncthreads();
    if (kgroup == 1) put(slot0);
    __syncthreads();
    if (kgroup == 0) add(slot0);
    if (kgroup != 0) return;
    // D[i][j] of tile (t, u): i = kq + 4 reg is the A-side index (column 16 t + i of Q0), j = cl the B-side index
    int q = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = t; u < NT; ++u, ++q)
            if ((q & 1) == HALF)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) out[(int64_t)(16 * t + kq + 4 * reg) * rp + 16 * u + cl] = acc[q >> 1][reg];
}

// FULL: weight and evec are both given (the per-iteration call) -- their loads are unconditional
template <int NT, bool FULL>
__global__ __launch_bounds__(512) void gram_tri_kernel(const double *__restrict__ Q0, int64_t rows, int rp,
                                                       const double *__restrict__ weight, int64_t rows_per_slab,
                                                       double *__restrict__ partial, const double *__restrict__ evec, int64_t npts,
                                                       double *__restrict__ rhs_partial, ZeroGate gate) {
    constexpr int kTiles = NT * (NT + 1) / 2;
    if (!gate_open(gate)) return;  // (workgroup-uniform: the downdate launch in front did the work)
    __shared__ double red[(kTiles + 1) * 256];
    __shared__ double xchg[2 * 4 * 2 * NT * 64];
    __shared__ double rsh[4 * NT * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, cl = lane & 15;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab;
    const int64_t r1 = min(rows, r0 + rows_per_slab);
    double *out = partial + (int64_t)blockIdx.x * rp * rp;
    double *rhs_out = rhs_partial ? rhs_partial + (int64_t)blockIdx.x * rp : nullptr;  // one row of right-hand-side partials per slab
    if ((wave >> 2) == 0)
        gram_tri_half<NT, 0, FULL>(Q0, rp, weight, r0, r1, wave & 3, kq, cl, lane, red, xchg, out, evec, npts, rhs_out, rsh);
    else
        gram_tri_half<NT, 1, FULL>(Q0, rp, weight, r0, r1, wave & 3, kq, cl, lane, red, xchg, out, evec, npts, rhs_out, rsh);
}

// G[i][j] = sum over slabs (fixed order): 32 consecutive elements x 8 slab groups per workgroup, so every load instruction
// reads 256-byte runs of one slab; group g adds the slabs g, g+8, ... in ascending order, the groups are combined as
// ((0+1)+(2+3))+((4+5)+(6+7)).  symmetric: only i <= j is read (upper patches) and mirrored; otherwise every element.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double *__restrict__ partial, int nslabs, int rp, int symmetric,
                                                          double *__restrict__ G) {
    __shared__ double sh[8][33];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + el;
    const int rr = rp * rp;
    const int i = idx / rp, j = idx - i * rp;
    const bool need = idx < rr && !(symmetric && i > j);
    double s = 0.0;
    if (need) {
        const double *p = partial + idx;
#pragma unroll 8
        for (int b = g; b < nslabs; b += 8) s += p[(int64_t)b * rr];
    }
    sh[g][el] = s;
    __syncthreads();
    if (g == 0 && need) {
        const double t = ((sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el])) + ((sh[4][el] + sh[5][el]) + (sh[6][el] + sh[7][el]));
        G[i * rp + j] = t;
        if (symmetric) G[j * rp + i] = t;
    }
}

// Everything that turns the partial sums of phase 1 into the shard's exchange segment, in ONE launch (instead of gram_reduce +
// block_partials_reduce + cpd_scalars_finish): workgroups [0, nG) reduce the Gram slab partials exactly like gram_reduce_kernel
// (same grouping, same order), workgroups [nG, nG + rp) the right-hand-side partials of the basis sweep exactly like
// block_partials_reduce_kernel, and the last workgroup the four scalar sums of the CPD passes like cpd_scalars_finish_kernel
// (scalar_mode 1) or just clears the eight scalars (mode 0: ICP).  nslabs == 0: G was produced elsewhere (scaled moment copy).
__global__ __launch_bounds__(256) void phase1_finalize_kernel(Phase1FinalizeArgs A) {
    __shared__ double sh[8][33];
    __shared__ double sv[256];
    const int rp = A.rp, rr = rp * rp;
    const int nG = (A.nslabs > 0 || A.scaled_src) ? (rr + 31) / 32 : 0;
    const int b = blockIdx.x;
    if (A.gate.counts && gate_open(A.gate)) {  // (gate.run_if_many = 1) the weighted pass over the basis ran instead of the downdate
        A.nslabs = A.alt_nslabs;
        A.scaled_src = nullptr;
        A.sweep_blocks = A.alt_nslabs;
    }
    if (b < nG) {
        const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
        const int idx = b * 32 + el;
        if (A.nslabs <= 0) {  // no Gram pass: every row carries the weight 1 / sigma2, G is the model's moment scaled
            if (g == 0 && idx < rr) A.G[idx] = A.scaled_contribute ? A.scaled_src[idx] * (1.0 / A.sigma2[0]) : 0.0;
            return;
        }
        const int i = idx / rp, j = idx - i * rp;
        const bool need = idx < rr && !(i > j);
        double s = 0.0;
        if (need) {
            const double *p = A.gram_partial + idx;
#pragma unroll 8
            for (int q = g; q < A.nslabs; q += 8) s += p[(int64_t)q * rr];
        }
        sh[g][el] = s;
        __syncthreads();
        if (g == 0 && need) {
            double t = ((sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el])) + ((sh[4][el] + sh[5][el]) + (sh[6][el] + sh[7][el]));
            if (A.scaled_src)  // the partials are Q^T Q of the zero-weight rows: the model's moment minus them, every other row at 1 / sigma2
                t = ((A.scaled_contribute ? A.scaled_src[idx] : 0.0) - t) * (1.0 / A.sigma2[0]);
            A.G[i * rp + j] = t;
            A.G[j * rp + i] = t;
        }
        return;
    }
    if (b < nG + rp) {
        const int k = b - nG;
        double s = 0.0;
        for (int q = threadIdx.x; q < A.sweep_blocks; q += 256) s += A.sweep_partial[(int64_t)q * rp + k];
        sv[threadIdx.x] = s;
        __syncthreads();
#pragma unroll
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) sv[threadIdx.x] += sv[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) A.rhs[k] = sv[0];
        return;
    }
    // scalars
    if (A.scalar_mode == 1) {
        const int map[4] = {1, 0, 2, 3};  // part slot -> scalar index (cpd_scalars_finish_kernel)
        for (int q = 0; q < 4; ++q) {
            sv[threadIdx.x] = A.part[q * GINGR_SCALAR_BLOCKS + threadIdx.x];
            __syncthreads();
#pragma unroll
            for (int st = 128; st > 0; st >>= 1) {
                if ((int)threadIdx.x < st) sv[threadIdx.x] += sv[threadIdx.x + st];
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                const double tot = sv[0];
                if (A.scalars_local) A.scalars_local[map[q]] = tot;
                // xPx is a sum over ALL targets, computed on every shard: only one of them may contribute it
                A.sc8[map[q]] = (map[q] == 1 && !A.contribute_xpx) ? 0.0 : tot;
            }
            __syncthreads();
        }
        if (threadIdx.x >= 4 && threadIdx.x < 8) A.sc8[threadIdx.x] = 0.0;
    } else if (threadIdx.x < 8) {
        A.sc8[threadIdx.x] = 0.0;
    }
}

// Q^T Q of the vertices whose weight is exactly 0, slab by slab (launch_gram_downdate).  A workgroup owns the whole rp x rp matrix:
// thread (ti, tj) of a 16 x 16 arrangement keeps the entries (ti + 16 a, tj + 16 b), a, b < 7, in registers.  It walks its slab's
// vertices 256 at a time -- a ballot finds the zero-weight ones -- and adds, for each of them in ascending order, the three rows of the
// basis (staged in LDS four vertices at a time: 14 reads per row and thread) as outer products: fixed order, no atomics.
__global__ __launch_bounds__(256) void gram_downdate_kernel(const double *__restrict__ Q0, int64_t M, int rp, const double *__restrict__ weight,
                                                            int64_t verts_per_slab, double *__restrict__ partial, ZeroGate gate) {
    if (!gate_open(gate)) return;  // (workgroup-uniform: too many zero-weight rows, the pass over the basis behind this launch runs)
    constexpr int kBatch = 4;  // zero-weight vertices staged together: their rows are requested at once (a slab with several of them
                               // would otherwise pay one memory round trip per vertex, and the launch ends with its slowest slab)
    // Ranks above 112 (round 6): the matrix is cut into 112-column patches and blockIdx.y picks one of the upper ones (pa <= pb); a
    // workgroup then keeps the 7 x 7 entries per thread of ITS patch and stages the two column ranges of the rows.  One patch for
    // rp <= 112: the code (and the bits) of round 5.
    int pa = 0, pb = 0;
    {
        const int np = (rp + 111) / 112;
        int t = blockIdx.y;
        for (pa = 0; pa < np; ++pa) {
            if (t < np - pa) {
                pb = pa + t;
                break;
            }
            t -= np - pa;
        }
    }
    const int ca0 = 112 * pa, cb0 = 112 * pb;
    __shared__ double q[kBatch][3][112], qb[kBatch][3][112];
    __shared__ unsigned long long zmask[16];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;  // (tj fastest: the sixteen lanes of a row write 128 contiguous bytes)
    const int64_t v0 = (int64_t)blockIdx.x * verts_per_slab, v1 = v0 + verts_per_slab < M ? v0 + verts_per_slab : M;
    double acc[7][7];
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = 0; b < 7; ++b) acc[a][b] = 0.0;
    constexpr int kRounds = 4;  // 1 024 vertices per pass: their weights are requested together (one memory round trip, not four)
    for (int64_t base = v0; base < v1; base += 256 * kRounds) {
        double wv[kRounds];
#pragma unroll
        for (int r2 = 0; r2 < kRounds; ++r2) {
            const int64_t v = base + 256 * r2 + tid;
            wv[r2] = v < v1 ? weight[v] : 1.0;
        }
        __syncthreads();  // (the previous pass's readers of zmask are done)
#pragma unroll
        for (int r2 = 0; r2 < kRounds; ++r2) {
            const unsigned long long m = __ballot(wv[r2] == 0.0);
            if ((tid & 63) == 0) zmask[4 * r2 + (tid >> 6)] = m;
        }
        __syncthreads();
        // the mask words are walked in scalar registers (workgroup-uniform: a dynamically indexed per-lane copy of the sixteen words
        // would be a chain of selects per access -- it was most of this kernel's time)
        auto word = [&](int k) {
            const unsigned long long v = zmask[k];
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
            return ((unsigned long long)hi << 32) | (unsigned long long)lo;
        };
        int w = 0;
        unsigned long long cur = word(0);
        for (;;) {
            int64_t vz[kBatch];
            int nb = 0;
            while (nb < kBatch) {  // the next (up to) kBatch zero-weight vertices, ascending
                if (cur == 0) {
                    if (++w >= 4 * kRounds) break;
                    cur = word(w);
                    continue;
                }
                vz[nb++] = base + 64 * w + __builtin_ctzll(cur);
                cur &= cur - 1;
            }
            if (nb == 0) break;
            __syncthreads();  // (the previous batch's rows have been used)
            for (int t = tid; t < kBatch * 3 * 112; t += 256) {
                const int s2 = t / (3 * 112), r2 = t - s2 * (3 * 112), d = r2 / 112, k = r2 - 112 * d;
                if (s2 < nb) {
                    const double *row = Q0 + (3 * vz[s2] + d) * (int64_t)rp;
                    q[s2][d][k] = ca0 + k < rp ? row[ca0 + k] : 0.0;
                    if (pb != pa) qb[s2][d][k] = cb0 + k < rp ? row[cb0 + k] : 0.0;
                }
            }
            __syncthreads();
            const double(*qcol)[3][112] = pb != pa ? qb : q;  // (workgroup-uniform)
            for (int s2 = 0; s2 < nb; ++s2) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    double qi[7], qj[7];
#pragma unroll
                    for (int a = 0; a < 7; ++a) qi[a] = q[s2][d][ti + 16 * a], qj[a] = qcol[s2][d][tj + 16 * a];
#pragma unroll
                    for (int a = 0; a < 7; ++a)
#pragma unroll
                        for (int b = 0; b < 7; ++b) acc[a][b] = __builtin_fma(qi[a], qj[b], acc[a][b]);
                }
            }
        }
    }
    double *out = partial + (int64_t)blockIdx.x * rp * rp;
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const int i = ca0 + ti + 16 * a, j = cb0 + tj + 16 * b;
            if (i < rp && j < rp) out[i * rp + j] = acc[a][b];
        }
}

__global__ void centered_mean_kernel(const double *__restrict__ ref, const double *__restrict__ mean, int64_t M, double c0x,
                                     double c0y, double c0z, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    out[i] = ref[i] + mean[i] - c0x;
    out[M + i] = ref[M + i] + mean[M + i] - c0y;
    out[2 * M + i] = ref[2 * M + i] + mean[2 * M + i] - c0z;
}

// ------------------------------------------------------------------------------------------------- observations
__global__ void obs_cpd_kernel(const double *__restrict__ ref, const double *__restrict__ mean, int64_t M,
                               const DevState *__restrict__ st, Cloud fit, const double *__restrict__ P1,
                               const double *__restrict__ PX, double lambda, const int32_t *__restrict__ lm_mask,
                               double *__restrict__ weight, double *__restrict__ evec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (lm_mask && lm_mask[i]) {  // point overridden by a landmark observation (GingrAlgorithm.scala:289-292)
        weight[i] = 0.0;
        evec[i] = evec[M + i] = evec[2 * M + i] = 0.0;
        return;
    }
    const double p1inv = 1.0 / P1[i];                         // CPD.scala:37
    const double yx = fit.x[i], yy = fit.y[i], yz = fit.z[i];
    // td = y + (sum_j P1inv*P_ij*x_j - y)                     CPD.scala:44-46
    const double ox = yx + (PX[i] * p1inv - yx), oy = yy + (PX[M + i] * p1inv - yy), oz = yz + (PX[2 * M + i] * p1inv - yz);
    const double var = st->sigma2 * lambda * p1inv;           // CPD.scala:126
    const double w = 1.0 / var;
    const double *R = st->R;
    const double dx = ox - st->center[0] - st->t[0], dy = oy - st->center[1] - st->t[1], dz = oz - st->center[2] - st->t[2];
    const double ex = R[0] * dx + R[3] * dy + R[6] * dz - (ref[i] - st->center[0]) - mean[i];
    const double ey = R[1] * dx + R[4] * dy + R[7] * dz - (ref[M + i] - st->center[1]) - mean[M + i];
    const double ez = R[2] * dx + R[5] * dy + R[8] * dz - (ref[2 * M + i] - st->center[2]) - mean[2 * M + i];
    weight[i] = w;
    evec[i] = w * ex;
    evec[M + i] = w * ey;
    evec[2 * M + i] = w * ez;
}

// obs point given explicitly (planes ox/oy/oz), weight given or derived from sigma2 (ICP)
__global__ void obs_points_kernel(const double *__restrict__ ref, const double *__restrict__ mean, int64_t M,
                                  const DevState *__restrict__ st, const double *__restrict__ obs, Cloud target,
                                  const int32_t *__restrict__ idx, const double *__restrict__ weight_in,
                                  const int32_t *__restrict__ lm_mask, double *__restrict__ weight,
                                  double *__restrict__ evec, int32_t *__restrict__ zero_counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_counts) {  // the zero-weight vertices of this block (ZeroGate::counts); the same test as below
        const bool zero = i < M && ((lm_mask && lm_mask[i]) || (idx ? 1.0 / st->sigma2 : weight_in[i]) == 0.0);
        const int cnt = __syncthreads_count(zero);
        if (threadIdx.x == 0) zero_counts[blockIdx.x] = cnt;
    }
    if (i >= M) return;
    double w;
    double ox, oy, oz;
    if (idx) {  // ICP: closest target point, cov = I3 * sigma2 (ICP.scala:90-92)
        const int32_t j = idx[i];
        if (j >= 0 && (int64_t)j < target.n) {
            ox = target.x[j];
            oy = target.y[j];
            oz = target.z[j];
        } else {  // the searches leave -1 when no distance is finite: a NaN observation fails the posterior through the status path
            ox = oy = oz = __builtin_nan("");
        }
        w = 1.0 / st->sigma2;
    } else {
        ox = obs[i];
        oy = obs[M + i];
        oz = obs[2 * M + i];
        w = weight_in[i];
    }
    if ((lm_mask && lm_mask[i]) || w == 0.0) {
        weight[i] = 0.0;
        evec[i] = evec[M + i] = evec[2 * M + i] = 0.0;
        return;
    }
    const double *R = st->R;
    const double dx = ox - st->center[0] - st->t[0], dy = oy - st->center[1] - st->t[1], dz = oz - st->center[2] - st->t[2];
    const double ex = R[0] * dx + R[3] * dy + R[6] * dz - (ref[i] - st->center[0]) - mean[i];
    const double ey = R[1] * dx + R[4] * dy + R[7] * dz - (ref[M + i] - st->center[1]) - mean[M + i];
    const double ez = R[2] * dx + R[5] * dy + R[8] * dz - (ref[2 * M + i] - st->center[2]) - mean[2 * M + i];
    weight[i] = w;
    evec[i] = w * ex;
    evec[M + i] = w * ey;
    evec[2 * M + i] = w * ez;
}

// Landmark observations with a full 3x3 covariance: QtL block = Q_p^T Sigma^-1 in the posed frame, i.e.
// W = R^T Sigma^-1 R in the model frame.  Workgroup a < rp owns row a of G, workgroup rp owns rhs; every entry sums
// its landmarks in registers in landmark order and touches G once (the per-landmark read-modify-write of one
// workgroup cost 16 us a landmark).  The 3x3 algebra of a landmark is repeated by one lane of every workgroup.
constexpr int kLmChunk = 256;
__global__ __launch_bounds__(kLmChunk) void landmarks_kernel(const double *__restrict__ Q0, const double *__restrict__ ref,
                                                             const double *__restrict__ mean, int64_t M, int rp,
                                                             const DevState *__restrict__ st, int n_lm,
                                                             const int32_t *__restrict__ pid,
                                                             const double *__restrict__ xyz,
                                                             const double *__restrict__ cov, double *__restrict__ G,
                                                             double *__restrict__ rhs) {
    __shared__ double u[kLmChunk][3];  // row a: sum_d q[d][a] W[d][.]   |   rhs workgroup: W v
    __shared__ int32_t row[kLmChunk];
    const int a = blockIdx.x;
    const bool is_rhs = a == rp;
    constexpr int kCols = 2;  // columns per lane and pass (ranks up to 512 in one pass)
    for (int b0 = 0; b0 < rp; b0 += kCols * kLmChunk) {
        double s[kCols];
        for (int c = 0; c < kCols; ++c) s[c] = 0.0;
        for (int l0 = 0; l0 < n_lm; l0 += kLmChunk) {
            __syncthreads();
            const int l = l0 + (int)threadIdx.x;
            int32_t p = l < n_lm ? pid[l] : -1;
            if (p < 0 || p >= M) p = -1;  // owned by another shard
            double o0 = 0.0, o1 = 0.0, o2 = 0.0;
            if (p >= 0) {
                const double *C = cov + 9 * (int64_t)l;
                const double det = C[0] * (C[4] * C[8] - C[5] * C[7]) - C[1] * (C[3] * C[8] - C[5] * C[6]) +
                                   C[2] * (C[3] * C[7] - C[4] * C[6]);
                double Ci[9];
                Ci[0] = (C[4] * C[8] - C[5] * C[7]) / det;
                Ci[1] = (C[2] * C[7] - C[1] * C[8]) / det;
                Ci[2] = (C[1] * C[5] - C[2] * C[4]) / det;
                Ci[3] = (C[5] * C[6] - C[3] * C[8]) / det;
                Ci[4] = (C[0] * C[8] - C[2] * C[6]) / det;
                Ci[5] = (C[2] * C[3] - C[0] * C[5]) / det;
                Ci[6] = (C[3] * C[7] - C[4] * C[6]) / det;
                Ci[7] = (C[1] * C[6] - C[0] * C[7]) / det;
                Ci[8] = (C[0] * C[4] - C[1] * C[3]) / det;
                const double *R = st->R;
                double T[9], W[9];  // T = Ci * R
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        T[i * 3 + j] = Ci[i * 3] * R[j] + Ci[i * 3 + 1] * R[3 + j] + Ci[i * 3 + 2] * R[6 + j];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) W[i * 3 + j] = R[i] * T[j] + R[3 + i] * T[3 + j] + R[6 + i] * T[6 + j];
                if (is_rhs) {
                    const double dx = xyz[3 * l] - st->center[0] - st->t[0], dy = xyz[3 * l + 1] - st->center[1] - st->t[1],
                                 dz = xyz[3 * l + 2] - st->center[2] - st->t[2];
                    double v[3];
                    v[0] = R[0] * dx + R[3] * dy + R[6] * dz - (ref[p] - st->center[0]) - mean[p];
                    v[1] = R[1] * dx + R[4] * dy + R[7] * dz - (ref[M + p] - st->center[1]) - mean[M + p];
                    v[2] = R[2] * dx + R[5] * dy + R[8] * dz - (ref[2 * M + p] - st->center[2]) - mean[2 * M + p];
                    o0 = W[0] * v[0] + W[1] * v[1] + W[2] * v[2];
                    o1 = W[3] * v[0] + W[4] * v[1] + W[5] * v[2];
                    o2 = W[6] * v[0] + W[7] * v[1] + W[8] * v[2];
                } else {
                    const double *q = Q0 + (int64_t)3 * p * rp + a;
                    const double q0 = q[0], q1 = q[rp], q2 = q[2 * rp];
                    o0 = q0 * W[0] + q1 * W[3] + q2 * W[6];
                    o1 = q0 * W[1] + q1 * W[4] + q2 * W[7];
                    o2 = q0 * W[2] + q1 * W[5] + q2 * W[8];
                }
            }
            u[threadIdx.x][0] = o0;
            u[threadIdx.x][1] = o1;
            u[threadIdx.x][2] = o2;
            row[threadIdx.x] = p;
            __syncthreads();
            const int nl = min(kLmChunk, n_lm - l0);
            for (int k = 0; k < nl; ++k) {
                const int32_t pk = row[k];
                if (pk < 0) continue;
                const double *q = Q0 + (int64_t)3 * pk * rp;
                const double u0 = u[k][0], u1 = u[k][1], u2 = u[k][2];
#pragma unroll
                for (int c = 0; c < kCols; ++c) {
                    const int b = b0 + c * kLmChunk + (int)threadIdx.x;
                    if (b < rp) s[c] += u0 * q[b] + u1 * q[rp + b] + u2 * q[2 * rp + b];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < kCols; ++c) {
            const int b = b0 + c * kLmChunk + (int)threadIdx.x;
            if (b < rp) {
                if (is_rhs)
                    rhs[b] += s[c];
                else
                    G[(int64_t)a * rp + b] += s[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- small dense
// Blocked Cholesky solve in one workgroup (256 threads = one wave per SIMD, 16-wide panels); the bordered matrix is LDS resident
// for r <= 128 and lives in an L2-resident global workspace above that (same code through flat addressing).
// With one wave per SIMD the kernel is bound by the NUMBER of instructions it issues (5-8 cycles each), so everything is
// laid out to need no masks: the matrix is padded with an identity to n = rp (a multiple of 16: every panel is full), right-hand
// sides ride along as 16 extra rows of the bordered matrix (so the forward substitution is a by-product of the panel solves
// and trailing updates), loads are unconditional, and the three stages of a panel use the cross-lane hardware directly:
//   (1) wave 0 factors the 16x16 diagonal block in registers, one row per lane; the rank-1 updates fetch the pivot column
//       through the DPP of the FMA itself (v_fmac_f64_dpp row_newbcast) -- no LDS, no scalar round trip on the chain,
//   (2) the panel below it is solved by 16 lanes per matrix row with the same DPP recurrence, four rows interleaved,
//   (3) the trailing update runs on the matrix pipe, one wave per 16x16 tile (v_mfma_f64_16x16x4).
// The backward substitution is blocked the same way.  3 workgroup barriers per panel.
constexpr int kNB = 16;

// stage clock of tools/ubench_solve.hip (accumulates shader cycles per stage); nothing in the library build
#ifndef GINGR_STAGE_CLOCK
#define GINGR_STAGE_CLOCK(slot)
#endif


// value of lane J of the caller's 16-lane row, in every lane of that row: one v_mov_b64_dpp (gfx90a+ row_newbcast) instead of
// two v_readlane_b32 through the scalar file.  A VGPR written by a VALU instruction may be read through DPP only two wait
// states later and the compiler does not look into inline asm, hence the s_nop inside the statement.
template <int J>
__device__ __forceinline__ double row_bcast(double v) {
    double out;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(out) : "v"(v), "n"(J));
    return out;
}

// acc += (lane J's a) * b, one v_fmac_f64_dpp.  FRESH = true puts the two wait states into the same asm statement (use it
// whenever `a` could have been produced by the preceding instructions).
template <int J, bool FRESH>
__device__ __forceinline__ void fmac_row_bcast(double &acc, double a, double b) {
    if (FRESH)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(a), "v"(b), "n"(J));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
}

// One step of the panel recurrence for four interleaved matrix rows: t_u = x_u * rdl (lane K: the finished x_u[K]), then
// x_u += (lane K's t_u) * nl.  One asm statement: the four products are written four instructions before the DPP reads them, which
// covers the two wait states the DPP needs without any s_nop (the compiler cannot be trusted to keep plain multiplies away from
// an asm that follows them).
template <int K>
__device__ __forceinline__ void panel_step4(double (&x)[4], double rdl, double nl) {
    double t0, t1, t2, t3;
    asm volatile(
        "v_mul_f64 %4, %0, %8\n\t"
        "v_mul_f64 %5, %1, %8\n\t"
        "v_mul_f64 %6, %2, %8\n\t"
        "v_mul_f64 %7, %3, %8\n\t"
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %5, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %6, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %7, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(rdl), "v"(nl), "n"(K));
}

// sum over the 16-lane row of the caller, the same bits in every lane of the row: the butterfly 1, 2, 4, 8 on the DPP crossbar
// (quad permutes, half-row mirror, row mirror) -- __shfl_xor goes through ds_bpermute (~100 cycles per step), which is too long for
// the one-workgroup kernels where it sits on the critical path
__device__ __forceinline__ double row16_sum_dpp(double v) {
    auto step = [&](auto ctrl) {
        constexpr int c = decltype(ctrl)::value;
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, c, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), c, 0xf, 0xf, false);
        v += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    };
    step(std::integral_constant<int, 0xB1>{});   // quad_perm [1, 0, 3, 2]
    step(std::integral_constant<int, 0x4E>{});   // quad_perm [2, 3, 0, 1]
    step(std::integral_constant<int, 0x141>{});  // row_half_mirror
    step(std::integral_constant<int, 0x140>{});  // row_mirror
    return v;
}

// ---- building blocks: all 256 threads call them.  A is (n + xr) x n in LDS, n and xr multiples of 16, odd leading dimension ld.

// doubles of LDS the blocks need for an r x r system with xr extra rows
__host__ __device__ inline int solve_ld(int n) { return n | 1; }  // odd leading dimension: column walks hit distinct banks
__host__ __device__ inline size_t lds_solve_doubles(int rp, int xr) { return (size_t)(rp + xr) * solve_ld(rp) + 2 * (size_t)rp; }

// A (lower triangle of the leading r x r) = ca * G + cs * S + ci * I from global r x rp matrices (S may be nullptr), identity
// on the padding r <= i < n.  16 x 16 element blocks, one element per thread and block, eight blocks in flight; the loads are
// unconditional (clamped indices), only the value is selected.
// HAS_S (round 6): whether the second matrix is there is known at every call site -- as a run-time test of the pointer it sat between
// the loads of the two matrices for each of the 36 blocks, and the request of the next block waited for the previous block's value
// (one memory round trip per block: 19k cycles of the transition-density kernel's 93k, tools/ubench_logpdf_split.hip).
template <int NT, bool HAS_S = false>
__device__ __forceinline__ void lds_load_spd(double *A, int ld, int r, int n, const double *__restrict__ G, double ca,
                                             const double *__restrict__ S, double cs, double ci) {
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    constexpr int RB = NT / 16;  // rows per block of 16 columns: one element per thread and block
    if (RB == 16 && n <= 128) {
        // The LDS-resident case: all (at most 36) lower blocks are requested before the first value is used.  A lone workgroup
        // sees the full memory latency per dependent batch; with batches of eight blocks this stage was four round trips.
        double v[36], vs[HAS_S ? 36 : 1];
        int idx = 0;
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj, ++idx) {
                const int i = bi * 16 + ty, j = bj * 16 + tx;
                const int g = min(min(i, r - 1), n - 1) * n + min(j, r - 1);  // the global matrices have row stride rp == n
                v[idx] = G[g];  // (blocks past n: a clamped, valid address; the value is not stored)
                if (HAS_S) vs[idx] = S[g];
            }
        idx = 0;
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj, ++idx) {
                double t = ca * v[idx];
                if (HAS_S) t = __builtin_fma(cs, vs[idx], t);
                v[idx] = t;
            }
        idx = 0;
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj, ++idx) {
                const int i = bi * 16 + ty, j = bj * 16 + tx;
                if (bi * 16 < n && j <= i) {
                    double t = v[idx];
                    if (i == j) t += ci;
                    A[i * ld + j] = (i < r && j < r) ? t : (i == j ? 1.0 : 0.0);
                }
            }
        __syncthreads();
        return;
    }
    int ib = 0, jb = 0;          // block origin (workgroup-uniform)
    while (ib < n) {
        double v[8];
        int off[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ib + ty, j = jb + tx;
            const int g = min(i, r - 1) * n + min(j, r - 1);  // the global matrices have row stride rp == n
            double t = ca * G[g];
            if (HAS_S) t = __builtin_fma(cs, S[g], t);
            if (i == j) t += ci;
            v[u] = (i < r && j < r) ? t : (i == j ? 1.0 : 0.0);
            off[u] = (ib < n && i < n && j <= i) ? i * ld + j : -1;
            jb += 16;
            if (jb > ib + RB - 16 || jb >= n) {  // past the last column any row of this block needs
                jb = 0;
                ib += RB;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (off[u] >= 0) A[off[u]] = v[u];
    }
    __syncthreads();
}

// in-place blocked Cholesky (lower) of the leading n x n; rd[k] = 1 / L[k][k]; *bad_spd (LDS) is set on a non-positive /
// non-finite pivot.  Rows n .. n+xr-1 hold right-hand sides b^T; they ride along through the panel solves and the trailing
// updates (Cholesky of the bordered matrix), so on return they hold (L^-1 b)^T.
// NT threads (a multiple of 256): the panel solves and the trailing updates spread over NT / 64 waves.  With one wave per SIMD
// every stage is bound by the number of instructions that wave issues (~5 cycles each).
// Round 3: look-ahead.  The diagonal block of step k + 1 only needs block COLUMN k + 1 of the trailing update of step k, and it is
// factored by one wave while the others have nothing to do; so the trailing update is split: first the tiles of column k + 1 (all
// waves), then -- behind one more barrier -- wave 0 factors diagonal block k + 1 WHILE waves 1 .. NW-1 update the remaining tiles.
// Per step max(diagonal block, remaining tiles) replaces their sum: 79k -> 67k cycles at r = 100 (tools/ubench_solve.hip), 38 -> 33 us.
// wr (round 4): that many further rows behind the xr bordered ones take part in the panel solves ONLY (no trailing update).  Set to
// the tiled identity (row c: ones in the columns c, 16 + c, 32 + c, ...) they come back holding W_k = L_kk^-T, the transposed inverse
// of every diagonal block, in the columns of block k -- what lds_backward_w multiplies with instead of running the 16-step recurrence.
// IDENT (round 6; chol_block64_kernel): the xr = n extra rows are the identity (they come back as L^-1).  Row n + c then stays zero in
// every column left of c, so panel kb only has to carry the rows n .. n + kb + 16: the others' panel solves and trailing updates are
// multiplications by zero (62 % of the riding work of four panels instead of all of it).
template <int NT, bool IDENT = false>
__device__ __forceinline__ void lds_cholesky(double *A, int ld, int n, double *rd, int *bad_spd, int xr, int wr = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;           // waves
    constexpr int PU = NT >= 1024 ? 2 : 4;  // matrix rows interleaved per 16-lane group in the panel solve
    constexpr int PR = (NT / 16) * PU;     // matrix rows per panel pass
    const int rows_all = n + xr;
    // (1) diagonal block in registers, ONE wave (all four 16-lane rows do the same work: DPP needs the source lanes active).
    // No masks anywhere: the upper part of the block is loaded, carried and stored as it comes -- lane i's entries right of the
    // diagonal only ever feed lane i's own entries right of the diagonal, and nobody reads the upper part of A (the selects,
    // compares and exec-mask juggling of a masked version were a third of this stage's instructions).  The stage is bound by
    // the NUMBER of instructions the one wave issues, not by the dependent chain: a fraction-free variant (no reciprocal square
    // root on the chain, one more multiply per entry) measured slower, 24.5k against 21.4k cycles for seven blocks.
    auto diag_block = [&](int kb) {
        const int l15 = lane & 15;
        double row[kNB];
#pragma unroll
        for (int k = 0; k < kNB; ++k) row[k] = A[(kb + l15) * ld + kb + k];
        static_for<0, kNB>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            // the pivot, from lane c of this 16-lane row.  A non-positive or non-finite pivot is not tested here (the test
            // would sit on the sequential chain): it turns lc into NaN (rsq(d <= 0) * d, rsq(inf) * inf), the NaN reaches every
            // later diagonal entry of the block, and the check after the loop sees it.
            const double d = row_bcast<c>(row[c]);
            // 1/sqrt(d) by v_rsq_f64 + two Newton steps; lane c's own element d * rsqrt(d) is sqrt(d): the IEEE sqrt and
            // divide sequences are ~40 dependent instructions and would sit on the sequential chain of every column
            double rdk = __builtin_amdgcn_rsq(d);
            const double hd = 0.5 * d;
            rdk = rdk * __builtin_fma(-hd * rdk, rdk, 1.5);
            rdk = rdk * __builtin_fma(-hd * rdk, rdk, 1.5);
            const double lc = row[c] * rdk;
            const double nlc = -lc;
            row[c] = lc;
            rd[kb + c] = rdk;  // 1 / L[c][c]; the same value from every lane
            // row[j] -= L[lane][c] * L[j][c]: L[j][c] is lane j's lc, fetched by the DPP of the FMA itself
            static_for<c + 1, kNB>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                fmac_row_bcast<j, j == c + 1>(row[j], lc, nlc);
            });
        });
        if (lane < kNB) {
#pragma unroll
            for (int k = 0; k < kNB; ++k) A[(kb + lane) * ld + kb + k] = row[k];
            // L[lane][lane] = row[lane]: a register array cannot be indexed by the lane; read it back
            const double diag = A[(kb + lane) * ld + kb + lane];
            if (!(diag > 0.0) || !finite_d(diag)) *bad_spd = 1;
        }
    };
    // (3) one 16x16 tile of the trailing update on the matrix pipe: D = C - L_I L_J^T as four v_mfma_f64_16x16x4 (k = 16).
    // Fragment layout as in gram_kernel: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; it holds
    // D[i = (l >> 4) + 4 reg][j = l & 15].  All tiles are full; the upper half of a diagonal tile is updated too (nobody reads it).
    // (measured in round 2: bound by the LDS traffic of the fragments -- C in, A, B, C out = 8 KB per tile -- not by latency)
    auto tile_update = [&](int kb, int ti, int tj) {
        const int t0 = kb + kNB;
        const int l15 = lane & 15, l4 = lane >> 4;
        const int i0 = t0 + 16 * ti, j0 = t0 + 16 * tj;
        const double *pa = A + (i0 + l15) * ld + kb + l4, *pb = A + (j0 + l15) * ld + kb + l4;
        double *pc = A + (i0 + l4) * ld + j0 + l15;
        v4f64 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = pc[4 * g * ld];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * q], pb[4 * q], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) pc[4 * g * ld] = acc[g];
    };
    if (wave == 0) diag_block(0);
    __syncthreads();
    for (int kb = 0; kb < n; kb += kNB) {
        const int rows = IDENT ? n + min(xr, kb + kNB) : rows_all, prows = rows + wr;
        GINGR_STAGE_CLOCK(1)
        // (2) panel below the diagonal block: x L11^T = A[i][kb:kb+16].  Sixteen lanes per matrix row: lane c keeps x[c] and
        // row c of L11 in registers; at step k every lane with c > k takes x[k] / L[k][k] from lane k through the DPP of its
        // FMA.  No LDS traffic inside the recurrence; four matrix rows per 16-lane group are interleaved to fill the chain.
        {
            const int grp = tid >> 4, c16 = tid & 15;
            double nL[kNB];  // -L11[c16][k] for k < c16, else 0 (lanes c <= k must not move)
#pragma unroll
            for (int k = 0; k < kNB; ++k) {
                const double v = A[(kb + c16) * ld + kb + k];
                nL[k] = k < c16 ? -v : 0.0;
            }
            const double rdl = rd[kb + c16];
            for (int ib = kb + kNB; ib < prows; ib += PR) {  // workgroup-uniform trip count
                const int i0 = ib + grp;
                double x[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) x[u] = A[min(i0 + (NT / 16) * u, prows - 1) * ld + kb + c16];
                static_for<0, kNB>([&](auto kk) {
                    constexpr int k = decltype(kk)::value;
                    if constexpr (PU == 4) {
                        panel_step4<k>(x, rdl, nL[k]);
                    } else {
                        double t[PU];
#pragma unroll
                        for (int u = 0; u < PU; ++u) t[u] = x[u] * rdl;  // lane k: the finished x[k]
#pragma unroll
                        for (int u = 0; u < PU; ++u) fmac_row_bcast<k, true>(x[u], t[u], nL[k]);
                    }
                });
#pragma unroll
                for (int u = 0; u < PU; ++u)
                    if (i0 + (NT / 16) * u < prows) A[(i0 + (NT / 16) * u) * ld + kb + c16] = x[u] * rdl;
            }
        }
        __syncthreads();
        GINGR_STAGE_CLOCK(2)
        const int t0 = kb + kNB;
        const int nti = (rows - t0) >> 4, ntj = (n - t0) >> 4;
        // (3a) block column kb + 16 of the trailing matrix (tj = 0): what the next diagonal block and the next panel read
        if (ntj > 0)
            for (int ti = wave; ti < nti; ti += NW) tile_update(kb, ti, 0);
        __syncthreads();
        // (3b) wave 0 factors the next diagonal block while the other waves update the remaining tiles (tj >= 1).  One wave only
        // (NW == 1): everything in sequence.
        if (wave == 0 && ntj > 0) diag_block(t0);
        if (NW == 1 || wave > 0) {
            constexpr int NR = NW > 1 ? NW - 1 : 1;
            const int me = NW > 1 ? wave - 1 : 0;
            int tcount = 0;
            for (int ti = 1; ti < nti; ++ti)
                for (int tj = 1; tj <= ti && tj < ntj; ++tj, ++tcount)
                    if (tcount % NR == me) tile_update(kb, ti, tj);  // wave-uniform
        }
        __syncthreads();
        GINGR_STAGE_CLOCK(3)
    }
}

// y <- L^-T y for the n entries of y (blocked, bottom up; the 16x16 triangular solves run in registers of wave 0 with the DPP
// recurrence: lane c holds column c of the diagonal block)
template <int NT>
__device__ __forceinline__ void lds_backward(const double *A, int ld, int n, const double *rd, double *y) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int kb = n - kNB; kb >= 0; kb -= kNB) {
        if (wave == 0) {
            const int l15 = lane & 15;
            double ncol[kNB];  // -L11[k][lane] for k > lane, else 0 (lanes >= k must not move at step k)
#pragma unroll
            for (int k = 0; k < kNB; ++k) {
                const double v = A[(kb + k) * ld + kb + l15];
                ncol[k] = k > l15 ? -v : 0.0;
            }
            double yv = y[kb + l15];
            const double rdl = rd[kb + l15];
            static_for<0, kNB>([&](auto cc) {
                constexpr int c = kNB - 1 - decltype(cc)::value;
                const double t = yv * rdl;  // lane c: x_c (its yv is complete)
                fmac_row_bcast<c, true>(yv, t, ncol[c]);
            });
            if (lane < kNB) y[kb + lane] = yv * rdl;
        }
        __syncthreads();
        for (int i = tid; i < kb; i += NT) {
            double sacc = y[i];
#pragma unroll
            for (int k = 0; k < kNB; ++k) sacc = __builtin_fma(-A[(kb + k) * ld + i], y[kb + k], sacc);
            y[i] = sacc;
        }
        __syncthreads();
    }
}

// x = L^-T y with the transposed inverses of the diagonal blocks at hand (lds_cholesky, wr = 16: W points at the first identity row,
// W[c * ld + kb + j] = (L_kk^-T)[c][j], exact zeros left of the diagonal).  Per block a 16 x 16 mat-vec (one product per thread, DPP
// row sum) replaces the 16-step sequential recurrence of lds_backward, and x goes to its own array so that one barrier per stage is
// enough: 12k -> see tools/ubench_solve.hip (cycles of seven blocks at r = 100).  NT == 256; y is destroyed.
template <int NT>
__device__ __forceinline__ void lds_backward_w(const double *A, int ld, int n, const double *W, double *y, double *x) {
    static_assert(NT == 256, "one product of the 16 x 16 block per thread");
    const int tid = threadIdx.x, c = tid >> 4, j = tid & 15;
    for (int kb = n - kNB; kb >= 0; kb -= kNB) {
        const double p = row16_sum_dpp(W[c * ld + kb + j] * y[kb + j]);
        if (j == 0) x[kb + c] = p;
        __syncthreads();
        if (tid < kb) {  // y[i] -= sum_k L[kb + k][i] x[kb + k]: two interleaved chains of eight
            double s0 = y[tid], s1 = 0.0;
#pragma unroll
            for (int k = 0; k < kNB; k += 2) {
                s0 = __builtin_fma(-A[(kb + k) * ld + tid], x[kb + k], s0);
                s1 = __builtin_fma(-A[(kb + k + 1) * ld + tid], x[kb + k + 1], s1);
            }
            y[tid] = s0 + s1;
        }
        __syncthreads();
    }
}

// y <- L^-1 y for the n entries of y (blocked, top down): the mirror image of lds_backward -- lane c of wave 0 holds ROW c of the
// diagonal block, at step c every lane below takes x_c from lane c through the DPP of its FMA.
template <int NT>
__device__ __forceinline__ void lds_forward(const double *A, int ld, int n, const double *rd, double *y) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int kb = 0; kb < n; kb += kNB) {
        if (wave == 0) {
            const int l15 = lane & 15;
            double nrow[kNB];  // -L11[lane][k] for k < lane, else 0 (lanes <= k must not move at step k)
#pragma unroll
            for (int k = 0; k < kNB; ++k) {
                const double v = A[(kb + l15) * ld + kb + k];
                nrow[k] = k < l15 ? -v : 0.0;
            }
            double yv = y[kb + l15];
            const double rdl = rd[kb + l15];
            static_for<0, kNB>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const double t = yv * rdl;  // lane c: x_c (its yv is complete)
                fmac_row_bcast<c, true>(yv, t, nrow[c]);
            });
            if (lane < kNB) y[kb + lane] = yv * rdl;
        }
        __syncthreads();
        for (int i = kb + kNB + tid; i < n; i += NT) {
            double sacc = y[i];
#pragma unroll
            for (int k = 0; k < kNB; ++k) sacc = __builtin_fma(-A[i * ld + kb + k], y[kb + k], sacc);
            y[i] = sacc;
        }
        __syncthreads();
    }
}

// nrows x ncols (ncols <= 128) doubles from a row-major global matrix into the LDS matrix A, sixteen loads per thread requested
// before the first store (round 6: as a plain load-store loop each element waits for its own memory round trip -- the lone
// workgroup of these kernels has nothing else to hide it behind; chol_block64_kernel's copy went from 8.6k to 2.8k cycles this way)
template <int NT>
__device__ __forceinline__ void lds_fill_rows(double *A, int ld, const double *__restrict__ src, int64_t ld_src, int nrows, int ncols) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    for (int i0 = w; i0 < nrows; i0 += NW * 8) {  // (workgroup-uniform trip count)
        double v[8][2];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = i0 + NW * q, j = l + 64 * h;
                v[q][h] = (i < nrows && j < ncols) ? src[(int64_t)i * ld_src + j] : 0.0;
            }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = i0 + NW * q, j = l + 64 * h;
                if (i < nrows && j < ncols) A[i * ld + j] = v[q][h];
            }
    }
}

// Diagonal block of a blocked Cholesky that runs over many workgroups (classic_cpd.hip): factor the 64 x 64 block k of the
// row-major matrix Aw in LDS with the building blocks above and invert the factor on the way -- the identity rides along as 64
// extra rows, which come back as (L^-1 e_c)^T = row c of L^-T.  Aw block <- L (upper part zeroed), Linv[k] <- L^-1 (64 x 64, dense).
__global__ __launch_bounds__(256) void chol_block64_kernel(double *__restrict__ Aw, int64_t ld, int k, double *__restrict__ Linv,
                                                           int32_t *__restrict__ flag) {
    extern __shared__ double lds_sm[];
    constexpr int n = 64, lda = 65;  // solve_ld(64)
    double *A = lds_sm, *rd = A + 2 * n * lda;
    __shared__ int bad;
    const int tid = threadIdx.x;
    double *blk = Aw + ((int64_t)k * n) * ld + (int64_t)k * n;
#ifdef GINGR_CHOL64_STAMPS
    GINGR_STAGE_CLOCK(7)
#endif
    {   // the sixteen entries of a thread requested together (one memory round trip; as a load-store loop this stage was 8.6k of
        // the kernel's 40k cycles: every iteration waited for its own load)
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q;
            v[q] = blk[(int64_t)(e >> 6) * ld + (e & 63)];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q, r = e >> 6, c = e & 63;
            A[r * lda + c] = v[q];
            A[(n + r) * lda + c] = r == c ? 1.0 : 0.0;
        }
    }
    if (tid == 0) bad = 0;
    __syncthreads();
#ifdef GINGR_CHOL64_STAMPS  // tools/ubench_chol_block64.hip only
    GINGR_STAGE_CLOCK(0)
#endif
    lds_cholesky<256, true>(A, lda, n, rd, &bad, n);
#ifdef GINGR_CHOL64_STAMPS
    GINGR_STAGE_CLOCK(1)
#endif
    double *li = Linv + (int64_t)k * n * n;
    for (int e = tid; e < n * n; e += 256) {
        const int r = e >> 6, c = e & 63;
        blk[(int64_t)r * ld + c] = c <= r ? A[r * lda + c] : 0.0;
        li[e] = c <= r ? A[(n + c) * lda + r] : 0.0;
    }
    if (tid == 0 && bad) *flag = GINGR_ERR_NOT_SPD;
#ifdef GINGR_CHOL64_STAMPS
    GINGR_STAGE_CLOCK(4)
    GINGR_STAGE_CLOCK(6)
#endif
}

// a = (I + G)^-1 rhs; with zrand != nullptr a posterior SAMPLE of the coefficients: a + L^-T z, z ~ N(0, I)
// (Cov = L^-T L^-1 = (I + G)^-1, the posterior covariance of the coefficients: the same distribution as
//  posterior.sample() of scalismo's SVD-parameterised posterior model, G/api/GingrAlgorithm.scala:211).
// gwork == nullptr: the bordered matrix lives in LDS (r <= 128).  Otherwise it lives in the global workspace `gwork`
// (lds_solve_doubles(rp, 16) doubles, L2 resident): the same building blocks through flat addressing, for 128 < r <= 512 --
// one workgroup, so the workgroup barriers order its global writes.
// Threads of the one-workgroup solve kernels.  Measured (tools/ubench_solve.hip, r = 100): 1024 threads shorten the load stage
// (9.9k -> 6.8k cycles) but lengthen the trailing update (25k -> 37k: every wave walks the whole tile list) and leave the panel
// solve where it is (its per-thread set-up is replicated in four times the waves): 46 us against 42 us at 256 threads.
constexpr int kSolveThreads = 256;

// GW = false: the workspace is the dynamic LDS block and nothing else -- the compiler then proves every access of the building
// blocks to be address space 3 and emits ds_read / ds_write.  (With one kernel choosing between LDS and a global pointer at run
// time every access was a FLAT instruction: ~3x the latency of the LDS path, 64-bit address arithmetic, SGPR spills.)
// The solve on a workspace `sm` (LDS or global, see below); all kSolveThreads threads.  bad_spd / bad: two ints of LDS.
// FAST (LDS-resident, rp <= 112: sixteen more rows fit into the 160 KB): the identity rows of lds_cholesky / lds_backward_w.
template <bool FAST, typename PtrD>
__device__ __forceinline__ void posterior_solve_body(PtrD sm, int r, int rp, const double *__restrict__ G, const double *__restrict__ rhs,
                                                     const double *__restrict__ zrand, double *__restrict__ a, DevState *__restrict__ st,
                                                     int *bad_spd, int *bad) {
    const int n = rp, ld = solve_ld(n);
    constexpr int XR = FAST ? 2 * kNB : kNB;
    auto A = sm;
    auto y = sm + (size_t)n * ld;   // row n of the bordered matrix: the right-hand side (rows n+1 .. n+15 are zero)
    auto rd = sm + (size_t)(n + XR) * ld;  // reciprocal diagonal of L
    auto y2 = rd + n;                       // sampling direction
    const int tid = threadIdx.x;
    if (tid == 0) {
        *bad_spd = 0;
        *bad = 0;
    }
    for (int k = tid; k < kNB * ld; k += kSolveThreads) y[k] = k < r ? rhs[k] : 0.0;
    for (int k = tid; k < n; k += kSolveThreads) y2[k] = (zrand && k < r) ? zrand[k] : 0.0;
    if (FAST)  // rows n+16 .. n+31: the tiled identity
        for (int k = tid; k < kNB * ld; k += kSolveThreads) {
            const int c = k / ld, j = k - c * ld;
            y[kNB * ld + k] = (j < n && (j & 15) == c) ? 1.0 : 0.0;
        }
    // Mm = QtL Q + I     (scalismo genericRegressionComputations)
    GINGR_STAGE_CLOCK(7)
    lds_load_spd<kSolveThreads>(A, ld, r, n, G, 1.0, nullptr, 0.0, 1.0);
    GINGR_STAGE_CLOCK(0)
    lds_cholesky<kSolveThreads>(A, ld, n, rd, bad_spd, kNB, FAST ? kNB : 0);  // y <- L^-1 y on the way
    GINGR_STAGE_CLOCK(4)
    if constexpr (FAST) {
        // x = L^-T y into rd (the reciprocal diagonal is not needed any more); the sampling direction L^-T z the same way
        if (zrand) {  // (before y: the second call reuses rd for its result, so the first result moves to y2's place afterwards)
            lds_backward_w<kSolveThreads>(A, ld, n, y + kNB * ld, y2, rd);
            for (int k = tid; k < n; k += kSolveThreads) y2[k] = rd[k];
            __syncthreads();
        }
        lds_backward_w<kSolveThreads>(A, ld, n, y + kNB * ld, y, rd);
        for (int k = tid; k < n; k += kSolveThreads) y[k] = rd[k];
        __syncthreads();
    } else {
        lds_backward<kSolveThreads>(A, ld, n, rd, y);
        if (zrand) lds_backward<kSolveThreads>(A, ld, n, rd, y2);
    }
    GINGR_STAGE_CLOCK(5)
    GINGR_STAGE_CLOCK(6)
    for (int k = tid; k < rp; k += kSolveThreads) {
        const double v = k < r ? y[k] + y2[k] : 0.0;
        a[k] = v;
        if (!finite_d(v)) *bad = 1;
    }
    __syncthreads();
    if (tid == 0) {
        if (*bad_spd)
            st->err = GINGR_ERR_NOT_SPD;
        else if (*bad)
            st->err = GINGR_ERR_NONFINITE;
    }
}

// MODE 0: LDS, identity rows (rp <= 112); 1: LDS (rp = 128); 2: global workspace (r > 128)
template <int MODE>
__global__ __launch_bounds__(kSolveThreads) void posterior_solve_lds_kernel(int r, int rp, const double *__restrict__ G,
                                                                  const double *__restrict__ rhs,
                                                                  const double *__restrict__ zrand, double *__restrict__ a,
                                                                  DevState *__restrict__ st, double *gwork) {
    extern __shared__ double lds_sm[];
    __shared__ int bad_spd, bad;
#ifdef GINGR_SOLVE_TWICE  // tools/ubench_solve.hip only: the second pass runs with the kernel's code in the instruction cache
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#endif
    if constexpr (MODE == 2)
        posterior_solve_body<false>(gwork, r, rp, G, rhs, zrand, a, st, &bad_spd, &bad);
    else
        posterior_solve_body<MODE == 0>(lds_sm, r, rp, G, rhs, zrand, a, st, &bad_spd, &bad);
#ifdef GINGR_SOLVE_TWICE
    }
#endif
}

// ---- ranks above 128: the bordered matrix (n + 16 rows) does not fit the LDS.  One workgroup of eight waves factors it by SUPER-PANELS
// of SW columns, left-looking: for panel k (columns kb .. kb + SW) every 16 x 16 tile of the rows kb .. n + 16 is formed as
//     (I + G)[tile] - L[rows, 0 : kb] L[panel rows, 0 : kb]^T
// on the matrix pipe straight from the factor already written to the global workspace (L2 resident; fragments read as in
// lds_cholesky's tile_update) and lands in LDS, where lds_cholesky factors the SW x SW head and lets the rows below ride along -- the
// same building blocks, the same bordered form (row n carries the right-hand side through the forward substitution).  The panel then
// goes back to the workspace.  The backward substitution walks the panels bottom-up: a mat-vec with the rows below from the
// workspace, lds_backward on the diagonal block.  Until round 5 the LDS kernel ran on the global workspace through flat addressing,
// one 16-column panel at a time: 265 us at r = 256, 1.46 ms at r = 512 (profiles/r06_base_rank256_512_kernel_stats.txt).
constexpr int kWideSolveThreads = 512;

template <int SW>
struct WideSolveLds {
    double rdl[SW];
    double red[2][kWideSolveThreads / 64][SW];
    double xs[2][512];  // x = L^-T y and, when sampling, L^-T z (n <= 512)
    double yv[2][SW];
    int bad_spd;
};

// The body: factor C = ca G + cs S + ci I (identity on the padding r <= i < n) in the workspace gw, with the bordered row
// b1 + cb b2 riding through the forward substitution, then substitute back.  On return Ls.xs[0] = C^-1 (b1 + cb b2), Ls.xs[1] =
// L^-T zrand (zeros without zrand), the factor and its reciprocal diagonal are in gw ([(n + 16)][n], then [n]), Ls.bad_spd is set on
// a non-positive / non-finite pivot.  All kWideSolveThreads threads; P = the dynamic LDS block [(n + 16)][SW + 1].
template <int SW, bool HAS_S>
__device__ __forceinline__ void wide_solve_body(double *P, WideSolveLds<SW> &Ls, int r, int n, const double *__restrict__ G, double ca,
                                                const double *__restrict__ S, double cs, double ci, const double *__restrict__ b1,
                                                const double *__restrict__ b2, double cb, const double *__restrict__ zrand, double *gw) {
    constexpr int ldp = SW + 1, NW = kWideSolveThreads / 64;
    double *rdl = Ls.rdl;
    auto &red = Ls.red;
    auto &xs = Ls.xs;
    auto &yv = Ls.yv;
    int &bad_spd = Ls.bad_spd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double *Lg = gw;                          // [(n + 16)][n]: the factor, then the bordered rows
    double *rdg = gw + (size_t)(n + kNB) * n;  // [n] reciprocal diagonal
    const int nvec = zrand ? 2 : 1;
    if (tid == 0) bad_spd = 0;
    GINGR_STAGE_CLOCK(7)
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int NTJ = SW / 16;                                // tile columns of a panel
    constexpr int TSTEP = NW / NTJ;                             // wave w owns tile column w % NTJ, tile rows w / NTJ + TSTEP q
    constexpr int TACC = SW == 64 ? 9 : 9;                      // tiles per wave, at most: ceil((n + 16) / 16 / TSTEP), n <= 256 (512)
    constexpr int HALF = SW / 2;                                // 16-byte units per staged row
    constexpr int RSTEP = kWideSolveThreads / HALF;             // rows a staging round covers
    constexpr int PRE = 17;                                     // staging rounds, at most: ceil((n + 16) / RSTEP), n <= 256 (512)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int my_tj = wv % NTJ, ti0 = wv / NTJ;
    // the bordered rows of the workspace: the right-hand side, fifteen zero rows (they ride through every panel)
    for (int e = tid; e < kNB * n; e += kWideSolveThreads) Lg[(size_t)n * n + e] = e < r ? (b2 ? __builtin_fma(cb, b2[e], b1[e]) : b1[e]) : 0.0;
    __syncthreads();
    const int srow = tid / HALF, scp = tid % HALF;              // this thread's row (per round) and column pair of a staged slice
    for (int kb = 0; kb < n; kb += SW) {
        const int sw = min(SW, n - kb), rows = n - kb + kNB;
        const int nti = rows >> 4, ntj = sw >> 4;
        // The panel's tiles, C - L[rows, 0 : kb] L[panel rows, 0 : kb]^T, accumulate in registers while the panel's LDS block stages
        // the operands: the factor's columns go through it SW at a time (coalesced 16-byte loads, the next slice requested before the
        // MFMAs of the current one), and C = I + G (bordered rows: the right-hand side) goes through it last and stays, minus the
        // accumulated products.
        // (Fragments fetched straight from the workspace, 8 bytes per lane and sixteen rows per instruction, made this stage 62 % of
        // the kernel: 354k cycles at r = 256.)
        v4f64 acc[TACC];
#pragma unroll
        for (int q = 0; q < TACC; ++q) acc[q] = v4f64{0, 0, 0, 0};
        const bool col_live = my_tj < ntj;
        const int nsl = kb / SW;  // slices of the factor; slice nsl is C
        d2 pre[PRE];
        auto fetch_slice = [&](int sl) __attribute__((always_inline)) {
            // slice nsl is C itself: the same rows and row stride, read from G; the bordered rows always come from the workspace
            const bool is_c = sl == nsl;
            const int col = sl * SW + 2 * scp;  // (== kb + 2 scp for C)
            const double *base = (is_c ? G : Lg) + col;
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                const int gi = kb + srow + u * RSTEP;
                if (gi < n + kNB) pre[u] = *reinterpret_cast<const d2 *>((gi < n ? base : Lg + col) + (size_t)gi * n);
            }
            if (is_c) {  // workgroup-uniform.  Mm = QtL Q + I, identity on the padding   (scalismo genericRegressionComputations)
                d2 ps[HAS_S ? PRE : 1];
                if (HAS_S) {  // the second matrix of the transition density's system, same rows: one more batch of loads per panel
#pragma unroll
                    for (int u = 0; u < PRE; ++u) {
                        const int gi = kb + srow + u * RSTEP;
                        if (gi < n) ps[u] = *reinterpret_cast<const d2 *>(S + col + (size_t)gi * n);
                    }
                }
#pragma unroll
                for (int u = 0; u < PRE; ++u) {
                    const int gi = kb + srow + u * RSTEP;
                    if (gi < n) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const double one = gi == col + h ? 1.0 : 0.0;
                            double t = ca * pre[u][h];
                            if (HAS_S) t = __builtin_fma(cs, ps[u][h], t);
                            pre[u][h] = (gi < r && col + h < r) ? t + ci * one : one;
                        }
                    }
                }
            }
        };
        fetch_slice(0);
        for (int sl = 0; sl <= nsl; ++sl) {
            __syncthreads();  // (the previous slice, or the previous panel's write-back, has been read)
#pragma unroll
            for (int u = 0; u < PRE; ++u)
                if (srow + u * RSTEP < rows) {
                    P[(srow + u * RSTEP) * ldp + 2 * scp] = pre[u][0];
                    P[(srow + u * RSTEP) * ldp + 2 * scp + 1] = pre[u][1];
                }
            __syncthreads();
            if (sl == nsl) break;
            fetch_slice(sl + 1);
            if (col_live) {  // wave-uniform
                const double *pb = P + (16 * my_tj + l15) * ldp + l4;
#pragma unroll
                for (int q = 0; q < TACC; ++q) {
                    const int ti = ti0 + TSTEP * q;
                    if (ti < nti && ti >= my_tj) {  // wave-uniform; strictly above the diagonal: nobody reads it
                        const double *pa = P + (16 * ti + l15) * ldp + l4;
#pragma unroll 8
                        for (int u = 0; u < SW / 4; ++u) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * u], pb[4 * u], acc[q], 0, 0, 0);
                    }
                }
            }
        }
        if (kb > 0 && col_live) {
#pragma unroll
            for (int q = 0; q < TACC; ++q) {
                const int ti = ti0 + TSTEP * q;
                if (ti < nti && ti >= my_tj) {
                    double *pc = P + (16 * ti + l4) * ldp + 16 * my_tj + l15;  // D[i = l4 + 4 g][j = l15]
#pragma unroll
                    for (int g = 0; g < 4; ++g) pc[4 * g * ldp] -= acc[q][g];
                }
            }
        }
        __syncthreads();
        GINGR_STAGE_CLOCK(0)
        lds_cholesky<kWideSolveThreads>(P, ldp, sw, rdl, &bad_spd, rows - sw);
        {   // the panel goes back to the workspace
            const int j = tid % SW;
            if (j < sw)
                for (int i = tid / SW; i < rows; i += kWideSolveThreads / SW) Lg[(size_t)(kb + i) * n + kb + j] = P[i * ldp + j];
        }
        if (tid < sw) rdg[kb + tid] = rdl[tid];
        GINGR_STAGE_CLOCK(4)
        // The next panel's first slice is requested straight away, and at kb = SW that slice IS the panel just written, by other
        // threads: without this barrier a rare race (one posterior in ~1 500 at rank 150 when other kernels share the device; found
        // through the three-shard tests, tools/experiments/stress_group2.py / stress_group3.py).
        __syncthreads();
    }
    // x = L^-T y: y = row n of the workspace (L^-1 rhs); the sampling direction L^-T z rides along
    const int kb_last = ((n - 1) / SW) * SW;
    for (int kb = kb_last; kb >= 0; kb -= SW) {
        const int sw = min(SW, n - kb);
        {   // t_c = y_c - sum over the rows j below the block of L[j][kb + c] x_j: NW row groups, combined in order
            const int c = tid % SW, g = tid / SW;
            constexpr int NG = kWideSolveThreads / SW;
            double s0 = 0.0, s1 = 0.0;
            if (c < sw)
                for (int j = kb + sw + g; j < n; j += NG) {
                    const double l = Lg[(size_t)j * n + kb + c];
                    s0 = __builtin_fma(l, xs[0][j], s0);
                    if (nvec == 2) s1 = __builtin_fma(l, xs[1][j], s1);
                }
            // (NG row groups of SW columns: the first NW of them land in red, the others are added by their owners below)
            static_assert(NG == NW || NG == 2 * NW, "row groups of the backward mat-vec");
            if (NG == 2 * NW) {
                s0 += __shfl_xor(s0, 32);  // SW = 32: groups g and g + 1 share a wave
                s1 += __shfl_xor(s1, 32);
            }
            if (NG == NW || (lane < 32)) {
                red[0][wave][c] = s0;
                red[1][wave][c] = s1;
            }
        }
        {
            const int j = tid % SW;
            if (j < sw)
                for (int i = tid / SW; i < sw; i += kWideSolveThreads / SW) P[i * ldp + j] = Lg[(size_t)(kb + i) * n + kb + j];
        }
        if (tid < sw) rdl[tid] = rdg[kb + tid];
        __syncthreads();
        if (tid < sw) {
            for (int v = 0; v < nvec; ++v) {
                double t = v == 0 ? Lg[(size_t)n * n + kb + tid] : (kb + tid < r ? zrand[kb + tid] : 0.0);
                for (int w = 0; w < NW; ++w) t -= red[v][w][tid];
                yv[v][tid] = t;
            }
        }
        __syncthreads();
        lds_backward<kWideSolveThreads>(P, ldp, sw, rdl, yv[0]);
        if (nvec == 2) lds_backward<kWideSolveThreads>(P, ldp, sw, rdl, yv[1]);
        if (tid < sw) {
            xs[0][kb + tid] = yv[0][tid];
            xs[1][kb + tid] = nvec == 2 ? yv[1][tid] : 0.0;
        }
        __syncthreads();
    }
}

// The solve itself keeps its own, specialised copy of the body (C = I + G, one right-hand side known at compile time): instantiating
// wide_solve_body for it costs 45 more spilled registers and 15 % of the kernel (135 -> 155 us at r = 256, tools/ubench_solve_wide.hip) --
// the register allocator's doing, not the arithmetic's; the transition density above ranks 112 uses the general body.
template <int SW>
__global__ __launch_bounds__(kWideSolveThreads) void posterior_solve_wide_kernel(int r, int n, const double *__restrict__ G,
                                                                                 const double *__restrict__ rhs,
                                                                                 const double *__restrict__ zrand, double *__restrict__ a,
                                                                                 DevState *__restrict__ st, double *gw) {
    extern __shared__ double P[];  // the panel: [rows][SW + 1]
    constexpr int ldp = SW + 1, NW = kWideSolveThreads / 64;
    __shared__ double rdl[SW];
    __shared__ double red[2][NW][SW];
    __shared__ double xs[2][512];  // x = L^-T y and, when sampling, L^-T z (n <= 512)
    __shared__ double yv[2][SW];
    __shared__ int bad_spd, bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double *Lg = gw;                          // [(n + 16)][n]: the factor, then the bordered rows
    double *rdg = gw + (size_t)(n + kNB) * n;  // [n] reciprocal diagonal
    const int nvec = zrand ? 2 : 1;
    if (tid == 0) bad_spd = 0, bad = 0;
    GINGR_STAGE_CLOCK(7)
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int NTJ = SW / 16;                                // tile columns of a panel
    constexpr int TSTEP = NW / NTJ;                             // wave w owns tile column w % NTJ, tile rows w / NTJ + TSTEP q
    constexpr int TACC = SW == 64 ? 9 : 9;                      // tiles per wave, at most: ceil((n + 16) / 16 / TSTEP), n <= 256 (512)
    constexpr int HALF = SW / 2;                                // 16-byte units per staged row
    constexpr int RSTEP = kWideSolveThreads / HALF;             // rows a staging round covers
    constexpr int PRE = 17;                                     // staging rounds, at most: ceil((n + 16) / RSTEP), n <= 256 (512)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int my_tj = wv % NTJ, ti0 = wv / NTJ;
    // the bordered rows of the workspace: the right-hand side, fifteen zero rows (they ride through every panel)
    for (int e = tid; e < kNB * n; e += kWideSolveThreads) Lg[(size_t)n * n + e] = e < r ? rhs[e] : 0.0;
    __syncthreads();
    const int srow = tid / HALF, scp = tid % HALF;              // this thread's row (per round) and column pair of a staged slice
    for (int kb = 0; kb < n; kb += SW) {
        const int sw = min(SW, n - kb), rows = n - kb + kNB;
        const int nti = rows >> 4, ntj = sw >> 4;
        // The panel's tiles, C - L[rows, 0 : kb] L[panel rows, 0 : kb]^T, accumulate in registers while the panel's LDS block stages
        // the operands: the factor's columns go through it SW at a time (coalesced 16-byte loads, the next slice requested before the
        // MFMAs of the current one), and C = I + G (bordered rows: the right-hand side) goes through it last and stays, minus the
        // accumulated products.
        // (Fragments fetched straight from the workspace, 8 bytes per lane and sixteen rows per instruction, made this stage 62 % of
        // the kernel: 354k cycles at r = 256.)
        v4f64 acc[TACC];
#pragma unroll
        for (int q = 0; q < TACC; ++q) acc[q] = v4f64{0, 0, 0, 0};
        const bool col_live = my_tj < ntj;
        const int nsl = kb / SW;  // slices of the factor; slice nsl is C
        d2 pre[PRE];
        auto fetch_slice = [&](int sl) __attribute__((always_inline)) {
            // slice nsl is C itself: the same rows and row stride, read from G; the bordered rows always come from the workspace
            const bool is_c = sl == nsl;
            const int col = sl * SW + 2 * scp;  // (== kb + 2 scp for C)
            const double *base = (is_c ? G : Lg) + col;
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                const int gi = kb + srow + u * RSTEP;
                if (gi < n + kNB) pre[u] = *reinterpret_cast<const d2 *>((gi < n ? base : Lg + col) + (size_t)gi * n);
            }
            if (is_c) {  // workgroup-uniform.  Mm = QtL Q + I, identity on the padding   (scalismo genericRegressionComputations)
#pragma unroll
                for (int u = 0; u < PRE; ++u) {
                    const int gi = kb + srow + u * RSTEP;
                    if (gi < n) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const double one = gi == col + h ? 1.0 : 0.0;
                            pre[u][h] = (gi < r && col + h < r) ? pre[u][h] + one : one;
                        }
                    }
                }
            }
        };
        fetch_slice(0);
        for (int sl = 0; sl <= nsl; ++sl) {
            __syncthreads();  // (the previous slice, or the previous panel's write-back, has been read)
#pragma unroll
            for (int u = 0; u < PRE; ++u)
                if (srow + u * RSTEP < rows) {
                    P[(srow + u * RSTEP) * ldp + 2 * scp] = pre[u][0];
                    P[(srow + u * RSTEP) * ldp + 2 * scp + 1] = pre[u][1];
                }
            __syncthreads();
            if (sl == nsl) break;
            fetch_slice(sl + 1);
            if (col_live) {  // wave-uniform
                const double *pb = P + (16 * my_tj + l15) * ldp + l4;
#pragma unroll
                for (int q = 0; q < TACC; ++q) {
                    const int ti = ti0 + TSTEP * q;
                    if (ti < nti && ti >= my_tj) {  // wave-uniform; strictly above the diagonal: nobody reads it
                        const double *pa = P + (16 * ti + l15) * ldp + l4;
#pragma unroll 8
                        for (int u = 0; u < SW / 4; ++u) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * u], pb[4 * u], acc[q], 0, 0, 0);
                    }
                }
            }
        }
        if (kb > 0 && col_live) {
#pragma unroll
            for (int q = 0; q < TACC; ++q) {
                const int ti = ti0 + TSTEP * q;
                if (ti < nti && ti >= my_tj) {
                    double *pc = P + (16 * ti + l4) * ldp + 16 * my_tj + l15;  // D[i = l4 + 4 g][j = l15]
#pragma unroll
                    for (int g = 0; g < 4; ++g) pc[4 * g * ldp] -= acc[q][g];
                }
            }
        }
        __syncthreads();
        GINGR_STAGE_CLOCK(0)
        lds_cholesky<kWideSolveThreads>(P, ldp, sw, rdl, &bad_spd, rows - sw);
        {   // the panel goes back to the workspace
            const int j = tid % SW;
            if (j < sw)
                for (int i = tid / SW; i < rows; i += kWideSolveThreads / SW) Lg[(size_t)(kb + i) * n + kb + j] = P[i * ldp + j];
        }
        if (tid < sw) rdg[kb + tid] = rdl[tid];
        GINGR_STAGE_CLOCK(4)
        // The next panel's first slice is requested straight away, and at kb = SW that slice IS the panel just written, by other
        // threads: without this barrier a rare race (one posterior in ~1 500 at rank 150 when other kernels share the device; found
        // through the three-shard tests, tools/experiments/stress_group2.py / stress_group3.py).
        __syncthreads();
    }
    // x = L^-T y: y = row n of the workspace (L^-1 rhs); the sampling direction L^-T z rides along.  Everything a block reads from
    // the workspace -- the rows below it for the mat-vec, its diagonal block, its reciprocal diagonal -- does not depend on the x of the
    // blocks behind it, so it is requested one block AHEAD, before the sequential substitution of the current block, and waits in
    // registers (round 6: two exposed memory round trips per block less).
    const int kb_last = ((n - 1) / SW) * SW;
    constexpr int NG = kWideSolveThreads / SW;          // row groups of the mat-vec
    constexpr int LV = SW == 64 ? 24 : 30;              // rows a thread takes below a block, at most: (n - SW) / NG
    constexpr int DV = SW * SW / kWideSolveThreads;     // diagonal-block entries per thread
    static_assert(NG == NW || NG == 2 * NW, "row groups of the backward mat-vec");
    const int bc = tid % SW, bg = tid / SW;
    double lv[LV], dv[DV], rdv = 0.0;
    auto prefetch_block = [&](int kb) __attribute__((always_inline)) {
        const int sw = min(SW, n - kb);
#pragma unroll
        for (int q = 0; q < LV; ++q) {
            const int j = kb + sw + bg + q * NG;
            lv[q] = (bc < sw && j < n) ? Lg[(size_t)j * n + kb + bc] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < DV; ++q) {
            const int i = bg + q * NG;
            dv[q] = (bc < sw && i < sw) ? Lg[(size_t)(kb + i) * n + kb + bc] : 0.0;
        }
        rdv = tid < sw ? rdg[kb + tid] : 0.0;
    };
    prefetch_block(kb_last);
    for (int kb = kb_last; kb >= 0; kb -= SW) {
        const int sw = min(SW, n - kb);
        {   // t_c = y_c - sum over the rows j below the block of L[j][kb + c] x_j: NG row groups, combined in order
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int q = 0; q < LV; ++q) {
                const int j = kb + sw + bg + q * NG;
                if (j < n) {
                    s0 = __builtin_fma(lv[q], xs[0][j], s0);
                    if (nvec == 2) s1 = __builtin_fma(lv[q], xs[1][j], s1);
                }
            }
            if (NG == 2 * NW) {
                s0 += __shfl_xor(s0, 32);  // SW = 32: groups g and g + 1 share a wave
                s1 += __shfl_xor(s1, 32);
            }
            if (NG == NW || (lane < 32)) {
                red[0][wave][bc] = s0;
                red[1][wave][bc] = s1;
            }
        }
#pragma unroll
        for (int q = 0; q < DV; ++q) {
            const int i = bg + q * NG;
            if (bc < sw && i < sw) P[i * ldp + bc] = dv[q];
        }
        if (tid < sw) rdl[tid] = rdv;
        const double ybase = tid < sw ? Lg[(size_t)n * n + kb + tid] : 0.0;  // (the forward-substituted right-hand side: written long ago)
        if (kb > 0) prefetch_block(kb - SW);
        __syncthreads();
        if (tid < sw) {
            for (int v = 0; v < nvec; ++v) {
                double t = v == 0 ? ybase : (kb + tid < r ? zrand[kb + tid] : 0.0);
                for (int w = 0; w < NW; ++w) t -= red[v][w][tid];
                yv[v][tid] = t;
            }
        }
        __syncthreads();
        lds_backward<kWideSolveThreads>(P, ldp, sw, rdl, yv[0]);
        if (nvec == 2) lds_backward<kWideSolveThreads>(P, ldp, sw, rdl, yv[1]);
        if (tid < sw) {
            xs[0][kb + tid] = yv[0][tid];
            xs[1][kb + tid] = nvec == 2 ? yv[1][tid] : 0.0;
        }
        __syncthreads();
    }
    GINGR_STAGE_CLOCK(5)
    GINGR_STAGE_CLOCK(6)
    for (int k = tid; k < n; k += kWideSolveThreads) {
        const double v = k < r ? xs[0][k] + xs[1][k] : 0.0;
        a[k] = v;
        if (!finite_d(v)) bad = 1;
    }
    __syncthreads();
    if (tid == 0) {
        if (bad_spd)
            st->err = GINGR_ERR_NOT_SPD;
        else if (bad)
            st->err = GINGR_ERR_NONFINITE;
    }
}

// posterior_logpdf_split_kernel for ranks above 112 (round 6; until then posterior_logpdf_lds_kernel<true>: both factorisations one
// after the other in ONE workgroup of 256 threads, 16-column panels over the global workspace).  Two workgroups of eight waves, the
// same algebra: workgroup 0 factors N = I + G and solves N a = rhs; workgroup 1 factors K = S_tot + eps N with the bordered row
// Q0^T e + eps rhs, so that w = K^-1 (Q0^T e + eps rhs), u = w - a and |c|^2 = u^T (N w - rhs); a travels through fx under the
// release / acquire pair on sync[0] (the launch number `epoch`), sync[1] carries workgroup 0's failure flag.  The factor of K, its
// reciprocal diagonal and a are left in fx ([n x n][n][n]) for posterior_logpdf_cached_kernel.  gw: two workspaces of
// posterior_work_doubles(n) / 2 doubles each.
template <int SW>
__global__ __launch_bounds__(kWideSolveThreads) void posterior_logpdf_wide_kernel(int r, int n, const double *__restrict__ G,
                                                                                  const double *__restrict__ rhs,
                                                                                  const double *__restrict__ Stot,
                                                                                  const double *__restrict__ qte, double *__restrict__ fx,
                                                                                  double *__restrict__ out2, unsigned *sync, unsigned epoch,
                                                                                  double *gw, int64_t gw_stride) {
    extern __shared__ double P[];
    __shared__ WideSolveLds<SW> Ls;
    __shared__ int failed0;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        wide_solve_body<SW, false>(P, Ls, r, n, G, 1.0, nullptr, 0.0, 1.0, rhs, nullptr, 0.0, nullptr, gw);
        for (int k = tid; k < n; k += kWideSolveThreads) fx[(int64_t)n * n + k] = k < r ? Ls.xs[0][k] : 0.0;
        __syncthreads();
        if (tid == 0) {
            sync[1] = (unsigned)Ls.bad_spd;
            __hip_atomic_store(&sync[0], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // a and the flag are visible before it
        }
        return;
    }
    double *gw1 = gw + gw_stride;
    wide_solve_body<SW, true>(P, Ls, r, n, G, GINGR_COEFF_NOISE, Stot, 1.0, GINGR_COEFF_NOISE, qte, rhs, GINGR_COEFF_NOISE, nullptr, gw1);
    const double *w = Ls.xs[0];
    // (G w)_k: the thread's column k (G is symmetric: coalesced over k), two halves of the row range, four chains each
    double *hv = P;  // [2][512] (the panel block is free)
    __syncthreads();
    for (int kk = tid & 255; kk < r; kk += 256) {
        const int half = tid >> 8;
        const int j0 = half ? (r + 1) / 2 : 0, j1 = half ? r : (r + 1) / 2;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = j0;
        for (; j + 3 < j1; j += 4) {
            s0 = __builtin_fma(G[(int64_t)j * n + kk], w[j], s0);
            s1 = __builtin_fma(G[(int64_t)(j + 1) * n + kk], w[j + 1], s1);
            s2 = __builtin_fma(G[(int64_t)(j + 2) * n + kk], w[j + 2], s2);
            s3 = __builtin_fma(G[(int64_t)(j + 3) * n + kk], w[j + 3], s3);
        }
        for (; j < j1; ++j) s0 = __builtin_fma(G[(int64_t)j * n + kk], w[j], s0);
        hv[half * 512 + kk] = (s0 + s1) + (s2 + s3);
    }
    // the state-only part for posterior_logpdf_cached_kernel: the factor of K (lower part; the workspace rows have stride n) and its
    // reciprocal diagonal
    for (int64_t e = tid; e < (int64_t)n * n / 2; e += kWideSolveThreads) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        reinterpret_cast<d2 *>(fx)[e] = reinterpret_cast<const d2 *>(gw1)[e];
    }
    for (int k = tid; k < n; k += kWideSolveThreads) fx[(int64_t)n * n + n + k] = gw1[(size_t)(n + kNB) * n + k];
    if (tid == 0) {
        while (__hip_atomic_load(&sync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(2);
        failed0 = (int)__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    double part = 0.0;
    for (int k = tid; k < r; k += kWideSolveThreads) {
        const double av = __hip_atomic_load(&fx[(int64_t)n * n + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double nw = (hv[k] + hv[512 + k]) + w[k];  // (N w)_k
        part = __builtin_fma(w[k] - av, nw - rhs[k], part);
    }
    double *redv = P + 1024;  // [kWideSolveThreads]
    redv[tid] = part;
    __syncthreads();
    for (int st2 = kWideSolveThreads / 2; st2 > 0; st2 >>= 1) {
        if (tid < st2) redv[tid] += redv[tid + st2];
        __syncthreads();
    }
    if (tid == 0) {
        const bool bad = Ls.bad_spd || failed0;
        out2[0] = bad ? __builtin_nan("") : -0.5 * redv[0] - 0.5 * (double)r * 1.8378770664093454836;  // log(2 pi)
        out2[1] = bad ? 1.0 : 0.0;
    }
}

// ---- pieces shared by the three transition-density kernels (256 threads: 128 entries x 2 column halves) -------------------------
// u[k] = qte[k] - (S_tot a)[k]: two threads per entry (column halves of the symmetric S_tot: coalesced), four loads in flight.
// Ends with the entries written but NOT yet synchronised.
__device__ __forceinline__ void logpdf_rhs(int r, int rp, const double *__restrict__ Stot, const double *__restrict__ qte, const double *av,
                                           double (*hv)[512], double *u) {
    const int tid = threadIdx.x;
    {
        const int k = tid & 127, half = tid >> 7;
        for (int kk = k; kk < r; kk += 128) {
            const int j0 = half ? (r + 1) / 2 : 0, j1 = half ? r : (r + 1) / 2;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = j0;
            for (; j + 3 < j1; j += 4) {
                s0 = __builtin_fma(Stot[(int64_t)j * rp + kk], av[j], s0);
                s1 = __builtin_fma(Stot[(int64_t)(j + 1) * rp + kk], av[j + 1], s1);
                s2 = __builtin_fma(Stot[(int64_t)(j + 2) * rp + kk], av[j + 2], s2);
                s3 = __builtin_fma(Stot[(int64_t)(j + 3) * rp + kk], av[j + 3], s3);
            }
            for (; j < j1; ++j) s0 = __builtin_fma(Stot[(int64_t)j * rp + kk], av[j], s0);
            hv[half][kk] = (s0 + s1) + (s2 + s3);
        }
    }
    __syncthreads();
    for (int k = tid; k < r; k += kSolveThreads) u[k] = qte[k] - (hv[0][k] + hv[1][k]);
}

// u^T (I + G) u, the same value in every thread (red: kSolveThreads doubles of LDS)
__device__ __forceinline__ double logpdf_quadratic(int r, int rp, const double *__restrict__ G, const double *u, double *red) {
    const int tid = threadIdx.x;
    double part = 0.0;
    {
        const int k = tid & 127, half = tid >> 7;
        for (int kk = k; kk < r; kk += 128) {
            const int j0 = half ? (r + 1) / 2 : 0, j1 = half ? r : (r + 1) / 2;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = j0;
            for (; j + 3 < j1; j += 4) {  // G symmetric: G[j][kk], coalesced over kk
                s0 = __builtin_fma(G[(int64_t)j * rp + kk], u[j], s0);
                s1 = __builtin_fma(G[(int64_t)(j + 1) * rp + kk], u[j + 1], s1);
                s2 = __builtin_fma(G[(int64_t)(j + 2) * rp + kk], u[j + 2], s2);
                s3 = __builtin_fma(G[(int64_t)(j + 3) * rp + kk], u[j + 3], s3);
            }
            for (; j < j1; ++j) s0 = __builtin_fma(G[(int64_t)j * rp + kk], u[j], s0);
            double g = (s0 + s1) + (s2 + s3);
            if (half == 0) g += u[kk];
            part = __builtin_fma(u[kk], g, part);
        }
    }
    red[tid] = part;
    __syncthreads();
    for (int st2 = kSolveThreads / 2; st2 > 0; st2 >>= 1) {
        if (tid < st2) red[tid] += red[tid + st2];
        __syncthreads();
    }
    return red[0];
}

// log-density of a mesh under the posterior model in scalismo's parameterisation:
//   posterior.gp.logpdf(posterior.coefficients(mesh))   (G/api/sampling/generators/GeneratorWrapperStochastic.scala:42-63)
// With N = Q0' L^-T (any square root of the posterior covariance gives the same norm) the ridge-regression coefficients are
//   c = (N^T N + eps I)^-1 N^T d = L^T (S_tot + eps (I + G))^-1 b,   b = Q0^T e - S_tot a,
// e = R^T(mesh - c - t) - (ref - c) - mean (model frame residual), a = posterior coefficients;  logpdf = -|c|^2/2 - r/2 log(2 pi).
template <bool GW>  // see posterior_solve_lds_kernel
__global__ __launch_bounds__(kSolveThreads) void posterior_logpdf_lds_kernel(int r, int rp, const double *__restrict__ G,
                                                                   const double *__restrict__ rhs,
                                                                   const double *__restrict__ Stot,
                                                                   const double *__restrict__ qte,
                                                                   double *__restrict__ fx /* nullable: [rp*rp] second factor, [rp] a, [rp] 1/diag */,
                                                                   double *__restrict__ out2, double *gwork) {
    extern __shared__ double lds_sm[];
    double *sm;
    if constexpr (GW)
        sm = gwork;
    else
        sm = lds_sm;
    const int n = rp, ld = solve_ld(n);
    double *A = sm;
    double *u = sm + (size_t)n * ld;  // extra row block of the bordered matrix
    double *rd = sm + (size_t)(n + kNB) * ld;
    __shared__ int bad_spd;
    __shared__ double red[kSolveThreads];
    __shared__ double av[512];  // posterior coefficients a (rp <= 512)
    __shared__ double hv[2][512];  // the two half sums of the mat-vecs
    static_assert(kSolveThreads == 256, "the mat-vecs below split 256 threads into 128 entries x 2 halves");
    const int tid = threadIdx.x;
    if (tid == 0) bad_spd = 0;
    // (1) a = (I + G)^-1 rhs: the posterior coefficients of the state (what posterior_solve_lds_kernel computes)
    for (int k = tid; k < kNB * ld; k += kSolveThreads) u[k] = k < r ? rhs[k] : 0.0;
    lds_load_spd<kSolveThreads>(A, ld, r, n, G, 1.0, nullptr, 0.0, 1.0);
    lds_cholesky<kSolveThreads>(A, ld, n, rd, &bad_spd, kNB);  // u <- L^-1 rhs on the way
    lds_backward<kSolveThreads>(A, ld, n, rd, u);
    for (int k = tid; k < rp; k += kSolveThreads) {
        av[k] = k < r ? u[k] : 0.0;
        if (fx) fx[(int64_t)rp * rp + k] = av[k];
    }
    __syncthreads();
    // (2) b = Q0^T e - S_tot a
    for (int k = tid; k < kNB * ld; k += kSolveThreads) u[k] = 0.0;
    __syncthreads();
    logpdf_rhs(r, rp, Stot, qte, av, hv, u);
    // (3) u = (S_tot + eps (I + G))^-1 b
    lds_load_spd<kSolveThreads, true>(A, ld, r, n, G, GINGR_COEFF_NOISE, Stot, 1.0, GINGR_COEFF_NOISE);
    lds_cholesky<kSolveThreads>(A, ld, n, rd, &bad_spd, kNB);                                        // u <- L2^-1 u on the way
    lds_backward<kSolveThreads>(A, ld, n, rd, u);
    __syncthreads();
    if (fx) {  // everything of the second system that depends on the state alone: posterior_logpdf_cached_kernel starts from here
        for (int i = tid >> 6; i < n; i += kSolveThreads / 64)  // one wave per row: no index division, coalesced
            for (int j = tid & 63; j < n; j += 64) fx[(int64_t)i * rp + j] = A[i * ld + j];
        for (int k = tid; k < n; k += kSolveThreads) fx[(int64_t)rp * rp + rp + k] = rd[k];
    }
    // (4) |c|^2 with c = L^T u, L L^T = I + G:  |c|^2 = u^T (I + G) u -- a quadratic form with G itself, so the first factor does
    //     not have to survive the second factorisation (it used to be parked in a global scratch and read back)
    const double n2 = logpdf_quadratic(r, rp, G, u, red);
    if (tid == 0) {
        out2[0] = bad_spd ? __builtin_nan("") : -0.5 * n2 - 0.5 * (double)r * 1.8378770664093454836;  // log(2 pi)
        out2[1] = bad_spd ? 1.0 : 0.0;
    }
}

// posterior_logpdf_lds_kernel on TWO workgroups (r <= 128): the factor of K = S_tot + eps (I + G) does not depend on the posterior
// coefficients a, so workgroup 1 factors it while workgroup 0 factors N = I + G and solves N a = rhs.  Round 4: neither does the
// right-hand side have to wait for a -- S_tot a = K a - eps N a = K a - eps rhs, hence
//     u = K^-1 (Q0^T e - S_tot a) = K^-1 (Q0^T e + eps rhs) - a =: w - a,        |c|^2 = u^T N u = u^T (N w - rhs)
// -- so workgroup 1 lets the forward solve of (Q0^T e + eps rhs) ride through its factorisation, substitutes back, forms N w - rhs,
// and only then picks a up (agent-scope release / acquire on sync[0], the value `epoch` of this launch; sync[1] carries workgroup 0's
// failure flag) for two vector operations: the mat-vec with S_tot, the forward solve and every wait are off the critical path
// (87 us in one workgroup -> 60 -> ~45).  keep != 0 leaves the factor of K, its reciprocal diagonal and a in fx for
// posterior_logpdf_cached_kernel (a Metropolis-Hastings step that asks again about the same state); a always travels through fx.
// Workgroup 0 is dispatched first, so workgroup 1 never waits for a workgroup that has no compute unit.
__global__ __launch_bounds__(kSolveThreads) void posterior_logpdf_split_kernel(int r, int rp, const double *__restrict__ G,
                                                                     const double *__restrict__ rhs,
                                                                     const double *__restrict__ Stot,
                                                                     const double *__restrict__ qte, double *__restrict__ fx,
                                                                     double *__restrict__ out2, unsigned *sync, unsigned epoch, int keep,
                                                                     double *__restrict__ nfac) {
    extern __shared__ double sm[];
    const int n = rp, ld = solve_ld(n);
    double *A = sm;
    double *u = sm + (size_t)n * ld;
    double *rd = sm + (size_t)(n + kNB) * ld;
    __shared__ int bad_spd;
    __shared__ double red[kSolveThreads];
    __shared__ double av[512];
    __shared__ double hv[2][512];
    const int tid = threadIdx.x;
    if (tid == 0) bad_spd = 0;
    if (blockIdx.x == 0) {
        // exactly the arithmetic of posterior_solve_lds_kernel<0> (identity rows through the panel solves, W-form backward substitution):
        // the coefficients a -- and the factor a later sampled proposal a + L^-T z is formed from -- carry the bits the solve kernel
        // would produce for this state
        double *y = u, *W = u + kNB * ld, *x = sm + (size_t)(n + 2 * kNB) * ld;
        for (int k = tid; k < kNB * ld; k += kSolveThreads) {
            y[k] = k < r ? rhs[k] : 0.0;
            const int c = k / ld, j = k - c * ld;
            W[k] = (j < n && (j & 15) == c) ? 1.0 : 0.0;
        }
        lds_load_spd<kSolveThreads>(A, ld, r, n, G, 1.0, nullptr, 0.0, 1.0);
        lds_cholesky<kSolveThreads>(A, ld, n, x, &bad_spd, kNB, kNB);  // y <- L^-1 rhs, W <- the transposed inverses of the diagonal blocks
        lds_backward_w<kSolveThreads>(A, ld, n, W, y, x);
        for (int k = tid; k < rp; k += kSolveThreads) fx[(int64_t)rp * rp + k] = k < r ? x[k] : 0.0;
        __syncthreads();
        if (tid == 0) {
            sync[1] = (unsigned)bad_spd;
            __hip_atomic_store(&sync[0], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // a and the flag are visible before it
        }
        if (nfac) {  // the factor for posterior_sample_cached_kernel: L, then the 16 W rows (the substitution above left both as they
                     // were) -- written BEHIND the hand-over of a (round 6): 115 KB that workgroup 1 does not have to wait for
            for (int i = tid >> 6; i < n + kNB; i += kSolveThreads / 64) {
                const double *src = i < n ? A + i * ld : W + (i - n) * ld;
                for (int j = tid & 63; j < n; j += 64) nfac[(int64_t)i * rp + j] = src[j];
            }
        }
        return;
    }
    GINGR_STAGE_CLOCK(7)
    // (round 6: the W form of the backward substitution here too -- sixteen identity rows ride through the panel solves and come back
    // as the transposed inverses of the diagonal blocks, as in workgroup 0; 12.5k -> ~6k cycles of this workgroup's critical path)
    double *Wk = u + kNB * ld, *rdk = sm + (size_t)(n + 2 * kNB) * ld, *wv = rdk + n;
    for (int k = tid; k < kNB * ld; k += kSolveThreads) {
        u[k] = k < r ? __builtin_fma(GINGR_COEFF_NOISE, rhs[k], qte[k]) : 0.0;
        const int c = k / ld, j = k - c * ld;
        Wk[k] = (j < n && (j & 15) == c) ? 1.0 : 0.0;
    }
    lds_load_spd<kSolveThreads, true>(A, ld, r, n, G, GINGR_COEFF_NOISE, Stot, 1.0, GINGR_COEFF_NOISE);
    GINGR_STAGE_CLOCK(0)
    lds_cholesky<kSolveThreads>(A, ld, n, rdk, &bad_spd, kNB, kNB);  // u <- L_K^-1 (Q0^T e + eps rhs) on the way
    lds_backward_w<kSolveThreads>(A, ld, n, Wk, u, wv);              // wv = w
    __syncthreads();
    rd = rdk;
    u = wv;
    GINGR_STAGE_CLOCK(4)
    {   // hv[0] + hv[1] = G w (two halves of the row range per column, coalesced over the column)
        const int k = tid & 127, half = tid >> 7;
        for (int kk = k; kk < r; kk += 128) {
            const int j0 = half ? (r + 1) / 2 : 0, j1 = half ? r : (r + 1) / 2;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = j0;
            for (; j + 3 < j1; j += 4) {
                s0 = __builtin_fma(G[(int64_t)j * rp + kk], u[j], s0);
                s1 = __builtin_fma(G[(int64_t)(j + 1) * rp + kk], u[j + 1], s1);
                s2 = __builtin_fma(G[(int64_t)(j + 2) * rp + kk], u[j + 2], s2);
                s3 = __builtin_fma(G[(int64_t)(j + 3) * rp + kk], u[j + 3], s3);
            }
            for (; j < j1; ++j) s0 = __builtin_fma(G[(int64_t)j * rp + kk], u[j], s0);
            hv[half][kk] = (s0 + s1) + (s2 + s3);
        }
    }
    if (keep) {  // the state-only part for posterior_logpdf_cached_kernel
        for (int i = tid >> 6; i < n; i += kSolveThreads / 64)
            for (int j = tid & 63; j < n; j += 64) fx[(int64_t)i * rp + j] = A[i * ld + j];
        for (int k = tid; k < n; k += kSolveThreads) fx[(int64_t)rp * rp + rp + k] = rd[k];
    }
    if (tid == 0) {
        while (__hip_atomic_load(&sync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(2);
        if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) bad_spd = 1;
    }
    __syncthreads();
    double part = 0.0;
    for (int k = tid; k < r; k += kSolveThreads) {
        const double a = __hip_atomic_load(&fx[(int64_t)rp * rp + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double nw = (hv[0][k] + hv[1][k]) + u[k];  // (N w)_k
        part = __builtin_fma(u[k] - a, nw - rhs[k], part);
    }
    red[tid] = part;
    __syncthreads();
    for (int st2 = kSolveThreads / 2; st2 > 0; st2 >>= 1) {
        if (tid < st2) red[tid] += red[tid + st2];
        __syncthreads();
    }
    if (tid == 0) {
        out2[0] = bad_spd ? __builtin_nan("") : -0.5 * red[0] - 0.5 * (double)r * 1.8378770664093454836;  // log(2 pi)
        out2[1] = bad_spd ? 1.0 : 0.0;
    }
    GINGR_STAGE_CLOCK(5)
    GINGR_STAGE_CLOCK(6)
    (void)av;
}

// The sampled proposal a + L^-T z of a state whose I + G the two-workgroup log-density kernel has factored already (nfac: [rp*rp] L,
// [16*rp] the W rows; a_mean): one load and one W-form backward substitution instead of the factorisation (32 -> ~12 us at r = 100),
// with the bits posterior_solve_lds_kernel<0> gives (same routines on the same factor, same order of the final addition).
__global__ __launch_bounds__(kSolveThreads) void posterior_sample_cached_kernel(int r, int rp, const double *__restrict__ nfac,
                                                                      const double *__restrict__ a_mean,
                                                                      const double *__restrict__ zrand, double *__restrict__ a,
                                                                      DevState *__restrict__ st) {
    extern __shared__ double sm[];
    const int n = rp, ld = solve_ld(n);
    double *A = sm;
    double *W = sm + (size_t)n * ld;
    double *y2 = sm + (size_t)(n + kNB) * ld;
    double *x = y2 + n;
    __shared__ int bad;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    lds_fill_rows<kSolveThreads>(A, ld, nfac, rp, n + kNB, n);  // (rows n .. n+15 of the copy are the W rows: same stride)
    for (int k = tid; k < n; k += kSolveThreads) y2[k] = k < r ? zrand[k] : 0.0;
    __syncthreads();
    lds_backward_w<kSolveThreads>(A, ld, n, W, y2, x);
    __syncthreads();
    for (int k = tid; k < rp; k += kSolveThreads) {
        const double v = k < r ? a_mean[k] + x[k] : 0.0;
        a[k] = v;
        if (!finite_d(v)) bad = 1;
    }
    __syncthreads();
    if (tid == 0 && bad) st->err = GINGR_ERR_NONFINITE;
}

// The same log-density for a state whose posterior_logpdf_lds_kernel has run before (fx: its posterior coefficients and the factor
// of S_tot + eps (I + G), both functions of the state alone): only the mesh-dependent part is left -- b, two triangular solves, the
// quadratic form.  A Metropolis-Hastings step asks for q(x' | x) of a state x that was the x' or the x of the step before.
template <bool GW>
__global__ __launch_bounds__(kSolveThreads) void posterior_logpdf_cached_kernel(int r, int rp, const double *__restrict__ G,
                                                                      const double *__restrict__ Stot,
                                                                      const double *__restrict__ qte,
                                                                      const double *__restrict__ fx, double *__restrict__ out2,
                                                                      double *gwork) {
    extern __shared__ double lds_sm[];
    double *sm;
    if constexpr (GW)
        sm = gwork;
    else
        sm = lds_sm;
    const int n = rp, ld = solve_ld(n);
    double *A = sm;
    double *u = sm + (size_t)n * ld;
    double *rd = sm + (size_t)(n + kNB) * ld;
    __shared__ double red[kSolveThreads];
    __shared__ double av[512];
    __shared__ double hv[2][512];
    const int tid = threadIdx.x;
    if constexpr (GW) {
        for (int i = tid >> 6; i < n; i += kSolveThreads / 64)
            for (int j = tid & 63; j < n; j += 64) A[i * ld + j] = fx[(int64_t)i * rp + j];
    } else {
        lds_fill_rows<kSolveThreads>(A, ld, fx, rp, n, n);
    }
    for (int k = tid; k < n; k += kSolveThreads) {
        rd[k] = fx[(int64_t)rp * rp + rp + k];
        av[k] = k < r ? fx[(int64_t)rp * rp + k] : 0.0;
    }
    for (int k = tid; k < kNB * ld; k += kSolveThreads) u[k] = 0.0;
    __syncthreads();
    logpdf_rhs(r, rp, Stot, qte, av, hv, u);  // step (2) of posterior_logpdf_lds_kernel, same order of operations
    __syncthreads();
    lds_forward<kSolveThreads>(A, ld, n, rd, u);
    lds_backward<kSolveThreads>(A, ld, n, rd, u);
    __syncthreads();
    const double n2 = logpdf_quadratic(r, rp, G, u, red);  // step (4)
    if (tid == 0) {
        out2[0] = -0.5 * n2 - 0.5 * (double)r * 1.8378770664093454836;  // log(2 pi)
        out2[1] = 0.0;
    }
}

// out[i] = (sum_j Binv[i][j] * p[j]) / eps: 16 lanes per output row, fixed-order shuffle reduction.
// Every lane of the 16-lane group returns the result.
__device__ __forceinline__ double binv_row_apply16(const double *__restrict__ Binv, const double *__restrict__ p, int r, int rp,
                                                   int i, int lane16) {
    double s = 0.0;
    if (i < r)
        for (int j = lane16; j < r; j += 16) s = __builtin_fma(Binv[(int64_t)i * rp + j], p[j], s);
    s = group16_sum(s);
    return s / GINGR_COEFF_NOISE;
}


__global__ __launch_bounds__(256) void coeff_solve_kernel(int r, int rp, const double *__restrict__ Binv,
                                                          const double *__restrict__ p, double *__restrict__ out) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), lane16 = threadIdx.x & 15;
    const double v = binv_row_apply16(Binv, p, r, rp, i, lane16);
    if (i < rp && lane16 == 0) out[i] = i < r ? v : 0.0;
}

// ------------------------------------------------------------------------------------------------- fused post-solve
// one-off r x r products at finalisation
// out = scale * A B on the r x r block, zero on the padding (A, B, out: [rp][rp], rp a multiple of 16, zero padded).  One wave per
// 16 x 16 tile on the matrix pipe; the operands are L2 resident (one thread per entry with a serial dot product was 160 us at
// rank 512, 28 of them per model).
__global__ __launch_bounds__(256) void small_gemm_kernel(int r, int rp, const double *__restrict__ A, const double *__restrict__ B, double scale,
                                                         double *__restrict__ out) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int nt = rp >> 4;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= nt * nt) return;
    const int ti = tile / nt, tj = tile - ti * nt;
    const double *pa = A + (int64_t)(16 * ti + l15) * rp + l4;  // A[i = l15][k = l4]
    const double *pb = B + (int64_t)l4 * rp + 16 * tj + l15;    // B[k = l4][j = l15]
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < rp; k0 += 16) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = pa[k0 + 4 * u];
            b[u] = pb[(int64_t)(k0 + 4 * u) * rp];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = 16 * ti + l4 + 4 * g, col = 16 * tj + l15;
        out[(int64_t)row * rp + col] = (row < r && col < r) ? acc[g] * scale : 0.0;
    }
}

// one-off, the model's PostVec block (column-major for the post-solve kernel: entry i of vector v at pvec[i * kRows + v]):
// v = blockIdx.x < 12: the moment vector in[v] itself (V[d][e], W[d]); v >= 12: Binv in[v - 12]
__global__ __launch_bounds__(256) void postvec_kernel(int r, int rp, const double *__restrict__ Binv, const double *__restrict__ in,
                                                      double *__restrict__ pvec) {
    const int v = blockIdx.x;
    const double *x = in + (int64_t)(v % 12) * rp;
    for (int i = threadIdx.x; i < rp; i += 256) {
        double s = 0.0;
        if (v < 12)
            s = x[i];
        else if (i < r)
            for (int j = 0; j < r; ++j) s = __builtin_fma(Binv[(int64_t)i * rp + j], x[j], s);
        pvec[(int64_t)i * PostVec::kRows + v] = s;
    }
}

// zbuf[b] = M_b x_b, b = blockIdx.x (PostVec layout: gp.h): b < 9: S[d][e] alpha; 9 <= b < 18: (Binv S[d][e]) alpha; 18 <= b < 27:
// (Binv S[d][e] C) a; b == 27: C a.  blockIdx.y picks a strip of 16 output rows, 16 lanes per row; a lane's (up to 32) matrix
// elements are all requested before the first FMA -- the matrices live in L2 / the Infinity Cache and a dependent load-FMA chain
// costs one round trip per element.
__global__ __launch_bounds__(256) void post_matvecs_kernel(int r, int rp, const double *__restrict__ mom,
                                                           const double *__restrict__ cmat, const double *__restrict__ alpha,
                                                           const double *__restrict__ a, double *__restrict__ zbuf) {
    const int b = blockIdx.x;
    const MomentLayout ml{rp};
    const double *Mat = b < 9 ? mom + ml.S(b / 3, b % 3) : (b < 27 ? cmat + (int64_t)(b - 8) * rp * rp : cmat);
    const double *src = b < 18 ? alpha : a;
    const int lane16 = threadIdx.x & 15;
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4);  // < rp (the grid covers rp / 16 strips)
    const double *row = Mat + (int64_t)i * rp;
    double m[32], x[32];  // rp <= 512
    const int nj = rp >> 4;
#pragma unroll
    for (int jj = 0; jj < 32; ++jj)
        if (jj < nj) {
            m[jj] = row[lane16 + 16 * jj];
            x[jj] = src[lane16 + 16 * jj];  // padding entries of alpha / a are zero
        }
    double s = 0.0;
#pragma unroll
    for (int jj = 0; jj < 32; ++jj)
        if (jj < nj) s = __builtin_fma(m[jj], x[jj], s);
    s = group16_sum(s);
    if (lane16 == 0) zbuf[(int64_t)i * PostVec::kZRows + b] = i < r ? s : 0.0;  // column-major: entry i of all 28 vectors is contiguous
}

// ---- post-solve: everything of GingrAlgorithm.update after the posterior coefficients (GingrAlgorithm.scala:212-246) in one small
// workgroup.  Round 4 rebuilt it around one measured fact (profiles/r03_exp_post_solve_code_touch.txt): run behind the long all-pairs
// kernels the round-3 kernel took 16-18 us against 6-9 us with warm caches -- it waited for its own CODE (20 KB of run-once
// straight-line float64 code through a cold instruction cache), not for its data.  So this version is built to EXECUTE few bytes:
//   * the second coefficient projection alpha' = Binv proj / eps is linear in the 3 x 3 pose quantities, proj = sum (B - I)_de V[d][e]
//     + sum B_de S[d][e] alpha_c + sum h_d W[d], so Binv is applied BEFORE the pose is known: Binv V / Binv W once per model
//     (gingr_model::pvec), (Binv S[d][e]) alpha and (Binv S[d][e] C) a by the mat-vec launch in front of this kernel.  No r x r matrix is
//     read here (round 3 copied the 100 KB of Binv through LDS and ran a mat-vec on it), one variant serves every rank <= 512;
//   * thread k keeps entry k of all 52 input vectors in registers: zbuf and pvec are stored entry-major ([rp][28], [rp][24]), so a
//     thread reads two contiguous runs with 16-byte loads at immediate offsets, all in flight at once; the 36 dot products of the
//     Umeyama sums are 36 multiplies per thread and ONE transposed reduction through LDS ([rp][37] products, 144 threads add a
//     quarter of a column each, fixed order) instead of 36 wave reductions;
//   * the 3 x 3 algebra of the pose step runs with one matrix entry per lane (a 3 x 3 product is three multiply-adds per lane, the
//     16 divisions by n are one division in 16 lanes) instead of every lane repeating all of it; the polar iteration and the
//     Euler round trip (svd3.h; the reference rebuilds R from the stored angles, so it cannot be dropped) are as before;
//   * the state is committed by 20 lanes from a staged copy instead of ~40 scalar stores of one thread.
// 16.5 KB + 3.9 KB of callees (1 024 threads, 138 KB of LDS)  ->  see tools/kernel_resources.sh / DESIGN.md section 4 for the figures.
//
// Failure semantics of GingrAlgorithm.update (G/api/GingrAlgorithm.scala:192-254):
//   posterior failed (Try of computePosterior, here: the solve flagged st->err)
//       iteration 0                      -> state unchanged                                          (:206-208)
//       iteration > 0, deterministic     -> ModelFlexibilityError                                    (:203-205)
//       iteration > 0, probabilistic     -> retryCounter == 0 ? ModelFlexibilityError
//                                           : { retryCounter -= 1; state unchanged }                  (:196-202)
//   posterior fine                       -> retryCounter = min(10, retryCounter + 1)                  (:210)
//       a coefficients() projection (or the alignment between them) failed -> ModelFlexibilityError at ANY iteration
//                                                                                                     (:248-251)
// Non-finite values count as failures: in the reference they make Breeze's SVD throw inside the Try.

// scratch of the pose step (doubles in LDS)
struct PoseLds {
    double D[36];     // the dot products: [0..2] W[d].alpha, [3..5] W[d].alpha_c, [6..14] V[b][d].alpha (index d*3+b),
                      // [15..23] V[d][b].alpha_c, [24..32] alpha_c.za[d][b], [33..35] alpha.za[d][d]
    double R[9];      // rotation of the current state
    double su[3], sv[3], gt[3], q[3];
    double Mvu[9];
    double sums[16];  // [0..2] sum x~, [3..5] sum y~, [6..14] sum y~ x~^T (row-major), [15] sum |x~|^2; x~ = x - c0, y~ = y - c0
    double qn[16];    // sums / n
    double Sxy[9];
    double R2[9];
    double coef[21];  // (B - I)[9], B[9], h[3]: the 3 x 3 quantities of the second projection, B = R2^T R
    double stage[20]; // the committed DevState: R[9], euler[3], center[3], t[3], scale, sigma2
};

// One wave: Umeyama between the current shape u~_i = p~_i + Q0_i alpha (unposed) and the blended posterior mean
// newshape - c0 = R v~_i + g~, v~_i = p~_i + Q0_i alpha_c, from the 36 dot products (moment form), then the 3 x 3 quantities of the
// second projection e_i = R2^T (newshape_i - t2) - p_i = (B - I) p~_i + B Q0_i alpha_c + h.  Lane l < 9 owns matrix entry (l / 3, l % 3).
// cst (device): Pp[9] = sum p~ p~^T, Ps[3] = sum p~, c0[3], n  (PostVec::consts)
__device__ void post_pose_step(const PostSolveArgs &A, const DevState *st, const double *__restrict__ cst, PoseLds &L, int *bad) {
    const int l = threadIdx.x & 63;
    const int l9 = l < 9 ? l : 8, a = l9 / 3, b = l9 - 3 * a, l3 = l < 3 ? l : 2;
    const double Ppl = cst[l9], Psl = cst[9 + l3], c0l = cst[12 + l3], n = cst[15];
    const double Rl = st->R[l9], cenl = st->center[l3], tl = st->t[l3];
    if (l < 9) {
        L.R[l] = Rl;
        L.Mvu[l] = Ppl + L.D[6 + l] + L.D[15 + l] + L.D[24 + l];
    }
    if (l < 3) {
        L.su[l] = Psl + L.D[l];
        L.sv[l] = Psl + L.D[3 + l];
        L.q[l] = c0l - cenl;
    }
    double s15 = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) s15 += cst[4 * d] + 2.0 * L.D[6 + 4 * d] + L.D[33 + d];
    __builtin_amdgcn_wave_barrier();
    if (l < 3) L.gt[l] = L.R[3 * l] * L.q[0] + L.R[3 * l + 1] * L.q[1] + L.R[3 * l + 2] * L.q[2] + cenl + tl - c0l;
    __builtin_amdgcn_wave_barrier();
    if (l < 3) {
        L.sums[l] = L.su[l];
        L.sums[3 + l] = L.R[3 * l] * L.sv[0] + L.R[3 * l + 1] * L.sv[1] + L.R[3 * l + 2] * L.sv[2] + n * L.gt[l];
    }
    if (l < 9) L.sums[6 + l] = L.R[3 * a] * L.Mvu[b] + L.R[3 * a + 1] * L.Mvu[3 + b] + L.R[3 * a + 2] * L.Mvu[6 + b] + L.gt[a] * L.su[b];
    if (l == 15) L.sums[15] = s15;
    __builtin_amdgcn_wave_barrier();
    double R2[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, eul[3] = {0, 0, 0}, t2[3] = {0, 0, 0}, c = 1.0;
    const double c0[3] = {cst[12], cst[13], cst[14]};
    bool fin = true;
    if (A.global_transform != GINGR_NO_TRANSFORMS) {  // (identityTransformation otherwise, GingrAlgorithm.scala:230)
        L.qn[l & 15] = L.sums[l & 15] / n;  // mu_x, mu_y, the second moments and the variance term: one division in 16 lanes
        __builtin_amdgcn_wave_barrier();
        if (l < 9) L.Sxy[l] = L.qn[6 + l] - L.qn[3 + a] * L.qn[b];
        const double mux[3] = {L.qn[0], L.qn[1], L.qn[2]}, muy[3] = {L.qn[3], L.qn[4], L.qn[5]};
        const double sig2x = L.qn[15] - (mux[0] * mux[0] + mux[1] * mux[1] + mux[2] * mux[2]);
        __builtin_amdgcn_wave_barrier();
        double S[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) S[q] = L.Sxy[q];
        // R = U diag(1, 1, sign det) V^T and, for similarity transforms, c = (d1 + d2 + sign d3) / var_x.  With det > 0 (every
        // non-degenerate registration) R is the polar factor of Sxy and d1 + d2 + d3 = trace(R^T Sxy): no SVD (svd3.h)
        double Rr[9], trace_ds = 0.0;
        if (!polar3_rotation(S, Rr, &trace_ds)) {
            double U[9], Dg[3], V[9];
            svd3(S, U, Dg, V);
            const double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
            const double s3 = det < 0 ? -1.0 : 1.0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Rr[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + s3 * U[i * 3 + 2] * V[j * 3 + 2];
            trace_ds = Dg[0] + Dg[1] + s3 * Dg[2];
        }
        c = (A.global_transform == GINGR_SIMILARITY_TRANSFORMS) ? trace_ds / sig2x : 1.0;
        // t = mu_y - c R mu_x in absolute coordinates (rotation about the origin)
        double mxa[3], mya[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            mxa[i] = mux[i] + c0[i];
            mya[i] = muy[i] + c0[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) t2[i] = mya[i] - c * (Rr[i * 3] * mxa[0] + Rr[i * 3 + 1] * mxa[1] + Rr[i * 3 + 2] * mxa[2]);
        // the registration result carries its rotation as Euler angles (rigid3DLandmarkRegistration builds Rotation3D)
        rot_to_euler_wave(Rr, eul);
        euler_to_rot_wave(eul, R2);
        fin = finite_d(c);
#pragma unroll
        for (int q = 0; q < 9; ++q) fin = fin && finite_d(R2[q]);
#pragma unroll
        for (int q = 0; q < 3; ++q) fin = fin && finite_d(t2[q]);
    }
    if (l == 0) {
        if (!fin) *bad = 1;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            L.R2[q] = R2[q];
            L.stage[q] = R2[q];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            L.stage[9 + q] = eul[q];
            L.stage[12 + q] = 0.0;  // Umeyama about Point(0,0,0), GingrAlgorithm.scala:81,266
            L.stage[15 + q] = t2[q];
        }
        L.stage[18] = c;
        if (A.is_icp) {
            const double ns = st->sigma2 - A.icp_step;  // ICP.scala:96-99
            L.stage[19] = ns > A.icp_end ? ns : A.icp_end;
        } else {
            const double *sc = A.scalars;  // CPD.scala:142-145
            L.stage[19] = (sc[1] - 2 * sc[2] + sc[3]) / (sc[0] * 3.0);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // B = R2^T R, h = R2^T (g~ + c0 - t2) - c0   (p_i = p~_i + c0)
    if (l < 9) {
        const double v = L.R2[a] * L.R[b] + L.R2[3 + a] * L.R[3 + b] + L.R2[6 + a] * L.R[6 + b];
        L.coef[l] = v - (a == b ? 1.0 : 0.0);
        L.coef[9 + l] = v;
    }
    if (l < 3) {
        const double w0 = L.gt[0] + c0[0] - t2[0], w1 = L.gt[1] + c0[1] - t2[1], w2 = L.gt[2] + c0[2] - t2[2];
        L.coef[18 + l] = (L.R2[l] * w0 + L.R2[3 + l] * w1 + L.R2[6 + l] * w2) - c0l;
    }
}

constexpr int kPostMinThreads = 256;

// A.zbuf: [rp][28] of launch_post_matvecs; A.pvec: the model's PostVec block.  blockDim = max(256, rp rounded up to 64);
// dynamic LDS: 37 rp doubles.
__global__ __launch_bounds__(512) void post_solve_kernel(PostSolveArgs A) {
    extern __shared__ double prod[];  // [rp][37]: the 36 products of entry k (odd row stride: the column sums below spread over the banks)
    __shared__ PoseLds L;
    __shared__ int bad;
    constexpr int ld = 37;
    const int r = A.r, rp = A.rp, tid = threadIdx.x;
    DevState *st = A.state;
    if (tid == 0 && A.zero_slot) *A.zero_slot = 0.0;  // see SweepArgs::absmax_slot: the fit pass behind this kernel takes a maximum into it
    if (st->status == GINGR_FIT_MODEL_FLEXIBILITY_ERROR || st->stopped) return;  // a failed fit stays as it is (run stops, :149-157);
                                                                                 // so does one the run's own rule stopped at
    const PostVec pvl{rp};
    // ---- column tid of every input vector: all loads in flight at once
    double za[9], bz[9], V[9], W[3], BV[9], BW[3], al = 0.0, ac = 0.0;
    if (tid < rp) {
        const double *zb = A.zbuf + (int64_t)tid * PostVec::kZRows, *pv = A.pvec + (int64_t)tid * PostVec::kRows;
        double bs[9], bt[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            za[q] = zb[PostVec::kZa + q];
            bs[q] = zb[PostVec::kBSa + q];
            bt[q] = zb[PostVec::kBTa + q];
            V[q] = pv[PostVec::kV + q];
            BV[q] = pv[PostVec::kBV + q];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            W[q] = pv[PostVec::kW + q];
            BW[q] = pv[PostVec::kBW + q];
        }
        const double a1 = zb[PostVec::kA1];  // alpha_1 = C a: coefficients of the posterior mean (transformedModelInit.coefficients,
        al = tid < r ? A.alpha[tid] : 0.0;        // :212-216; Q^T (Q a) = S_tot a and the R / R^T round trip of the displacement cancels)
        ac = tid < r ? al + (a1 - al) * A.step : 0.0;  // the step blend (:218-220)
        // Binv S[d][e] alpha_c = (1 - step) (Binv S[d][e]) alpha + step (Binv S[d][e] C) a
#pragma unroll
        for (int q = 0; q < 9; ++q) bz[q] = (1.0 - A.step) * bs[q] + A.step * bt[q];
        // ---- the 36 products of the dot products (PoseLds::D), column tid
        double *p = prod + tid * ld;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            p[d] = W[d] * al;
            p[3 + d] = W[d] * ac;
            p[33 + d] = al * za[4 * d];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                p[6 + d * 3 + b] = V[b * 3 + d] * al;
                p[15 + d * 3 + b] = V[d * 3 + b] * ac;
                p[24 + d * 3 + b] = ac * za[d * 3 + b];
            }
        }
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    if (tid < 144) {  // thread (t, part) adds the products k = part, part + 4, ... of dot product t; then (p0 + p1) + (p2 + p3)
        const int t = tid >> 2, part = tid & 3;
        const double *colp = prod + part * ld + t;
        double s = 0.0;
        for (int k = 0; k < rp; k += 4) s += colp[k * ld];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0) L.D[t] = s;
    }
    __syncthreads();
    if (tid < 64) post_pose_step(A, st, A.pvec + pvl.consts(), L, &bad);
    __syncthreads();
    // ---- second projection (transformedModel.coefficients(newshape), :234-237), Binv already applied to every term
    double anew = 0.0;
    if (tid < rp) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            s = __builtin_fma(L.coef[q], BV[q], s);
            s = __builtin_fma(L.coef[9 + q], bz[q], s);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) s = __builtin_fma(L.coef[18 + d], BW[d], s);
        anew = tid < r ? s / GINGR_COEFF_NOISE : 0.0;
        if (!finite_d(anew)) bad = 1;
    }
    __syncthreads();
    const bool posterior_failed = st->err != 0;
    const bool failed = posterior_failed || bad != 0;
    const double sigma2_before = st->sigma2;
    __syncthreads();  // (every thread has read st->err and sigma2 before they are rewritten)
    if (!failed) {
        if (tid < rp) A.alpha[tid] = anew;
        if (tid < 20) reinterpret_cast<double *>(st)[tid] = L.stage[tid];  // R, euler, center, t, scale, sigma2
    }
    if (tid == 0) {
        if (posterior_failed) {
            if (st->iteration > 0) {
                if (A.probabilistic && A.retry && *A.retry > 0)
                    *A.retry -= 1;
                else
                    st->status = GINGR_FIT_MODEL_FLEXIBILITY_ERROR;
            }
        } else {
            if (A.retry) *A.retry = *A.retry + 1 < GINGR_RETRY_INIT ? *A.retry + 1 : GINGR_RETRY_INIT;
            if (bad != 0) st->status = GINGR_FIT_MODEL_FLEXIBILITY_ERROR;
        }
        st->pad = failed ? (st->err != 0 ? st->err : GINGR_ERR_NONFINITE) : 0;  // last error, readable by the host
        st->err = 0;
        st->iteration += 1;  // GingrGeneratorWrapper.propose: updateIteration()
        // the dropWhile of GingrAlgorithm.run (:142-153) looks at (last state, this state): converged -- or failed -- and the chain ends HERE
        if (A.stop_threshold >= 0.0 && fabs(sigma2_before - (failed ? sigma2_before : L.stage[19])) < A.stop_threshold) st->stopped = 1;
    }
}

__global__ void state_init_kernel(DevState *st, const gingr_state_scalars *h, double *zero_slot) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    state_init_body(st, h, zero_slot);
}

}  // namespace

// =====================================================================================================  launchers
int sweep_num_blocks(int64_t M) {
    const int64_t nb = ceil_div(M, kGroups);
    return (int)(nb < kSweepMaxBlocks ? (nb > 0 ? nb : 1) : kSweepMaxBlocks);
}

int64_t sweep_ws_doubles(int64_t M, int32_t rp) {
    const int w = rp > 24 ? rp : 24;
    return (int64_t)sweep_num_blocks(M) * w;
}

template <int MODE>
static void launch_sweep_mode(gingr_ctx *ctx, const SweepArgs &a, int width) {
    if (MODE == SWEEP_FIT && a.qboxes && a.rp <= 512) {  // one workgroup per 64-point quarter (gp.h: SweepArgs::qboxes)
        const int nq = (int)std::min<int64_t>(4096, ceil_div(a.M, 64));
        const size_t l2 = (size_t)(a.rp + kGroups * 8 + 8) * sizeof(double);
        TimerScope ts(ctx, 4);
        if (a.rp <= 64)
            hipLaunchKernelGGL((sweep_fit_boxes_kernel<4, 4>), dim3(nq), dim3(kSweepThreads), l2, ctx->stream, a);
        else if (a.rp <= 112)  // (rank 100: 84 basis values per thread in flight; eight column blocks would spill into AGPRs)
            hipLaunchKernelGGL((sweep_fit_boxes_kernel<7, 4>), dim3(nq), dim3(kSweepThreads), l2, ctx->stream, a);
        else if (a.rp <= 128)
            hipLaunchKernelGGL((sweep_fit_boxes_kernel<8, 4>), dim3(nq), dim3(kSweepThreads), l2, ctx->stream, a);
        else if (a.rp <= 192)
            hipLaunchKernelGGL((sweep_fit_boxes_kernel<12, 1>), dim3(nq), dim3(kSweepThreads), l2, ctx->stream, a);
        else if (a.rp <= 256)
            hipLaunchKernelGGL((sweep_fit_boxes_kernel<16, 1>), dim3(nq), dim3(kSweepThreads), l2, ctx->stream, a);
        else if (a.rp <= 384)
            hipLaunchKernelGGL((sweep_fit_boxes_kernel<24, 1>), dim3(nq), dim3(kSweepThreads), l2, ctx->stream, a);
        else
            hipLaunchKernelGGL((sweep_fit_boxes_kernel<32, 1>), dim3(nq), dim3(kSweepThreads), l2, ctx->stream, a);
        return;
    }
    const int nb = sweep_num_blocks(a.M);
    const size_t lds = (size_t)(2 * a.rp + kGroups * (a.rp > 16 ? a.rp : 16)) * sizeof(double);
    TimerScope ts(ctx, 4);
    if (a.rp <= 128) {
        hipLaunchKernelGGL((sweep_kernel<MODE, 8>), dim3(nb), dim3(kSweepThreads), lds, ctx->stream, a);
    } else {
        if (lds > 48 * 1024)
            set_dynamic_lds(&sweep_kernel<MODE, 32>, (size_t)(lds));
        hipLaunchKernelGGL((sweep_kernel<MODE, 32>), dim3(nb), dim3(kSweepThreads), lds, ctx->stream, a);
    }
    ts.stop();
    if (width > 0 && !a.no_reduce)
        hipLaunchKernelGGL(block_partials_reduce_kernel, dim3((unsigned)width), dim3(256), 0, ctx->stream, a.partial, nb,
                           width, a.out);
}

void launch_sweep(gingr_ctx *ctx, SweepMode mode, const SweepArgs &a) {
    switch (mode) {
        case SWEEP_RHS: launch_sweep_mode<SWEEP_RHS>(ctx, a, a.rp); break;
        case SWEEP_PROJ1: launch_sweep_mode<SWEEP_PROJ1>(ctx, a, a.rp); break;
        case SWEEP_SHAPES: launch_sweep_mode<SWEEP_SHAPES>(ctx, a, 24); break;
        case SWEEP_PROJ2: launch_sweep_mode<SWEEP_PROJ2>(ctx, a, a.rp); break;
        case SWEEP_FIT: launch_sweep_mode<SWEEP_FIT>(ctx, a, 0); break;
        case SWEEP_POSED: launch_sweep_mode<SWEEP_POSED>(ctx, a, 0); break;
        case SWEEP_RHS_ICP: launch_sweep_mode<SWEEP_RHS_ICP>(ctx, a, a.rp); break;
    }
}

static void gram_plan(int64_t M, int32_t rp, int *nbp, int *npatch, int *nslabs, int64_t *rows_per_slab) {
    const int64_t rows = 3 * M;
    *nbp = (rp + 63) / 64;
    *npatch = *nbp * (*nbp + 1) / 2;
    int64_t want = 768 / *npatch;  // ~3 workgroups of 4 waves per CU
    if (want < 1) want = 1;
    // a slab writes a full rp x rp partial: below ~256 rows per slab the partials cost more than the rows they summarise
    const int64_t max_slabs = ceil_div(rows, 256);
    if (want > max_slabs) want = max_slabs;
    if (want < 1) want = 1;
    *rows_per_slab = round_up(ceil_div(rows, want), 16);
    *nslabs = (int)ceil_div(rows, *rows_per_slab);
}

// slabs of gram_tri_kernel (one workgroup each): 256 = one per CU; small shards keep at least 64 rows per slab
static void gram_tri_plan(int64_t M, int *nslabs, int64_t *rows_per_slab) {
    const int64_t rows = 3 * M;
    const int64_t want = std::min<int64_t>(256, std::max<int64_t>(1, ceil_div(rows, 64)));
    *rows_per_slab = round_up(ceil_div(rows, want), 16);
    *nslabs = (int)ceil_div(rows, *rows_per_slab);
}

int64_t gram_ws_doubles(int64_t M, int32_t rp) {
    int nbp, npatch, nslabs, nslabs_tri;
    int64_t rps;
    gram_plan(M, rp, &nbp, &npatch, &nslabs, &rps);
    gram_tri_plan(M, &nslabs_tri, &rps);
    const int64_t n = (int64_t)std::max(nslabs, nslabs_tri) * rp * rp;
    return rp >= 128 ? std::max(n, gram_wide_ws_doubles(M, rp)) : n;
}

int launch_gram(gingr_ctx *ctx, const double *Q0, int64_t M, int32_t rp, const double *weight, double *ws, double *G, const double *evec,
                double *rhs_partial, bool *rhs_done, const ZeroGate *gate) {
    if (rhs_done) *rhs_done = false;
    int nbp, npatch, nslabs;
    int64_t rps;
    gram_plan(M, rp, &nbp, &npatch, &nslabs, &rps);
    {
        const int nt = rp / 16;
        if (nt <= 7) {
            TimerScope ts(ctx, 2);
            // whole upper triangle per wave: one workgroup per slab; ~3 slabs' worth of waves per SIMD is not needed (one wave per
            // SIMD, deep prefetch), so 256 slabs = one workgroup per CU
            gram_tri_plan(M, &nslabs, &rps);
            const bool fuse = evec && rhs_partial && rhs_done;  // Q0^T evec out of the same pass: [nslabs][rp] partials
            if (fuse) *rhs_done = true;
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(nslabs), dim3(512), 0, ctx->stream, Q0, 3 * M, (int)rp, weight, rps, ws,
                                   fuse ? evec : (const double *)nullptr, M, fuse ? rhs_partial : (double *)nullptr, gate ? *gate : ZeroGate{});
            };
            const bool full = fuse && weight;
#define GINGR_GRAM_TRI(n) \
    case n: \
        if (full) go(gram_tri_kernel<n, true>); \
        else go(gram_tri_kernel<n, false>); \
        break;
            switch (nt) {
                GINGR_GRAM_TRI(1)
                GINGR_GRAM_TRI(2)
                GINGR_GRAM_TRI(3)
                GINGR_GRAM_TRI(4)
                GINGR_GRAM_TRI(5)
                GINGR_GRAM_TRI(6)
                default:
                    if (full) go(gram_tri_kernel<7, true>);
                    else go(gram_tri_kernel<7, false>);
                    break;
            }
#undef GINGR_GRAM_TRI
        } else {
            // rp >= 128: eight waves share the triangle (gp_wide.hip); the right-hand side rides along whenever it is asked for
            const bool fuse = evec && rhs_partial && rhs_done;
            if (fuse) *rhs_done = true;
            nslabs = launch_gram_wide(ctx, Q0, M, rp, weight, ws, fuse ? evec : nullptr, fuse ? rhs_partial : nullptr, gate);
        }
    }
    if (G)  // nullptr: the caller reduces the slab partials itself (launch_phase1_finalize with the returned slab count)
        hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)ceil_div((int64_t)rp * rp, 32)), dim3(256), 0, ctx->stream, ws,
                           nslabs, (int)rp, 1, G);
    return nslabs;
}

int launch_gram_downdate(gingr_ctx *ctx, const double *Q0, int64_t M, int32_t rp, const double *weight, double *ws, const ZeroGate *gate) {
    int nslabs_tri;  // never more slabs than the weighted Gram pass would write: the workspace behind them belongs to the right-hand-side sweep
    int64_t rows_per_slab;
    gram_tri_plan(M, &nslabs_tri, &rows_per_slab);
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(std::min(nslabs_tri, 128), ceil_div(M, 64)));
    const int64_t vps = ceil_div(M, want);
    const int nslabs = (int)ceil_div(M, vps);
    const int np = (rp + 111) / 112;  // 112-column patches; the upper ones only
    hipLaunchKernelGGL(gram_downdate_kernel, dim3((unsigned)nslabs, (unsigned)(np * (np + 1) / 2)), dim3(256), 0, ctx->stream, Q0, M, (int)rp, weight,
                       vps, ws, gate ? *gate : ZeroGate{});
    return nslabs;
}

void launch_phase1_finalize(gingr_ctx *ctx, const Phase1FinalizeArgs &a) {
    const int nG = (a.nslabs > 0 || a.scaled_src) ? (a.rp * a.rp + 31) / 32 : 0;
    hipLaunchKernelGGL(phase1_finalize_kernel, dim3((unsigned)(nG + a.rp + 1)), dim3(256), 0, ctx->stream, a);
}

namespace {
// S[d][e][a][b] = T[d rp + a][e rp + b], T the (3 rp) x (3 rp) product of the three-rows-per-point view
__global__ __launch_bounds__(256) void moment_scatter_kernel(const double *__restrict__ T, int rp, MomentLayout ml, double *__restrict__ mom) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t w = 3 * (int64_t)rp;
    if (idx >= w * w) return;
    const int I = (int)(idx / w), J = (int)(idx - (int64_t)I * w);
    const int d = I / rp, a = I - d * rp, e = J / rp, b = J - e * rp;
    mom[ml.S(d, e) + (int64_t)a * rp + b] = T[idx];
}
// out = in^T ([rp][rp])
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ in, int rp, double *__restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rp * rp) return;
    const int i = idx / rp, j = idx - i * rp;
    out[idx] = in[j * rp + i];
}
bool moments_as_rows(int32_t rp) { return 3 * rp >= 128 && 3 * rp <= 512; }
}  // namespace

int64_t moment_grams_ws_doubles(int64_t M, int32_t rp) {
    if (moments_as_rows(rp)) return gram_rows_ws_doubles(M, 3 * rp) + 9 * (int64_t)rp * rp;
    return gram_ws_doubles(M, rp);
}

// The nine blocks are the blocks of Z^T Z with Z the basis read as M rows of width 3 rp (the rows 3i, 3i+1, 3i+2 of a point are
// contiguous), so for 3 rp in 128 .. 512 (ranks 43 .. 170) they are ONE symmetric product on the triangle kernel of gp_wide.hip
// instead of nine general ones (0.9 ms -> 0.1 ms at rank 100).  Outside that range: the six blocks d <= e by gram_kernel, the other
// three by transposition.
void launch_moment_grams(gingr_ctx *ctx, const double *Q0, int64_t M, int32_t rp, double *ws, double *mom) {
    const MomentLayout ml{rp};
    if (moments_as_rows(rp)) {
        const int32_t w = 3 * rp;
        double *T = ws + gram_rows_ws_doubles(M, w);
        const int nslabs = launch_gram_rows(ctx, Q0, M, w, ws);
        hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)ceil_div((int64_t)w * w, 32)), dim3(256), 0, ctx->stream, ws, nslabs, (int)w, 1, T);
        hipLaunchKernelGGL(moment_scatter_kernel, dim3((unsigned)ceil_div((int64_t)w * w, 256)), dim3(256), 0, ctx->stream, T, (int)rp, ml, mom);
        return;
    }
    // logical rows = points; same slab plan as the weighted Gram (its workspace is large enough: nslabs is capped by rows/64)
    int nbp, npatch, nslabs;
    int64_t rps;
    gram_plan(M, rp, &nbp, &npatch, &nslabs, &rps);
    rps = round_up(ceil_div(M, nslabs), 16);
    nslabs = (int)ceil_div(M, rps);
    for (int d = 0; d < 3; ++d)
        for (int e = d; e < 3; ++e) {
            hipLaunchKernelGGL(gram_kernel, dim3(nslabs, nbp * nbp), dim3(256), 0, ctx->stream, Q0, M, (int)rp, (const double *)nullptr, rps, nbp,
                               3, d, e, 1, ws);
            hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)ceil_div((int64_t)rp * rp, 32)), dim3(256), 0, ctx->stream, ws, nslabs, (int)rp, 0,
                               mom + ml.S(d, e));
            if (e > d)
                hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div((int64_t)rp * rp, 256)), dim3(256), 0, ctx->stream, mom + ml.S(d, e),
                                   (int)rp, mom + ml.S(e, d));
        }
}

void launch_centered_mean(gingr_ctx *ctx, const gingr_model *m, double *ptil) {
    hipLaunchKernelGGL(centered_mean_kernel, dim3((unsigned)ceil_div(m->M, 256)), dim3(256), 0, ctx->stream, m->ref, m->mean,
                       m->M, m->c0[0], m->c0[1], m->c0[2], ptil);
}

void launch_obs_cpd(gingr_ctx *ctx, const gingr_model *m, const DevState *st, Cloud fit, const double *P1,
                    const double *PX, double lambda, const int32_t *lm_mask, double *weight, double *evec) {
    hipLaunchKernelGGL(obs_cpd_kernel, dim3((unsigned)ceil_div(m->M, 256)), dim3(256), 0, ctx->stream, m->ref, m->mean, m->M,
                       st, fit, P1, PX, lambda, lm_mask, weight, evec);
}

void launch_obs_icp(gingr_ctx *ctx, const gingr_model *m, const DevState *st, Cloud target, const int32_t *idx,
                    const int32_t *lm_mask, double *weight, double *evec) {
    hipLaunchKernelGGL(obs_points_kernel, dim3((unsigned)ceil_div(m->M, 256)), dim3(256), 0, ctx->stream, m->ref, m->mean,
                       m->M, st, (const double *)nullptr, target, idx, (const double *)nullptr, lm_mask, weight, evec, (int32_t *)nullptr);
}

void launch_obs_points(gingr_ctx *ctx, const gingr_model *m, const DevState *st, const double *obs_soa,
                       const double *weight_in, double *weight, double *evec, const int32_t *lm_mask, int32_t *zero_counts) {
    Cloud none{nullptr, nullptr, nullptr, 0};
    hipLaunchKernelGGL(obs_points_kernel, dim3((unsigned)ceil_div(m->M, 256)), dim3(256), 0, ctx->stream, m->ref, m->mean,
                       m->M, st, obs_soa, none, (const int32_t *)nullptr, weight_in, lm_mask, weight, evec, zero_counts);
}

void launch_landmarks(gingr_ctx *ctx, const gingr_model *m, const DevState *st, int32_t n_lm, const int32_t *lm_pid_local,
                      const double *lm_xyz, const double *lm_cov, double *G, double *rhs) {
    if (n_lm <= 0) return;
    hipLaunchKernelGGL(landmarks_kernel, dim3((unsigned)m->rp + 1), dim3(kLmChunk), 0, ctx->stream, m->Q0, m->ref, m->mean, m->M, (int)m->rp, st,
                       (int)n_lm, lm_pid_local, lm_xyz, lm_cov, G, rhs);
}

// (two workspaces from rp = 128 on: the two workgroups of posterior_logpdf_wide_kernel factor side by side)
int64_t posterior_work_doubles(int32_t rp) {
    const int64_t two_panel_workspaces = (int64_t)lds_solve_doubles(rp, kNB) * (rp >= 128 ? 2 : 1);
    const int64_t two_dense_systems = rp > 240 ? 2 * (round_up(rp, 64) + 64) * round_up(rp, 64) + 2 * (round_up(rp, 64) / 64) * 4096 + 6 * round_up(rp, 64) + 4 : 0;
    return std::max(two_panel_workspaces, two_dense_systems);
}

void launch_chol_block64(gingr_ctx *ctx, double *Aw, int64_t ld, int k, double *Linv, int32_t *flag) {
    const size_t lds = lds_solve_doubles(64, 64) * sizeof(double);
    // (the attribute is per function AND per device, and the group's worker threads launch concurrently: set whenever needed)
    if (lds > 48 * 1024)
        set_dynamic_lds(&chol_block64_kernel, (size_t)(lds));
    hipLaunchKernelGGL(chol_block64_kernel, dim3(1), dim3(256), lds, ctx->stream, Aw, ld, k, Linv, flag);
}

namespace {
// From rank 241 on the posterior mean takes the multi-workgroup blocked solve of the classic non-rigid CPD (classic_cpd.hip:
// dense_spd_solve3 -- 64-column panels, the trailing update spread over the chip, one launch per stage): one compute unit's matrix
// pipe is the floor of the one-workgroup kernel (380k cycles of MFMA at r = 512), the launches of this form cost ~6 us each.
// Aw: (Mp + 64) x Mp, lower triangle of I + G with the identity on the padding, the right-hand side in border row Mp.
__global__ __launch_bounds__(256) void solve_system_kernel(int r, int n, int64_t Mp, const double *__restrict__ G, const double *__restrict__ rhs,
                                                           double *__restrict__ Aw, int32_t *__restrict__ flag) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (c == 0 && row == 0) *flag = 0;
    if (c >= Mp) return;
    double v = 0.0;
    if (row < Mp) {
        if (c <= row) v = (row < r && c < r) ? G[row * n + c] + (row == c ? 1.0 : 0.0) : (row == c ? 1.0 : 0.0);
    } else if (row == Mp && c < r) {
        v = rhs[c];
    }
    Aw[row * Mp + c] = v;
}
__global__ __launch_bounds__(256) void solve_finish_kernel(int r, int rp, const double *__restrict__ W, const int32_t *__restrict__ flag,
                                                           double *__restrict__ a, DevState *__restrict__ st) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < rp; k += 256) {
        const double v = k < r ? W[k] : 0.0;
        a[k] = v;
        if (!finite_d(v)) bad = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (*flag == GINGR_ERR_NOT_SPD)
            st->err = GINGR_ERR_NOT_SPD;
        else if (bad)
            st->err = GINGR_ERR_NONFINITE;
    }
}
}  // namespace

namespace {
// The transition density above padded rank 384 on the same multi-workgroup solve, twice: N a = rhs (solve_system_kernel) and K w = Q0^T e + eps
// rhs with K = S_tot + eps N (this kernel builds its bordered system), then one workgroup forms |c|^2 = (w - a)^T (N w - rhs) and leaves
// the factor of K, its reciprocal diagonal and a in fx for posterior_logpdf_cached_kernel.
__global__ __launch_bounds__(256) void logpdf_system_kernel(int r, int n, int64_t Mp, const double *__restrict__ G, const double *__restrict__ Stot,
                                                            const double *__restrict__ rhs, const double *__restrict__ qte, double *__restrict__ Aw,
                                                            int32_t *__restrict__ flag) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (c == 0 && row == 0) *flag = 0;
    if (c >= Mp) return;
    double v = 0.0;
    if (row < Mp) {
        if (c <= row) {
            if (row < r && c < r) {
                v = __builtin_fma(GINGR_COEFF_NOISE, G[row * n + c], Stot[row * n + c]);
                if (row == c) v += GINGR_COEFF_NOISE;
            } else {
                v = row == c ? 1.0 : 0.0;
            }
        }
    } else if (row == Mp && c < r) {
        v = __builtin_fma(GINGR_COEFF_NOISE, rhs[c], qte[c]);
    }
    Aw[row * Mp + c] = v;
}
__global__ __launch_bounds__(512) void logpdf_finish_kernel(int r, int n, int64_t Mp, const double *__restrict__ G, const double *__restrict__ rhs,
                                                            const double *__restrict__ Wn, const double *__restrict__ Wk, const double *__restrict__ Lk,
                                                            const int32_t *__restrict__ flag_n, const int32_t *__restrict__ flag_k,
                                                            double *__restrict__ fx, double *__restrict__ out2) {
    __shared__ double w[512], hv[2][512], red[512];
    const int tid = threadIdx.x;
    for (int k = tid; k < n; k += 512) w[k] = k < r ? Wk[k] : 0.0;
    __syncthreads();
    for (int kk = tid & 255; kk < r; kk += 256) {  // (G w)_k: column k of the symmetric G, two halves of the row range, four chains each
        const int half = tid >> 8;
        const int j0 = half ? (r + 1) / 2 : 0, j1 = half ? r : (r + 1) / 2;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = j0;
        for (; j + 3 < j1; j += 4) {
            s0 = __builtin_fma(G[(int64_t)j * n + kk], w[j], s0);
            s1 = __builtin_fma(G[(int64_t)(j + 1) * n + kk], w[j + 1], s1);
            s2 = __builtin_fma(G[(int64_t)(j + 2) * n + kk], w[j + 2], s2);
            s3 = __builtin_fma(G[(int64_t)(j + 3) * n + kk], w[j + 3], s3);
        }
        for (; j < j1; ++j) s0 = __builtin_fma(G[(int64_t)j * n + kk], w[j], s0);
        hv[half][kk] = (s0 + s1) + (s2 + s3);
    }
    // the state-only part for the cached form: factor of K (rows of stride n), its reciprocal diagonal, a
    for (int64_t e = tid; e < (int64_t)n * n; e += 512) {
        const int64_t i = e / n, j = e - i * n;
        fx[e] = j <= i ? Lk[i * Mp + j] : 0.0;
    }
    for (int k = tid; k < n; k += 512) {
        fx[(int64_t)n * n + k] = k < r ? Wn[k] : 0.0;
        fx[(int64_t)n * n + n + k] = 1.0 / Lk[(int64_t)k * Mp + k];
    }
    __syncthreads();
    double part = 0.0;
    for (int k = tid; k < r; k += 512) {
        const double nw = (hv[0][k] + hv[1][k]) + w[k];  // (N w)_k
        part = __builtin_fma(w[k] - Wn[k], nw - rhs[k], part);
    }
    red[tid] = part;
    __syncthreads();
    for (int st2 = 256; st2 > 0; st2 >>= 1) {
        if (tid < st2) red[tid] += red[tid + st2];
        __syncthreads();
    }
    if (tid == 0) {
        const bool bad = *flag_n != 0 || *flag_k != 0;
        out2[0] = bad ? __builtin_nan("") : -0.5 * red[0] - 0.5 * (double)r * 1.8378770664093454836;  // log(2 pi)
        out2[1] = bad ? 1.0 : 0.0;
    }
}
}  // namespace

void launch_posterior_solve(gingr_ctx *ctx, int32_t r, int32_t rp, const double *G, const double *rhs, const double *zrand,
                            double *work, double *a, DevState *st) {
    TimerScope ts(ctx, 5);
    if (rp > 240 && !zrand) {  // (from rp = 256 on: 138 us against 145 there, 277 against 709 at rp = 512; a sampled proposal needs L^-T z as well: the one-workgroup kernel below)
        const int64_t Mp = round_up(rp, 64), nb = Mp / 64;
        double *Aw = work, *Linv = Aw + (Mp + 64) * Mp, *W = Linv + nb * 64 * 64;
        int32_t *flag = reinterpret_cast<int32_t *>(W + 3 * Mp);
        hipLaunchKernelGGL(solve_system_kernel, dim3((unsigned)ceil_div(Mp, 256), (unsigned)(Mp + 64)), dim3(256), 0, ctx->stream, (int)r, (int)rp, Mp, G,
                           rhs, Aw, flag);
        dense_spd_solve3(ctx, Aw, Mp, Linv, W, flag);
        hipLaunchKernelGGL(solve_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, (int)r, (int)rp, W, flag, a, st);
        return;
    }
    if (r <= 128) {
        const bool fast = rp <= 112;  // sixteen identity rows fit beside the bordered matrix
        const size_t lds = lds_solve_doubles(rp, fast ? 2 * kNB : kNB) * sizeof(double);
        auto go = [&](auto kern) {
            if (lds > 48 * 1024)  // per function and per device: set whenever needed
                set_dynamic_lds(kern, (size_t)(lds));
            hipLaunchKernelGGL(kern, dim3(1), dim3(kSolveThreads), lds, ctx->stream, (int)r, (int)rp, G, rhs, zrand, a, st, (double *)nullptr);
        };
        if (fast)
            go(posterior_solve_lds_kernel<0>);
        else
            go(posterior_solve_lds_kernel<1>);
        return;
    }
    // r > 128: the bordered matrix does not fit the LDS; super-panels of 64 (rp <= 256) or 32 columns on the global workspace
    // (posterior_work_doubles)
    auto gow = [&](auto kern, int sw) {
        const size_t lds = (size_t)(rp + kNB) * (sw + 1) * sizeof(double);
        set_dynamic_lds(kern, (size_t)(lds));
        hipLaunchKernelGGL(kern, dim3(1), dim3(kWideSolveThreads), lds, ctx->stream, (int)r, (int)rp, G, rhs, zrand, a, st, work);
    };
    if (rp <= 256)
        gow(posterior_solve_wide_kernel<64>, 64);
    else
        gow(posterior_solve_wide_kernel<32>, 32);
}

namespace {
// a = V ((V^T rhs) / (1 + lam / sigma2)): two r x r mat-vecs (V from L2), one workgroup; 16 lanes per output entry
__global__ __launch_bounds__(1024) void posterior_solve_eig_kernel(int r, int rp, const double *__restrict__ V,
                                                                   const double *__restrict__ lam, const double *__restrict__ sigma2,
                                                                   const double *__restrict__ rhs, double *__restrict__ a, DevState *st) {
    __shared__ double x[512], t[512];
    __shared__ int bad;
    const int tid = threadIdx.x, lane16 = tid & 15, grp = tid >> 4;
    if (tid == 0) bad = 0;
    for (int k = tid; k < r; k += 1024) x[k] = rhs[k];
    __syncthreads();
    const double inv_s2 = 1.0 / sigma2[0];
    for (int k = grp; k < r; k += 64) {  // t_k = (V[:, k] . rhs) / (1 + lam_k / sigma2)
        double s = 0.0;
        for (int i = lane16; i < r; i += 16) s = __builtin_fma(V[(int64_t)i * r + k], x[i], s);
        s = group16_sum(s);
        if (lane16 == 0) t[k] = s / (1.0 + lam[k] * inv_s2);
    }
    __syncthreads();
    for (int i = grp; i < rp; i += 64) {  // a_i = V[i, :] . t
        double s = 0.0;
        if (i < r)
            for (int k = lane16; k < r; k += 16) s = __builtin_fma(V[(int64_t)i * r + k], t[k], s);
        s = group16_sum(s);
        if (lane16 == 0) {
            a[i] = i < r ? s : 0.0;
            if (!finite_d(s)) bad = 1;
        }
    }
    __syncthreads();
    if (tid == 0 && bad) st->err = GINGR_ERR_NONFINITE;
}
}  // namespace

void launch_posterior_solve_eig(gingr_ctx *ctx, int32_t r, int32_t rp, const double *eigV, const double *eigL, const double *sigma2,
                                const double *rhs, double *a, DevState *st) {
    hipLaunchKernelGGL(posterior_solve_eig_kernel, dim3(1), dim3(1024), 0, ctx->stream, (int)r, (int)rp, eigV, eigL, sigma2, rhs, a, st);
}

int launch_posterior_logpdf(gingr_ctx *ctx, int32_t r, int32_t rp, const double *G, const double *rhs, const double *Stot,
                            const double *qte, double *fx, bool cached, double *work, double *out2, unsigned *sync, unsigned epoch,
                            bool keep_factor, double *nfac) {
    const size_t lds = lds_solve_doubles(rp, kNB) * sizeof(double);
    // the log-density kernels keep 14 KB of static LDS (mat-vec scratch) next to the bordered matrix: with rp = 128 the two exceed the
    // 160 KB of a compute unit, so ranks above 112 take the global-workspace variants (the plain solve fits up to rp = 128)
    const bool in_lds = rp <= 112;
    if (!cached && in_lds && fx && sync) {  // the two factorisations side by side
        const size_t lds2 = lds_solve_doubles(rp, 2 * kNB) * sizeof(double);  // (workgroup 0 carries the identity rows of the solve kernel)
        if (lds2 > 48 * 1024)  // per function and per device: set whenever needed
            set_dynamic_lds(&posterior_logpdf_split_kernel, (size_t)(lds2));
        hipLaunchKernelGGL(posterior_logpdf_split_kernel, dim3(2), dim3(kSolveThreads), lds2, ctx->stream, (int)r, (int)rp, G, rhs, Stot, qte, fx,
                           out2, sync, epoch, keep_factor ? 1 : 0, nfac);
        return GINGR_OK;
    }
    if (!cached && rp > 384 && fx) {  // above padded rank 384: both systems through the multi-workgroup blocked solve, one after the other
                                      // (r = 512: 743 us against 1 024 for the two workgroups below; r = 300: 460 against 366, hence the limit)
        const int64_t Mp = round_up(rp, 64), nb = Mp / 64, sys = (Mp + 64) * Mp + nb * 4096 + 3 * Mp + 2;
        double *Aw0 = work, *Li0 = Aw0 + (Mp + 64) * Mp, *W0 = Li0 + nb * 4096;
        double *Aw1 = work + sys, *Li1 = Aw1 + (Mp + 64) * Mp, *W1 = Li1 + nb * 4096;
        int32_t *f0 = reinterpret_cast<int32_t *>(W0 + 3 * Mp), *f1 = reinterpret_cast<int32_t *>(W1 + 3 * Mp);
        const dim3 grid((unsigned)ceil_div(Mp, 256), (unsigned)(Mp + 64));
        hipLaunchKernelGGL(solve_system_kernel, grid, dim3(256), 0, ctx->stream, (int)r, (int)rp, Mp, G, rhs, Aw0, f0);
        dense_spd_solve3(ctx, Aw0, Mp, Li0, W0, f0);
        hipLaunchKernelGGL(logpdf_system_kernel, grid, dim3(256), 0, ctx->stream, (int)r, (int)rp, Mp, G, Stot, rhs, qte, Aw1, f1);
        dense_spd_solve3(ctx, Aw1, Mp, Li1, W1, f1);
        hipLaunchKernelGGL(logpdf_finish_kernel, dim3(1), dim3(512), 0, ctx->stream, (int)r, (int)rp, Mp, G, rhs, W0, W1, Aw1, f0, f1, fx, out2);
        return GINGR_OK;
    }
    if (!cached && !in_lds && fx && sync) {  // ranks above 112: the two factorisations side by side on the global workspaces
        const int64_t stride = posterior_work_doubles(rp) / 2;
        auto gow = [&](auto kern, int sw) {
            const size_t ldsw = (size_t)(rp + kNB) * (sw + 1) * sizeof(double);
            set_dynamic_lds(kern, (size_t)(ldsw));
            hipLaunchKernelGGL(kern, dim3(2), dim3(kWideSolveThreads), ldsw, ctx->stream, (int)r, (int)rp, G, rhs, Stot, qte, fx, out2, sync, epoch,
                               work, stride);
        };
        if (rp <= 256)
            gow(posterior_logpdf_wide_kernel<64>, 64);
        else
            gow(posterior_logpdf_wide_kernel<32>, 32);
        return GINGR_OK;
    }
    if (cached) {  // fx holds what an earlier launch for this state left
        if (in_lds) {
            if (lds > 48 * 1024)  // per function and per device: set whenever needed
                set_dynamic_lds(&posterior_logpdf_cached_kernel<false>, (size_t)(lds));
            hipLaunchKernelGGL(posterior_logpdf_cached_kernel<false>, dim3(1), dim3(kSolveThreads), lds, ctx->stream, (int)r, (int)rp, G, Stot,
                               qte, fx, out2, (double *)nullptr);
        } else {
            hipLaunchKernelGGL(posterior_logpdf_cached_kernel<true>, dim3(1), dim3(kSolveThreads), 0, ctx->stream, (int)r, (int)rp, G, Stot, qte,
                               fx, out2, work);
        }
        return GINGR_OK;
    }
    if (in_lds) {
        if (lds > 48 * 1024)  // per function and per device: set whenever needed
            set_dynamic_lds(&posterior_logpdf_lds_kernel<false>, (size_t)(lds));
        hipLaunchKernelGGL(posterior_logpdf_lds_kernel<false>, dim3(1), dim3(kSolveThreads), lds, ctx->stream, (int)r, (int)rp, G, rhs, Stot, qte,
                           fx, out2, (double *)nullptr);
    } else {
        hipLaunchKernelGGL(posterior_logpdf_lds_kernel<true>, dim3(1), dim3(kSolveThreads), 0, ctx->stream, (int)r, (int)rp, G, rhs, Stot, qte, fx,
                           out2, work);
    }
    return GINGR_OK;
}

void launch_posterior_sample_cached(gingr_ctx *ctx, int32_t r, int32_t rp, const double *nfac, const double *a_mean, const double *zrand,
                                    double *a, DevState *st) {
    const size_t lds = lds_solve_doubles(rp, kNB) * sizeof(double);
    if (lds > 48 * 1024)
        set_dynamic_lds(&posterior_sample_cached_kernel, (size_t)(lds));
    hipLaunchKernelGGL(posterior_sample_cached_kernel, dim3(1), dim3(kSolveThreads), lds, ctx->stream, (int)r, (int)rp, nfac, a_mean, zrand, a, st);
}

namespace {
// Binv = (Q^T Q / eps + I)^-1 on the matrix pipe (round 6; until then one workgroup with a column of the inverse per thread: 1.4 ms
// at r = 100, 80 ms at r = 512; then a blocked factor with a column per thread behind it: 11 ms at r = 512).  The system with an
// identity below it goes through the multi-workgroup blocked Cholesky, which leaves L^-T in place of the identity, and the inverse is
// one product of that triangle with itself (classic_cpd.hip dense_spd_inverse).
__global__ __launch_bounds__(256) void binv_system_kernel(int r, int rp, int64_t Mp, const double *__restrict__ S, double *__restrict__ Aw,
                                                          int32_t *__restrict__ flag) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (c == 0 && row == 0) *flag = 0;
    if (c >= Mp) return;
    double v = 0.0;  // M = Q^T Q / eps + I, identity on the padding; below it the identity the inverse grows from
    if (row >= Mp)
        v = row - Mp == c ? 1.0 : 0.0;
    else if (c <= row)
        v = (row < r && c < r) ? S[row * rp + c] / GINGR_COEFF_NOISE + (row == c ? 1.0 : 0.0) : (row == c ? 1.0 : 0.0);
    Aw[row * Mp + c] = v;
}
// Binv [rp][rp]: the r x r block of C, zero on the padding
__global__ __launch_bounds__(256) void binv_store_kernel(int r, int rp, int64_t Mp, const double *__restrict__ C, double *__restrict__ Binv) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rp * rp) return;
    const int row = idx / rp, c = idx - row * rp;
    Binv[idx] = (row < r && c < r) ? C[(int64_t)row * Mp + c] : 0.0;
}
}  // namespace

int64_t binv_work_doubles(int32_t rp) {
    const int64_t Mp = round_up(rp, 64);
    return 3 * Mp * Mp + (Mp / 64) * 64 * 64;  // the system over the identity, the product, the inverses of the diagonal blocks
}

void launch_binv(gingr_ctx *ctx, int32_t r, int32_t rp, const double *S, double *work, double *Binv, int32_t *err_flag) {
    const int64_t Mp = round_up(rp, 64);
    double *Aw = work, *C = Aw + 2 * Mp * Mp, *Linv = C + Mp * Mp;
    hipLaunchKernelGGL(binv_system_kernel, dim3((unsigned)ceil_div(Mp, 256), (unsigned)(2 * Mp)), dim3(256), 0, ctx->stream, (int)r, (int)rp, Mp,
                       S, Aw, err_flag);
    dense_spd_inverse(ctx, Aw, Mp, Linv, C, err_flag);
    hipLaunchKernelGGL(binv_store_kernel, dim3((unsigned)ceil_div((int64_t)rp * rp, 256)), dim3(256), 0, ctx->stream, (int)r, (int)rp, Mp, C, Binv);
}


void launch_coeff_solve(gingr_ctx *ctx, int32_t r, int32_t rp, const double *Binv, const double *p, double *out) {
    hipLaunchKernelGGL(coeff_solve_kernel, dim3((unsigned)ceil_div(rp, 16)), dim3(256), 0, ctx->stream, (int)r, (int)rp, Binv,
                       p, out);
}


void launch_post_solve(gingr_ctx *ctx, const PostSolveArgs &a) {
    const size_t lds = (size_t)37 * a.rp * sizeof(double);
    const int nt = std::max<int>(kPostMinThreads, (int)round_up(a.rp, 64));
    if (lds > 48 * 1024)  // (per function AND per device: set whenever it is needed, never cached in a process-wide static)
        set_dynamic_lds(&post_solve_kernel, (size_t)(lds));
    hipLaunchKernelGGL(post_solve_kernel, dim3(1), dim3(nt), lds, ctx->stream, a);
}

void launch_post_matvecs(gingr_ctx *ctx, const gingr_model *m, const double *alpha, const double *a, double *zbuf) {
    hipLaunchKernelGGL(post_matvecs_kernel, dim3(PostVec::kZRows, (unsigned)(m->rp / 16)), dim3(256), 0, ctx->stream, (int)m->r, (int)m->rp,
                       m->mom, m->cmat, alpha, a, zbuf);
}

void launch_postvec(gingr_ctx *ctx, int32_t r, int32_t rp, const double *Binv, const double *moment_vectors, double *pvec) {
    hipLaunchKernelGGL(postvec_kernel, dim3((unsigned)PostVec::kRows), dim3(256), 0, ctx->stream, (int)r, (int)rp, Binv, moment_vectors, pvec);
}

void launch_small_gemm(gingr_ctx *ctx, int32_t r, int32_t rp, const double *A, const double *B, double scale, double *out) {
    const int64_t tiles = (int64_t)(rp / 16) * (rp / 16);
    hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)ceil_div(tiles, 4)), dim3(256), 0, ctx->stream, (int)r, (int)rp, A, B, scale, out);
}


void launch_state_init(gingr_ctx *ctx, DevState *st, const gingr_state_scalars *host_scalars_dev, double *zero_slot) {
    hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(64), 0, ctx->stream, st, host_scalars_dev, zero_slot);
}

void launch_interp_pack(gingr_ctx *ctx, const double *Qs, int32_t rp, const int32_t *inv_src, const int32_t *ids, const double *w,
                        const int32_t *perm_new, int64_t row_begin, int64_t M, double *Q0) {
    const int64_t total = 3 * M * rp;
    hipLaunchKernelGGL(interp_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, Qs, rp, inv_src, ids, w,
                       perm_new, row_begin, M, Q0);
}
void launch_pack_basis(gingr_ctx *ctx, const double *stage_colmajor, const double *variance_dev, int64_t M, int32_t r,
                       int32_t rp, const int32_t *perm, double *Q0) {
    const int64_t total = 3 * M * rp;
    hipLaunchKernelGGL(pack_basis_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, stage_colmajor,
                       variance_dev, 3 * M, r, rp, perm, Q0);
}

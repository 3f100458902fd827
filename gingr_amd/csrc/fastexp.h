// Table-driven float64 2^x for the all-pairs Gaussian kernels (device code, gfx950).
//
// The affinity kernels evaluate exp(-d2 / (2 sigma2)) once per (reference, target) pair; gfx950 has no float64
// transcendental unit, so the kernel is bound by how many f64 VALU instructions one exponential costs.  The
// library exp() is a ~25-instruction polynomial; this one needs 11 plus one 8-byte LDS read:
//
//   t  = d2 * c64,  c64 = -64 log2(e) / (2 sigma2)        (argument in units of 1/64 octave)
//   kf = rint(t);  f = t - kf  (exact, |f| <= 1/2);  k = (int)kf;  j = k & 63;  e = k >> 6
//   2^(t/64) = 2^e * T[j] * 2^(f/64),   T[j] = 2^(j/64) from a 64-entry LDS table
//   2^(f/64) = 1 + f*q(f),  q = degree-4 polynomial (Taylor of exp(f ln2/64); truncation 3.5e-17 relative)
//
// Accuracy: <= 1 ulp of the correctly rounded result of its (already rounded) argument; v_ldexp_f64 produces
// gradual underflow and flushes to +0 below 2^-1075 like Math.exp in the reference (CPD.scala:56).  NaN propagates.
// v_cvt_i32_f64 saturates, so arguments far below the underflow threshold need no clamp.
#pragma once

#include <hip/hip_runtime.h>

#define GINGR_EXP_TABLE 64

// ln2/64 powers over factorials: 2^(f/64) = sum_n (f ln2/64)^n / n!
#define GINGR_EXP_C1 1.08304246962491451e-02  /* ln2/64 */
#define GINGR_EXP_C2 5.86490495505616929e-05  /* (ln2/64)^2/2 */
#define GINGR_EXP_C3 2.11731371554647736e-07  /* (ln2/64)^3/6 */
#define GINGR_EXP_C4 5.73285168864040189e-10  /* (ln2/64)^4/24 */
#define GINGR_EXP_C5 1.24178437017169233e-12  /* (ln2/64)^5/120 */

__device__ static const double gingr_exp_table_rom[GINGR_EXP_TABLE] = {
#include "exp_table.inc"
};

// copy T[j] = 2^(j/64) into LDS; every thread of the block must call it, followed by __syncthreads()
__device__ __forceinline__ void fastexp_table_init(double *T) {
    for (int j = threadIdx.x + threadIdx.y * blockDim.x; j < GINGR_EXP_TABLE; j += blockDim.x * blockDim.y)
        T[j] = gingr_exp_table_rom[j];
}

// returns 2^(t/64) for t <= 0 (any finite t is handled; large positive t overflows to inf as expected)
__device__ __forceinline__ double fastexp2_64(double t, const double *T) {
    double kf = __builtin_rint(t);
    double f = t - kf;
    int k = (int)kf;  // v_cvt_i32_f64: saturating
    double q = __builtin_fma(f, GINGR_EXP_C5, GINGR_EXP_C4);
    q = __builtin_fma(f, q, GINGR_EXP_C3);
    q = __builtin_fma(f, q, GINGR_EXP_C2);
    q = __builtin_fma(f, q, GINGR_EXP_C1);
    double tj = T[k & (GINGR_EXP_TABLE - 1)];
    double fq = f * q;
    double r = __builtin_fma(tj, fq, tj);
    return __builtin_ldexp(r, k >> 6);
}

// c64 such that exp(-d2/(2 sigma2)) = 2^(d2*c64/64)
__device__ __forceinline__ double fastexp_scale_for_variance(double two_sigma2) {
    return -64.0 * 1.4426950408889634074 / two_sigma2;
}

// Table-driven float64 2^x for the all-pairs Gaussian kernels (device code, gfx950).
//
// The affinity kernels evaluate exp(-d2 / (2 sigma2)) once per (reference, target) pair; gfx950 has no float64
// transcendental unit, so the kernels are bound by how many full-rate f64 VALU instructions one exponential costs
// (every f64 op -- fma, add, mul, ldexp, rndne, cvt -- issues at 4 cycles per wave; profiles/r01_ubench_*).
// The library exp() is a ~25-instruction polynomial; this one is 8 instructions plus one 8-byte LDS read:
//
//   tm = fma(d2, c, MAGIC)          c = -2048 log2(e) / (2 sigma2): argument in units of 1/2048 octave;
//                                   MAGIC = 1.5 * 2^52, so tm = MAGIC + k with k = round(d2*c) in the low mantissa bits
//   kf = tm - MAGIC                 (exact)
//   f  = fma(d2, c, -kf)            = d2*c - k with ONE rounding, |f| <= 1/2
//   2^(d2*c/2048) = 2^e * T[j] * 2^(f/2048),  j = k & 2047,  e = k >> 11,  T[j] = 2^(j/2048) (16 KB LDS table)
//   2^(f/2048) = 1 + f*q(f),  q = c1 + f*(c2 + f*c3)   (Taylor of exp(f ln2/2048); truncation 3.4e-17 relative)
//   result = ldexp(fma(T[j], f*q, T[j]), e)
//
// DEG = 2 variant (the CPD passes): q = c1' + f*c2 with the cubic term economised into c1' (Chebyshev: f^3 ~ 3/16 f on
// |f| <= 1/2); one FMA less, relative error <= 2.02e-13 -- the same 1e-12 budget on K_ij as the guarded norm expansion,
// against the 1e-5 bar on vertex positions.  Zero / subnormal / NaN behaviour is identical (it comes from v_ldexp_f64).
//
// Accuracy (DEG = 3): <= 1 ulp of the correctly rounded result of its argument; v_ldexp_f64 gives gradual underflow and flushes
// to +0 below 2^-1075 like Math.exp in the reference (CPD.scala:56).  NaN propagates (tm = NaN -> f = NaN).
// Range: e is taken from mantissa bits 11..42 of tm, valid for |d2*c| < 2^42 (= 1.5e9 in units of d2/(2 sigma2));
// callers clamp d2 when the inputs could exceed that (fastexp_needs_clamp).
#pragma once

#include <hip/hip_runtime.h>

// Table size: TB = 11 (2048 entries, 16 KB).  An 8192-entry table (byte offset of T[k & 8191] by ONE SDWA shift, economised degree-2
// polynomial good to 3.2e-15) was used by the CPD passes in round 1 and lost to its LDS footprint once the floor form below got the
// one-instruction offset with 2048 entries; tools/gen_exp_table.py still prints its constants.
#define GINGR_EXP_TABLE 2048
#define GINGR_EXP_TABLE_LOG2 11
#define GINGR_EXP_MAGIC 6755399441055744.0    /* 1.5 * 2^52 */

template <int TB>
struct ExpTab;
template <>
struct ExpTab<11> {
    static constexpr double C1 = 3.38450771757785784e-04;    // ln2/2048
    static constexpr double C2 = 5.72744624517204032e-08;    // (ln2/2048)^2/2
    static constexpr double C3 = 6.46152867293236500e-12;    // (ln2/2048)^3/6
    static constexpr double C1_D2 = 3.384507729693224e-04;   // C1 + C3 * 3/16: minimax degree 2 on |f| <= 1/2 (2.0e-13)
};
__device__ static const double gingr_exp_table_rom[2048] = {
#include "exp_table.inc"
};
// copy T[j] = 2^(j/2^TB) into LDS; every thread of the block must call it, followed by __syncthreads()
template <int TB = 11>
__device__ __forceinline__ void fastexp_table_init(double *T) {
    static_assert(TB == 11, "one table size");
    const double *rom = gingr_exp_table_rom;
    for (int j = threadIdx.x + threadIdx.y * blockDim.x; j < (1 << TB); j += blockDim.x * blockDim.y) T[j] = rom[j];
}

// 2^((tm - MAGIC + f) / 2^TB) given tm = MAGIC + k and the reduced argument f
template <int DEG = 3, int TB = 11>
__device__ __forceinline__ double fastexp2_core(double tm, double f, const double *T) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, tm);
    const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
    const int e = (int)__builtin_amdgcn_alignbit(hi, lo, TB);  // bits TB..TB+31: floor(k / 2^TB)
    double q;
    if (DEG == 3) {
        q = __builtin_fma(f, ExpTab<TB>::C3, ExpTab<TB>::C2);
        q = __builtin_fma(f, q, ExpTab<TB>::C1);
    } else {
        q = __builtin_fma(f, ExpTab<TB>::C2, ExpTab<TB>::C1_D2);
    }
    const double tj = T[lo & ((1u << TB) - 1)];
    const double fq = f * q;
    const double r = __builtin_fma(tj, fq, tj);
    return __builtin_ldexp(r, e);
}

// returns 2^(d2*c/2^TB); requires |d2*c| < 2^42 (see fastexp_needs_clamp)
template <int DEG = 3, int TB = 11>
__device__ __forceinline__ double fastexp2_scaled(double d2, double c, const double *T) {
    const double tm = __builtin_fma(d2, c, GINGR_EXP_MAGIC);
    const double kf = tm - GINGR_EXP_MAGIC;
    const double f = __builtin_fma(d2, c, -kf);
    return fastexp2_core<DEG, TB>(tm, f, T);
}

// ---------------------------------------------------------------------------------------------------------------------
// Floor form (the two CPD passes): two f64 instructions for the range reduction instead of three.
//
// With the wave's float64 rounding mode set to round-toward-minus-infinity (fastexp_round_down()), tm = u + MAGIC holds
// floor(u) in its low mantissa bits and v_fract_f64 gives f = u - floor(u) in [0, 1) EXACTLY -- one add and
// one fract instead of add, subtract, subtract.  The polynomial is the economised degree-2 one of ExpTab<11> re-centred on
// [0, 1):  2^((j + f)/2048) = T'[j] (1 + f (Q0 + Q1 f)),  T'[j] = T[j] * S,  S = 2^(1/4096) (1 - C1_D2/2 + C2/4); same
// 2.02e-13 bound (tools/gen_exp_table.py floor).  Everything a wave computes while the mode is set rounds down: sums of n
// terms carry a bias of at most n 2^-53 relative (1e-13 over a 512-point chunk), inside the same budget.  Zero / subnormal /
// NaN behaviour is unchanged (v_ldexp_f64; v_fract_f64 of NaN is NaN, of +-inf NaN as well -- callers clamp beforehand).
struct ExpFloor11 {
    static constexpr double Q0 = 3.384507681227659e-04;
    static constexpr double Q1 = 5.728415556485551e-08;
    static constexpr double S = 0x1.000000000038dp+0;
};

// MODE.FP_ROUND[3:2] (float64 / float16 rounding) <- 2 (toward -inf) resp. 0 (nearest even).  Inline assembly on purpose: the
// compiler's mode-register pass (SIModeRegister) tracks s_setreg it can see and restores the default rounding in front of the
// next float64 instruction (observed: `s_setreg_imm32_b32 hwreg(HW_REG_MODE, 3, 1), 0` right behind the builtin's setreg).
__device__ __forceinline__ void fastexp_round_down() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\ts_nop 3" ::: "memory"); }
__device__ __forceinline__ void fastexp_round_nearest() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\ts_nop 3" ::: "memory"); }

// Table of the floor form, T'[j] = 2^(j/2048) * S, multiplied out at build time (exp_floor_table.inc).  A 256-thread workgroup copies it
// to LDS in two halves: fetch256 only ISSUES the four 16-byte loads of a thread -- the caller requests its other prologue data
// (owned points, boxes) behind them, so that everything is one memory round trip -- and park256 writes them to LDS; every thread of
// the block calls both, then __syncthreads().  (Until round 4 this was a loop over blockDim with the multiplication inside: eight
// DEPENDENT load -> wait -> multiply -> store rounds, 4-5 us of the 6.8 us prologue of a one-round launch:
// profiles/r05_shard_pair_loop_stamps.txt.)
__device__ static const double gingr_exp_floor_table_rom[2048] __attribute__((aligned(16))) = {
#include "exp_floor_table.inc"
};
typedef double fastexp_v2f64 __attribute__((ext_vector_type(2)));
struct FloorTableRegs {
    fastexp_v2f64 v0, v1, v2, v3;
};
__device__ __forceinline__ void fastexp_floor_table_fetch256(FloorTableRegs &r) {
    const fastexp_v2f64 *rom = reinterpret_cast<const fastexp_v2f64 *>(gingr_exp_floor_table_rom) + threadIdx.x;
    r.v0 = rom[0];
    r.v1 = rom[256];
    r.v2 = rom[512];
    r.v3 = rom[768];
}
__device__ __forceinline__ void fastexp_floor_table_park256(double *T, const FloorTableRegs &r) {
    fastexp_v2f64 *t2 = reinterpret_cast<fastexp_v2f64 *>(T) + threadIdx.x;
    t2[0] = r.v0;
    t2[256] = r.v1;
    t2[512] = r.v2;
    t2[768] = r.v3;
}

// The magic constant of the floor form is 1.5 * 2^49 (ulp 1/8): tm = u + MAGIC8 then holds floor(8u) = 8 floor(u) + s in its low
// mantissa bits, s in [0, 7] being three fraction bits nobody needs -- f comes from v_fract_f64, not from tm -- so the BYTE
// offset of T'[floor(u) & 2047] is (lo & 0x3FF8): one v_and instead of v_and + v_lshl, and the exponent is bits 14..45.
#define GINGR_EXP_MAGIC8 844424930131968.0    /* 1.5 * 2^49 */

// 2^((k + f)/2048) with tm = MAGIC8 + k + s/8 (k = floor of the argument, any integer offset already folded into MAGIC8) and f in
// [0, 1); valid for |k| < 2^45
__device__ __forceinline__ double fastexp2_floor_core(double tm, double f, const double *T) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, tm);
    const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
    const int e = (int)__builtin_amdgcn_alignbit(hi, lo, 14);
    const double q = __builtin_fma(f, ExpFloor11::Q1, ExpFloor11::Q0);
    const double tj = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(T) + (lo & 0x3FF8u));
    const double fq = f * q;
    const double r = __builtin_fma(tj, fq, tj);
    return __builtin_ldexp(r, e);
}

// 2^(d2*c/2048), difference form (requires round-down mode).  The product is rounded before the reduction: a relative error of
// K of |ln K| 2^-53 <= 8e-14, below the polynomial's.
__device__ __forceinline__ double fastexp2_floor_scaled(double d2, double c, const double *T) {
    const double u = d2 * c;
    return fastexp2_floor_core(u + GINGR_EXP_MAGIC8, __builtin_amdgcn_fract(u), T);
}

// c such that exp(-d2 / two_sigma2) = 2^(d2*c/2^TB)
template <int TB = 11>
__device__ __forceinline__ double fastexp_scale_for_variance(double two_sigma2) {
    return -(double)(1 << TB) * 1.4426950408889634074 / two_sigma2;
}

// largest d2 for which d2*c stays representable; beyond it the result is +0 anyway (2^-1100 underflows)
template <int TB = 11>
__device__ __forceinline__ double fastexp_d2_limit(double c) {
    return -1100.0 * (double)(1 << TB) / c;
}

// true when an upper bound on d2 could push |d2*c| past 2^41 (a factor 2 of margin to the 2^42 limit)
__device__ __forceinline__ bool fastexp_needs_clamp(double d2_bound, double c) {
    return !(d2_bound * (-c) < 2199023255552.0);
}

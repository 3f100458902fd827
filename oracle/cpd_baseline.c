/*
 * CPU BASELINE (plain C, float64, OpenMP + SIMD) of the two all-pairs passes of one CPD update -- the "CPU baseline
 * (C restatement of the reference algorithm), -O3 -march=native, OpenMP over all host cores" that BASELINE.md / SURVEY.md
 * section 8d promise, timed by bench.py's cpu_baseline leg NEXT TO the GPU number.  TEST / MEASUREMENT INFRASTRUCTURE ONLY:
 * the product library never links or calls this file.  PARITY UNPINNED like the rest of oracle/ (no reference golden vectors).
 *
 * Difference to oracle/cpd_oracle.c (the strict CHECKER: -O2 -ffp-contract=off, scalar glibc exp, reference operation order):
 * this file is allowed to be fast -- structure-of-arrays inner loops under `#pragma omp simd`, FMA contraction, and a
 * vectorisable exponential (Cody-Waite reduction x = k ln2 + r, degree-11 Taylor polynomial in r, 2^k through the exponent
 * bits; results below 2^-1022 flush to zero).  baseline_exp_max_rel_error() lets the tests check it against libm (<= 1e-12 is
 * required, ~2e-16 measured).  Same formulas as the checker (G/api/registration/config/CPD.scala:54-75,133-147):
 *   K_ij = exp(-|x_j - y_i|^2 / (2 sigma2)); den_j = sum_i K_ij + c; P1_i = sum_j K_ij / den_j; PX_i = sum_j K_ij x_j / den_j.
 * Built at run time ON the box that times it (bench.py), because -march=native code must not travel between hosts.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int baseline_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void baseline_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

#pragma omp declare simd notinbranch
static inline double vexp(double x) {
    /* valid for x <= 0 (the only arguments the passes produce); flushes to 0 below -708 */
    const double LOG2E = 1.4426950408889634074, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double MAGIC = 6755399441055744.0; /* 1.5 * 2^52 */
    const double xc = fmax(x, -708.0);
    double t = xc * LOG2E + MAGIC;
    double k = t - MAGIC;
    double r = (xc - k * LN2_HI) - k * LN2_LO;
    double p = 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    union {
        double d;
        int64_t i;
    } ut, um, us;
    ut.d = t;
    um.d = MAGIC;
    us.i = (ut.i - um.i + 1023) << 52; /* 2^k, k >= -1022 because x >= -708 */
    return p * us.d * (double)(x >= -708.0);
}

double baseline_exp_max_rel_error(int64_t n, double lo) {
    double worst = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double x = lo * (double)i / (double)(n - 1);
        const double a = vexp(x), b = exp(x);
        const double e = b > 0 ? fabs(a - b) / b : fabs(a);
        if (e > worst) worst = e;
    }
    return worst;
}

/* points: structure of arrays (x[n], y[n], z[n]) */
void baseline_cpd_colsum(int64_t M, const double *fx, const double *fy, const double *fz, int64_t N, const double *tx, const double *ty,
                         const double *tz, double sigma2, double *den_partial) {
    const double c = -1.0 / (2.0 * sigma2);
    enum { JB = 64 };
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t jb = 0; jb < N; jb += JB) {
        const int64_t nj = N - jb < JB ? N - jb : JB;
        double acc[JB];
        for (int64_t j = 0; j < nj; ++j) acc[j] = 0.0;
        for (int64_t i = 0; i < M; ++i) {
            const double yx = fx[i], yy = fy[i], yz = fz[i];
#pragma omp simd
            for (int64_t j = 0; j < nj; ++j) {
                const double dx = tx[jb + j] - yx, dy = ty[jb + j] - yy, dz = tz[jb + j] - yz;
                acc[j] += vexp(c * (dx * dx + dy * dy + dz * dz));
            }
        }
        for (int64_t j = 0; j < nj; ++j) den_partial[jb + j] = acc[j];
    }
}

void baseline_cpd_rowstats(int64_t M, const double *fx, const double *fy, const double *fz, int64_t N, const double *tx,
                           const double *ty, const double *tz, double sigma2, const double *inv_den, double *P1, double *PXx,
                           double *PXy, double *PXz) {
    const double c = -1.0 / (2.0 * sigma2);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < M; ++i) {
        const double yx = fx[i], yy = fy[i], yz = fz[i];
        double p1 = 0.0, px = 0.0, py = 0.0, pz = 0.0;
#pragma omp simd reduction(+ : p1, px, py, pz)
        for (int64_t j = 0; j < N; ++j) {
            const double dx = tx[j] - yx, dy = ty[j] - yy, dz = tz[j] - yz;
            const double p = vexp(c * (dx * dx + dy * dy + dz * dz)) * inv_den[j];
            p1 += p;
            px += p * tx[j];
            py += p * ty[j];
            pz += p * tz[j];
        }
        P1[i] = p1;
        PXx[i] = px;
        PXy[i] = py;
        PXz[i] = pz;
    }
}

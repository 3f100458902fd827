/*
 * CPU oracle (plain C, float64) for the all-pairs loops of GiNGR's update path.
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / reported baseline.  The product
 * library (gingr_amd/csrc) never links or calls this file.
 *
 * PARITY UNPINNED: the reference (unibas-gravis/GiNGR, Scala) has no test or
 * golden vector for this path and cannot be run here (no JVM).  The functions
 * below restate reference source lines that ARE in /root/reference; they are
 * validated against the dense numpy restatement in oracle/gingr_oracle.py and
 * against closed-form known answers (tests/test_oracle_kat.py).
 *
 * G/ = src/main/scala/gingr/ in the reference.
 *
 * All point arrays are interleaved x,y,z (PointSequenceConverter.scala:54-59).
 * The streaming form never materialises P (M x N); it evaluates K_ij twice,
 * which is arithmetically what the reference's dense formulas compute:
 *   K_ij  = exp(-||x_j - y_i||^2 / (2 sigma2))            G/api/registration/config/CPD.scala:55-57,64-68
 *   c     = w/(1-w) * (2 pi sigma2)^(3/2) * (M/N)         CPD.scala:69-70
 *   den_j = sum_i K_ij + c                                CPD.scala:71-72
 *   P_ij  = K_ij / den_j                                  CPD.scala:74
 *   P1_i  = sum_j P_ij ; PX_i = sum_j P_ij x_j            CPD.scala:36, 144
 *   Pt1_j = sum_i P_ij ; Np = sum_i P1_i                  CPD.scala:139-140
 *   sigma2' = (xPx - 2 trPXY + yPy) / (3 Np)              CPD.scala:142-145
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_D 3.14159265358979323846

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline double norm2_3(const double *a, const double *b) {
    /* (a - b).norm2: x, y, z squared differences summed in order */
    double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz;
}

/* Column sums of K over a row range [i0, i1) -- the per-shard partial of den (no +c). */
void oracle_cpd_colsum_partial(int64_t i0, int64_t i1, int64_t N, const double *fit, const double *target,
                               double sigma2, double *den_partial) {
    const double inv = 1.0 / (2.0 * sigma2);
    (void)inv;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < N; ++j) {
        const double *x = target + 3 * j;
        double acc = 0.0;
        for (int64_t i = i0; i < i1; ++i) acc += exp(-norm2_3(x, fit + 3 * i) / (2.0 * sigma2));
        den_partial[j] = acc;
    }
}

double oracle_cpd_outlier_constant(int64_t M, int64_t N, double sigma2, double w) {
    return w / (1 - w) * pow(2.0 * PI_D * sigma2, 3.0 / 2.0) * ((double)M / (double)N);
}

/* Row statistics over a row range given the full den (already including c). */
void oracle_cpd_rowstats_partial(int64_t i0, int64_t i1, int64_t N, const double *fit, const double *target,
                                 double sigma2, const double *den, double *P1, double *PX) {
#pragma omp parallel for schedule(static)
    for (int64_t i = i0; i < i1; ++i) {
        const double *y = fit + 3 * i;
        double p1 = 0.0, px = 0.0, py = 0.0, pz = 0.0;
        for (int64_t j = 0; j < N; ++j) {
            const double *x = target + 3 * j;
            double p = exp(-norm2_3(x, y) / (2.0 * sigma2)) / den[j];
            p1 += p;
            px += p * x[0];
            py += p * x[1];
            pz += p * x[2];
        }
        P1[i] = p1;
        PX[3 * i + 0] = px;
        PX[3 * i + 1] = py;
        PX[3 * i + 2] = pz;
    }
}

/*
 * Full single-shard statistics.  scalars_out = { Np, xPx, trPXY, yPy, sigma2_next, c }.
 * Pt1_j = (den_j - c) / den_j is algebraically sum_i K_ij / den_j.
 */
void oracle_cpd_stats(int64_t M, int64_t N, const double *fit, const double *target, double sigma2, double w,
                      double *den, double *P1, double *PX, double *Pt1, double *scalars_out) {
    const double c = oracle_cpd_outlier_constant(M, N, sigma2, w);
    oracle_cpd_colsum_partial(0, M, N, fit, target, sigma2, den);
    for (int64_t j = 0; j < N; ++j) {
        double colsum = den[j];
        den[j] = colsum + c;
        Pt1[j] = colsum / den[j];
    }
    oracle_cpd_rowstats_partial(0, M, N, fit, target, sigma2, den, P1, PX);
    double Np = 0.0, xPx = 0.0, trPXY = 0.0, yPy = 0.0;
    for (int64_t i = 0; i < M; ++i) Np += P1[i];
    for (int64_t j = 0; j < N; ++j) {
        const double *x = target + 3 * j;
        xPx += Pt1[j] * (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    }
    for (int64_t i = 0; i < M; ++i) {
        const double *y = fit + 3 * i;
        yPy += P1[i] * (y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
        trPXY += y[0] * PX[3 * i] + y[1] * PX[3 * i + 1] + y[2] * PX[3 * i + 2];
    }
    scalars_out[0] = Np;
    scalars_out[1] = xPx;
    scalars_out[2] = trPXY;
    scalars_out[3] = yPy;
    scalars_out[4] = (xPx - 2 * trPXY + yPy) / (Np * 3.0);
    scalars_out[5] = c;
}

/* sum_ij ||x_j - y_i||^2 / (3 N M)                        CPD.scala:81-90 */
double oracle_cpd_initial_sigma2(int64_t M, int64_t N, const double *ref, const double *target) {
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t i = 0; i < M; ++i) {
        double acc = 0.0;
        for (int64_t j = 0; j < N; ++j) acc += norm2_3(target + 3 * j, ref + 3 * i);
        total += acc;
    }
    return total / (3.0 * (double)N * (double)M);
}

/*
 * Point-cloud closest point: idx_i = argmin_j ||x_j - y_i||, exact f64, lowest j on ties.
 * G/api/registration/utils/ClosestPointRegistrator.scala:133-148 (scalismo findClosestPoint is an exact
 * KD-tree nearest neighbour; its tie order is unspecified, the build defines lowest index).
 * Returns the mean Euclidean distance (the `distance / numberOfPoints` of :143,146).
 */
double oracle_nn(int64_t M, int64_t N, const double *query, const double *target, int32_t *idx, double *d2) {
    double dist = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dist)
    for (int64_t i = 0; i < M; ++i) {
        const double *y = query + 3 * i;
        double best = INFINITY;
        int32_t bj = -1;
        for (int64_t j = 0; j < N; ++j) {
            double v = norm2_3(target + 3 * j, y);
            if (v < best) {
                best = v;
                bj = (int32_t)j;
            }
        }
        idx[i] = bj;
        d2[i] = best;
        dist += sqrt(best);
    }
    return M > 0 ? dist / (double)M : 0.0;
}

/*
 * Gaussian-kernel covariance block, out[i*nb + j] = s * exp(-||a_i - b_j||^2 / sigma^2)
 * G/api/gpmm/GPMMHelper.scala:99-102 (scalismo GaussianKernel(sigma) * scaling; the x I3 of DiagonalKernel is implicit).
 */
void oracle_gauss_block(int64_t na, int64_t nb, const double *A, const double *B, double sigma, double scaling,
                        double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < nb; ++j)
            out[i * nb + j] = scaling * exp(-norm2_3(A + 3 * i, B + 3 * j) / (sigma * sigma));
}

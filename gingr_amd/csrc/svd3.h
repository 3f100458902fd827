// 3x3 singular value decomposition, polar rotation and the Euler conventions shared by the Umeyama step of the update (gp.hip),
// the classic rigid CPD (classic_cpd.hip) and the rigid ICP (rigid_icp.hip)
#pragma once
#include <hip/hip_runtime.h>

// The float64 sine, cosine, arctangent and arcsine are each ~150-250 instructions when inlined, and the Euler round trip below calls
// them ten times from ONE thread of a kernel that starts with a cold instruction cache behind the long all-pairs kernels (the
// post-solve kernel was 33 KB of straight-line code executed once per launch).  On the device they are real calls to one copy
// each: less code to fetch, the same values.
#if defined(__HIP_DEVICE_COMPILE__)
struct GingrSinCos {
    double s, c;
};
__device__ __attribute__((noinline)) inline GingrSinCos gingr_sincos(double x) {
    GingrSinCos r;
    r.s = sin(x);
    r.c = cos(x);
    return r;
}
__device__ __attribute__((noinline)) inline double gingr_atan2(double y, double x) { return atan2(y, x); }
__device__ __attribute__((noinline)) inline double gingr_asin(double x) { return asin(x); }
#define GINGR_SINCOS(x, sv, cv)                \
    const GingrSinCos sc_##sv = gingr_sincos(x); \
    const double sv = sc_##sv.s, cv = sc_##sv.c
#define GINGR_ATAN2(y, x) gingr_atan2((y), (x))
#define GINGR_ASIN(x) gingr_asin(x)
#else
#define GINGR_SINCOS(x, sv, cv) const double sv = sin(x), cv = cos(x)
#define GINGR_ATAN2(y, x) atan2((y), (x))
#define GINGR_ASIN(x) asin(x)
#endif

// ------------------------------------------------------------------------------------------ rotation conventions
// scalismo RotationSpace3D: R = Rz(phi) Ry(theta) Rx(psi) ("x-convention"), and Slabaugh's inverse.
__host__ __device__ inline void euler_to_rot(const double e[3], double R[9]) {
    GINGR_SINCOS(e[0], sphi, cphi);
    GINGR_SINCOS(e[1], sth, cth);
    GINGR_SINCOS(e[2], spsi, cpsi);
    R[0] = cth * cphi;
    R[1] = spsi * sth * cphi - cpsi * sphi;
    R[2] = spsi * sphi + cpsi * sth * cphi;
    R[3] = cth * sphi;
    R[4] = cpsi * cphi + spsi * sth * sphi;
    R[5] = cpsi * sth * sphi - spsi * cphi;
    R[6] = -sth;
    R[7] = spsi * cth;
    R[8] = cpsi * cth;
}

__host__ __device__ inline void rot_to_euler(const double R[9], double e[3]) {
    if (fabs(fabs(R[6]) - 1) > 0.0001) {
        const double theta = GINGR_ASIN(-R[6]);
        GINGR_SINCOS(theta, st_unused, ct);
        (void)st_unused;
        e[2] = GINGR_ATAN2(R[7] / ct, R[8] / ct);
        e[0] = GINGR_ATAN2(R[3] / ct, R[0] / ct);
        e[1] = theta;
    } else {
        e[0] = 0.0;  // gimbal lock: phi := 0
        if (fabs(R[6] + 1) < 0.0001) {
            e[1] = 3.14159265358979323846 / 2.0;
            e[2] = e[0] + GINGR_ATAN2(R[1], R[2]);
        } else {
            e[1] = -3.14159265358979323846 / 2.0;
            e[2] = -e[0] + GINGR_ATAN2(-R[1], -R[2]);
        }
    }
}


#if defined(__HIP_DEVICE_COMPILE__)
// The same two conversions for a caller whose WHOLE wave executes them with the same arguments (the pose step of the post-solve
// kernel): the independent transcendental calls run in different lanes at the same time -- three sincos of euler_to_rot in lanes
// 0..2, the two arctangents of rot_to_euler in lanes 0..1 -- and are broadcast; 4 call times instead of 7 on the one-wave chain.
// Same functions on the same arguments, so the same bits as the scalar forms above.
__device__ inline void euler_to_rot_wave(const double e[3], double R[9]) {
    const int l = (int)(__lane_id() % 3u);
    const GingrSinCos sc = gingr_sincos(l == 0 ? e[0] : (l == 1 ? e[1] : e[2]));
    const double sphi = __shfl(sc.s, 0), cphi = __shfl(sc.c, 0), sth = __shfl(sc.s, 1), cth = __shfl(sc.c, 1), spsi = __shfl(sc.s, 2),
                 cpsi = __shfl(sc.c, 2);
    R[0] = cth * cphi;
    R[1] = spsi * sth * cphi - cpsi * sphi;
    R[2] = spsi * sphi + cpsi * sth * cphi;
    R[3] = cth * sphi;
    R[4] = cpsi * cphi + spsi * sth * sphi;
    R[5] = cpsi * sth * sphi - spsi * cphi;
    R[6] = -sth;
    R[7] = spsi * cth;
    R[8] = cpsi * cth;
}

__device__ inline void rot_to_euler_wave(const double R[9], double e[3]) {
    if (fabs(fabs(R[6]) - 1) > 0.0001) {  // (the same branch in every lane: same R)
        const double theta = GINGR_ASIN(-R[6]);
        GINGR_SINCOS(theta, st_unused, ct);
        (void)st_unused;
        const bool odd = (__lane_id() & 1u) != 0;  // even lanes: psi = atan2(R7 / ct, R8 / ct); odd lanes: phi = atan2(R3 / ct, R0 / ct)
        const double a = GINGR_ATAN2((odd ? R[3] : R[7]) / ct, (odd ? R[0] : R[8]) / ct);
        e[2] = __shfl(a, 0);
        e[0] = __shfl(a, 1);
        e[1] = theta;
    } else {
        rot_to_euler(R, e);  // gimbal lock: one arctangent
    }
}
#else  // (the host pass of a kernel that names them)
__host__ __device__ inline void euler_to_rot_wave(const double e[3], double R[9]) { euler_to_rot(e, R); }
__host__ __device__ inline void rot_to_euler_wave(const double R[9], double e[3]) { rot_to_euler(R, e); }
#endif

// one-sided Jacobi SVD of a 3x3 matrix: A = U diag(s) V^T, s descending.  Out of line: it is the rarely taken fallback of
// polar3_rotation, and its dynamically indexed arrays would otherwise put the whole calling kernel on scratch memory.
__device__ __attribute__((noinline)) inline void svd3(const double Ain[9], double U[9], double s[3], double V[9]) {
    double A[9];
    for (int q = 0; q < 9; ++q) {
        A[q] = Ain[q];
        V[q] = (q % 4 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += A[i * 3 + p] * A[i * 3 + p];
                    beta += A[i * 3 + q] * A[i * 3 + q];
                    gamma += A[i * 3 + p] * A[i * 3 + q];
                }
                const double lim = 1e-17 * sqrt(alpha * beta);
                if (fabs(gamma) <= lim || gamma == 0.0) continue;
                off = fmax(off, fabs(gamma) / sqrt(alpha * beta));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double ap = A[i * 3 + p], aq = A[i * 3 + q];
                    A[i * 3 + p] = c * ap - sn * aq;
                    A[i * 3 + q] = sn * ap + c * aq;
                    const double vp = V[i * 3 + p], vq = V[i * 3 + q];
                    V[i * 3 + p] = c * vp - sn * vq;
                    V[i * 3 + q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-16) break;
    }
    double nrm[3];
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (nrm[ord[b]] > nrm[ord[a]]) {
                const int t = ord[a];
                ord[a] = ord[b];
                ord[b] = t;
            }
    double Vs[9];
    for (int j = 0; j < 3; ++j) {
        const int o = ord[j];
        s[j] = nrm[o];
        for (int i = 0; i < 3; ++i) {
            U[i * 3 + j] = nrm[o] > 0 ? A[i * 3 + o] / nrm[o] : 0.0;
            Vs[i * 3 + j] = V[i * 3 + o];
        }
    }
    for (int q = 0; q < 9; ++q) V[q] = Vs[q];
    // complete a rank-deficient U to an orthonormal basis (third column = cross product)
    if (!(s[2] > 1e-300 * s[0])) {
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}


// Orthogonal polar factor R of a 3x3 matrix with positive determinant (A = R H, H symmetric positive definite) by the scaled
// Newton iteration X <- (g X + X^-T / g) / 2, g = (|X^-1|_F / |X|_F)^(1/2) (Higham).  For det A > 0 this IS the rotation
// U diag(1, 1, det(U V^T)) V^T = U V^T that Kabsch / Umeyama build from the SVD, and trace(R^T A) is the sum of the singular values --
// so the Umeyama step needs no SVD in that case: ~8 short iterations of 3x3 cofactor algebra instead of Jacobi sweeps full of
// dependent float64 square roots and divisions run by ONE thread (the latency of that thread is on the critical path of every
// iteration: everything after the posterior solve is replicated O(r^2) work).  Returns false -- the caller falls back to svd3 --
// when det A <= 0, A is not finite, or the iteration does not settle (ill-conditioned A).
__device__ inline bool polar3_rotation(const double A[9], double R[9], double *trace_RtA) {
    const double det0 = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (!(det0 > 0.0) || !(det0 < 1.79769313486231570815e308)) return false;
    double X[9];
    for (int q = 0; q < 9; ++q) X[q] = A[q];
    bool done = false;
    for (int it = 0; it < 40 && !done; ++it) {
        // cofactor matrix C (X^-T = C / det)
        double C[9];
        C[0] = X[4] * X[8] - X[5] * X[7];
        C[1] = X[5] * X[6] - X[3] * X[8];
        C[2] = X[3] * X[7] - X[4] * X[6];
        C[3] = X[2] * X[7] - X[1] * X[8];
        C[4] = X[0] * X[8] - X[2] * X[6];
        C[5] = X[1] * X[6] - X[0] * X[7];
        C[6] = X[1] * X[5] - X[2] * X[4];
        C[7] = X[2] * X[3] - X[0] * X[5];
        C[8] = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
        if (!(det > 0.0)) return false;
        double nx = 0.0, nc = 0.0;
        for (int q = 0; q < 9; ++q) {
            nx += X[q] * X[q];
            nc += C[q] * C[q];
        }
        const double idet = 1.0 / det;
        // g^2 = |X^-1|_F / |X|_F = sqrt(nc) / (det sqrt(nx))
        const double g2 = sqrt(nc / nx) * idet;
        const double g = sqrt(g2);
        const double a = 0.5 * g, b = 0.5 * idet / g;
        double diff = 0.0;
        for (int q = 0; q < 9; ++q) {
            const double xn = a * X[q] + b * C[q];
            diff = fmax(diff, fabs(xn - X[q]));
            X[q] = xn;
        }
        if (!(diff == diff)) return false;
        done = diff < 4e-16;  // entries of a rotation are <= 1 in magnitude: absolute = relative accuracy
    }
    if (!done) return false;
    double tr = 0.0;
    for (int q = 0; q < 9; ++q) {
        R[q] = X[q];
        tr += X[q] * A[q];
    }
    *trace_RtA = tr;
    return true;
}

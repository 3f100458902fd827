// 3x3 singular value decomposition shared by the Umeyama step of the update (gp.hip) and the classic rigid CPD (classic_cpd.hip)
#pragma once
#include <hip/hip_runtime.h>

// one-sided Jacobi SVD of a 3x3 matrix: A = U diag(s) V^T, s descending
__device__ inline void svd3(const double Ain[9], double U[9], double s[3], double V[9]) {
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


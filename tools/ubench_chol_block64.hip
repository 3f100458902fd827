// Stage timing of chol_block64_kernel (gp.hip): the 64 x 64 diagonal block of the multi-workgroup blocked Cholesky -- factor + inverse with
// 64 identity rows riding along -- which sits on the critical path of the posterior solve from rank 241 on, of Binv above rank 112 and of
// the classic non-rigid CPD (one call per 64 columns).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I gingr_amd/csrc tools/ubench_chol_block64.hip -o tools/bin/ubench_chol_block64
#include <hip/hip_runtime.h>
__device__ unsigned long long g_stage[8];
__shared__ unsigned long long s_stage[8];
__shared__ unsigned long long s_last;
#define GINGR_STAGE_CLOCK(slot)                                          \
    if (threadIdx.x == 0) {                                              \
        const unsigned long long now__ = __builtin_readcyclecounter();   \
        if ((slot) == 7) {                                               \
            for (int q__ = 0; q__ < 8; ++q__) s_stage[q__] = 0;          \
        } else if ((slot) == 6) {                                        \
            for (int q__ = 0; q__ < 6; ++q__) g_stage[q__] += s_stage[q__]; \
        } else {                                                         \
            s_stage[(slot)] += now__ - s_last;                           \
        }                                                                \
        s_last = now__;                                                  \
    }
#define GINGR_CHOL64_STAMPS 1
#include "gp.hip"
TimerScope::TimerScope(gingr_ctx *c, int w) : ctx(c), which(w) {}
void TimerScope::stop() {}
TimerScope::~TimerScope() {}
int64_t gram_wide_ws_doubles(int64_t, int32_t) { return 0; }
int launch_gram_wide(gingr_ctx *, const double *, int64_t, int32_t, const double *, double *, const double *, double *, const ZeroGate *) { return 0; }
void dense_spd_solve3(gingr_ctx *, double *, int64_t, double *, double *, int32_t *) {}
#include <cstdio>
#include <random>
#include <vector>
int main() {
    const int n = 64;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> B(n * n), A(n * n);
    for (auto &v : B) v = nd(rng);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = i == j ? 1.0 : 0.0;
            for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k];
            A[i * n + j] = s;
        }
    double *dA, *dA0, *dLinv;
    int32_t *flag;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dA0, n * n * 8); hipMalloc(&dLinv, n * n * 8); hipMalloc(&flag, 4);
    hipMemcpy(dA0, A.data(), n * n * 8, hipMemcpyHostToDevice);
    const size_t lds = lds_solve_doubles(64, 64) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&chol_block64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int reps = 50;
    unsigned long long zero[8] = {0};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int w = 0; w < 2; ++w) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_stage), zero, sizeof(zero));
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i) {
            hipMemcpyAsync(dA, dA0, n * n * 8, hipMemcpyDeviceToDevice, 0);
            hipLaunchKernelGGL(chol_block64_kernel, dim3(1), dim3(256), lds, 0, dA, (int64_t)n, 0, dLinv, flag);
        }
        hipEventRecord(b); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, a, b);
    }
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stage), sizeof(h));
    const char *names[8] = {"load block + identity -> LDS", "diag 16x16 factor", "panel", "trailing update", "store L and L^-1", "-", "-", "-"};
    printf("chol_block64_kernel: %.1f us per (copy + launch) pair, instrumented\n", ms * 1e3 / reps);
    unsigned long long tot = 0;
    for (int i = 0; i < 5; ++i) tot += h[i];
    for (int i = 0; i < 5; ++i) printf("  %-30s %9.0f cycles  %5.1f %%\n", names[i], (double)h[i] / reps, 100.0 * h[i] / tot);
    std::vector<double> L(n * n), Li(n * n);
    hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(Li.data(), dLinv, n * n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;  // |L L^T - A|, |L^-1 L - I|
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0, t = 0;
            for (int k = 0; k < n; ++k) s += L[i * n + k] * L[j * n + k], t += Li[i * n + k] * L[k * n + j];
            e1 = fmax(e1, fabs(s - A[i * n + j]));
            e2 = fmax(e2, fabs(t - (i == j ? 1.0 : 0.0)));
        }
    printf("  |L L^T - A| %.2e   |L^-1 L - I| %.2e\n", e1, e2);
    return 0;
}

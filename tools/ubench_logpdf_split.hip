// Stage timing of workgroup 1 (the factor of K = S_tot + eps (I + G) and everything behind it) of the two-workgroup transition-density
// kernel (gp.hip: posterior_logpdf_split_kernel), the longer of the two: shader cycles per stage, accumulated by its thread 0.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I gingr_amd/csrc tools/ubench_logpdf_split.hip -o tools/bin/ubench_logpdf_split
#include <hip/hip_runtime.h>
__device__ unsigned long long g_stage[8];
__shared__ unsigned long long s_stage[8];
__shared__ unsigned long long s_last;
#define GINGR_STAGE_CLOCK(slot)                                          \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) {               \
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
#include "gp.hip"

TimerScope::TimerScope(gingr_ctx *c, int w) : ctx(c), which(w) {}
void TimerScope::stop() {}
TimerScope::~TimerScope() {}
int64_t gram_wide_ws_doubles(int64_t, int32_t) { return 0; }
int launch_gram_wide(gingr_ctx *, const double *, int64_t, int32_t, const double *, double *, const double *, double *, const ZeroGate *) { return 0; }

#include <cstdio>
#include <random>
#include <vector>

int main() {
    const int r = 104, rp = 112;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> B((size_t)rp * rp, 0.0), G((size_t)rp * rp, 0.0), S((size_t)rp * rp, 0.0), rhs(rp, 0.0), qte(rp, 0.0);
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j) B[i * rp + j] = nd(rng);
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j) {
            double s = 0;
            for (int k = 0; k < r; ++k) s += B[i * rp + k] * B[j * rp + k];
            G[i * rp + j] = 50.0 * s;
            S[i * rp + j] = 30.0 * s + (i == j ? 5.0 : 0.0);
        }
    for (int i = 0; i < r; ++i) rhs[i] = nd(rng), qte[i] = nd(rng);
    double *dG, *dS, *drhs, *dq, *fx, *out2, *nfac;
    unsigned *sync;
    hipMalloc(&dG, G.size() * 8);
    hipMalloc(&dS, S.size() * 8);
    hipMalloc(&drhs, rp * 8);
    hipMalloc(&dq, rp * 8);
    hipMalloc(&fx, ((size_t)rp * rp + 2 * rp) * 8);
    hipMalloc(&nfac, ((size_t)rp * rp + 16 * rp) * 8);
    hipMalloc(&out2, 16);
    hipMalloc(&sync, 8);
    hipMemset(sync, 0, 8);
    hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(drhs, rhs.data(), rp * 8, hipMemcpyHostToDevice);
    hipMemcpy(dq, qte.data(), rp * 8, hipMemcpyHostToDevice);
    const size_t lds = lds_solve_doubles(rp, 2 * kNB) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&posterior_logpdf_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int reps = 50;
    unsigned long long zero[8] = {0};
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    unsigned epoch = 0;
    for (int keep = 1; keep >= 0; --keep) {
        float ms = 0;
        for (int w = 0; w < 2; ++w) {
            hipMemcpyToSymbol(HIP_SYMBOL(g_stage), zero, sizeof(zero));
            hipEventRecord(a);
            for (int i = 0; i < reps; ++i)
                hipLaunchKernelGGL(posterior_logpdf_split_kernel, dim3(2), dim3(kSolveThreads), lds, 0, r, rp, dG, drhs, dS, dq, fx, out2, sync, ++epoch, keep, nfac);
            hipEventRecord(b);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, a, b);
        }
        unsigned long long h[8];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stage), sizeof(h));
        const char *names[8] = {"load eps G + S -> LDS", "diag 16x16 factor", "panel", "trailing update", "backward", "mat-vec, keep, wait, reduce", "-", "-"};
        printf("posterior_logpdf_split_kernel r=%d keep=%d: %.1f us per launch (back-to-back launches, instrumented); workgroup 1:\n", r, keep, ms * 1e3 / reps);
        unsigned long long tot = 0;
        for (int i = 0; i < 6; ++i) tot += h[i];
        for (int i = 0; i < 6; ++i) printf("  %-28s %9.0f cycles  %5.1f %%\n", names[i], (double)h[i] / reps, 100.0 * h[i] / tot);
        double o[2];
        hipMemcpy(o, out2, 16, hipMemcpyDeviceToHost);
        printf("  logpdf = %.12e  failed = %g\n", o[0], o[1]);
    }
    return 0;
}

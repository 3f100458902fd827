// Stage timing of the LDS-resident posterior solve (gp.hip): shader cycles per stage, accumulated by thread 0.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I gingr_amd/csrc tools/ubench_solve.hip -o tools/ubench_solve
#include <hip/hip_runtime.h>
__device__ unsigned long long g_stage[8];
// accumulators live in LDS (a global read-modify-write per stamp costs more than the stages it measures); stamp 7 resets,
// stamp 6 flushes to global at the end of the kernel
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
#include "gp.hip"

// the launchers of gp.hip reference the context's timer hooks (context.hip); unused here
TimerScope::TimerScope(gingr_ctx *c, int w) : ctx(c), which(w) {}
void TimerScope::stop() {}
TimerScope::~TimerScope() {}
int64_t gram_wide_ws_doubles(int64_t, int32_t) { return 0; }
int launch_gram_wide(gingr_ctx *, const double *, int64_t, int32_t, const double *, double *, const double *, double *, const ZeroGate *) { return 0; }

#include <cstdio>
#include <random>
#include <vector>

int main() {
    const int r = 100, rp = 112;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> B((size_t)rp * rp, 0.0), G((size_t)rp * rp, 0.0), rhs(rp, 0.0);
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j) B[i * rp + j] = nd(rng);
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j) {
            double s = 0;
            for (int k = 0; k < r; ++k) s += B[i * rp + k] * B[j * rp + k];
            G[i * rp + j] = 50.0 * s;
        }
    for (int i = 0; i < r; ++i) rhs[i] = nd(rng);
    double *dG, *drhs, *da;
    DevState *st;
    hipMalloc(&dG, G.size() * 8);
    hipMalloc(&drhs, rp * 8);
    hipMalloc(&da, rp * 8);
    hipMalloc(&st, sizeof(DevState));
    hipMemset(st, 0, sizeof(DevState));
    hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(drhs, rhs.data(), rp * 8, hipMemcpyHostToDevice);
    const size_t lds = lds_solve_doubles(rp, 2 * kNB) * sizeof(double);  // MODE 0: identity rows
    hipFuncSetAttribute(reinterpret_cast<const void *>(&posterior_solve_lds_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    const int reps = 50;
    unsigned long long zero[8] = {0};
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_stage), zero, sizeof(zero));
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL(posterior_solve_lds_kernel<0>, dim3(1), dim3(kSolveThreads), lds, 0, r, rp, dG, drhs, (const double *)nullptr, da, st, (double *)nullptr);
        hipEventRecord(b);
        hipDeviceSynchronize();
    }
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stage), sizeof(h));
    const char *names[8] = {"load G -> LDS", "diag 16x16 factor", "panel", "trailing update", "forward", "backward", "-", "-"};
    printf("posterior_solve_lds_kernel r=%d: %.1f us per launch (back-to-back launches, instrumented)\n", r, ms * 1e3 / reps);
    unsigned long long tot = 0;
    for (int i = 0; i < 6; ++i) tot += h[i];
    for (int i = 0; i < 6; ++i) printf("  %-20s %9.0f cycles  %5.1f %%\n", names[i], (double)h[i] / reps, 100.0 * h[i] / tot);
    std::vector<double> out(rp);
    hipMemcpy(out.data(), da, rp * 8, hipMemcpyDeviceToHost);
    printf("  a[0..2] = %.6e %.6e %.6e\n", out[0], out[1], out[2]);
    return 0;
}

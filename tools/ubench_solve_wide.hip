// Stage timing of the super-panel posterior solve for ranks above 128 (gp.hip: posterior_solve_wide_kernel): shader cycles per stage,
// accumulated by thread 0.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I gingr_amd/csrc tools/ubench_solve_wide.hip -o tools/bin/ubench_solve_wide
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
#include "gp.hip"

TimerScope::TimerScope(gingr_ctx *c, int w) : ctx(c), which(w) {}
void TimerScope::stop() {}
TimerScope::~TimerScope() {}
int64_t gram_wide_ws_doubles(int64_t, int32_t) { return 0; }
int launch_gram_wide(gingr_ctx *, const double *, int64_t, int32_t, const double *, double *, const double *, double *, const ZeroGate *) { return 0; }

#include <cstdio>
#include <random>
#include <vector>

// evicts the caches between two launches (the solve of an iteration runs behind 2 ms of pair loops: its code, G and the workspace are cold)
__global__ void thrash_kernel(double *buf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = buf[i] * 1.0000001 + 1.0;
}

template <int SW>
static void run(int r, int rp, bool cold = false) {
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
    double *dG, *drhs, *da, *work;
    DevState *st;
    hipMalloc(&dG, G.size() * 8);
    hipMalloc(&drhs, rp * 8);
    hipMalloc(&da, rp * 8);
    hipMalloc(&work, (size_t)posterior_work_doubles(rp) * 8);
    hipMalloc(&st, sizeof(DevState));
    hipMemset(st, 0, sizeof(DevState));
    hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(drhs, rhs.data(), rp * 8, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(rp + kNB) * (SW + 1) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&posterior_solve_wide_kernel<SW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int reps = 20;
    unsigned long long zero[8] = {0};
    double *junk = nullptr;
    const size_t njunk = (size_t)64 << 20;  // 512 MB: past L2 and the Infinity Cache
    if (cold) hipMalloc(&junk, njunk * 8), hipMemset(junk, 0, njunk * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_stage), zero, sizeof(zero));
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i) {
            if (cold) hipLaunchKernelGGL(thrash_kernel, dim3(2048), dim3(256), 0, 0, junk, njunk);
            hipLaunchKernelGGL(posterior_solve_wide_kernel<SW>, dim3(1), dim3(kWideSolveThreads), lds, 0, r, rp, dG, drhs, (const double *)nullptr, da, st, work);
        }
        hipEventRecord(b);
        hipDeviceSynchronize();
    }
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stage), sizeof(h));
    const char *names[8] = {"tiles: G - L L^T -> LDS", "diag 16x16 factor", "panel", "trailing update", "write-back", "backward", "-", "-"};
    printf("posterior_solve_wide_kernel<%d> r=%d rp=%d%s: %.1f us per launch (%s, instrumented)\n", SW, r, rp, cold ? " COLD" : "", ms * 1e3 / reps,
           cold ? "a 512 MB read-modify-write between launches; the stage cycles below are the solve's own" : "back-to-back launches");
    unsigned long long tot = 0;
    for (int i = 0; i < 6; ++i) tot += h[i];
    for (int i = 0; i < 6; ++i) printf("  %-26s %9.0f cycles  %5.1f %%\n", names[i], (double)h[i] / reps, 100.0 * h[i] / tot);
    std::vector<double> out(rp);
    hipMemcpy(out.data(), da, rp * 8, hipMemcpyDeviceToHost);
    // residual of (I + G) a = rhs
    double rn = 0, bn = 0;
    for (int i = 0; i < r; ++i) {
        double s = out[i];
        for (int j = 0; j < r; ++j) s += G[i * rp + j] * out[j];
        rn += (s - rhs[i]) * (s - rhs[i]);
        bn += rhs[i] * rhs[i];
    }
    printf("  relative residual %.3e\n", std::sqrt(rn / bn));
}

int main() {
    run<64>(256, 256);
    run<64>(256, 256, true);
    run<64>(200, 208);
    run<32>(512, 512);
    return 0;
}

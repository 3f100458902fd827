// Round 5: the all-pairs passes run at the package power cap (profiles/r05_power_probe.txt).  Would the matrix pipe do the distance part of a
// pair for fewer joules?  Per 256 pairs and wave:
//   form V: 48 float64 vector FMAs + 4 LDS gathers (12 + 1 per 64 pairs: the instruction mix of the column-sum pass)
//   form M: 1 v_mfma_f64_16x16x4 (the four distance instructions of 256 pairs) + 32 vector FMAs + 4 LDS gathers + 1 fragment read
// On gfx950 a float64 MFMA and float64 vector instructions do not overlap (64 cycles = 16 issue slots of 4), so both forms need the same
// issue time: what differs at steady state is the clock the power management grants.  Each form runs for ~3 s in 1024 workgroups x 256
// threads (4 waves per SIMD); the last 2 s are timed.      hipcc -O3 --offload-arch=gfx950 tools/ubench_energy.hip -o tools/bin/ubench_energy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ __launch_bounds__(256) void k(double *out, int steps) {
    __shared__ double tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-3 + i;
    v4f64 c = {0, 0, 0, 0};
    const double m = 1.0000001, d = 1e-9;
    unsigned idx = threadIdx.x * 37u;
    double acc = 0.0;
    for (int s = 0; s < steps; ++s) {
        if (FORM == 1) {
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0], f[1], c, 0, 0, 0);
            acc += tab[(threadIdx.x + s) & 2047];  // the B fragment of the step (regular LDS read)
        }
        constexpr int kFma = FORM == 1 ? 32 : 48;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            idx = idx * 1664525u + 1013904223u;
            const double t = tab[(idx >> 12) & 2047];  // the exponential's table gather of 64 pairs
#pragma unroll
            for (int i = 0; i < kFma / 4; ++i) f[i & 7] = __builtin_fma(f[i & 7], m, d);
            acc += t;
        }
        if (FORM == 1) f[2] += c[0] * 1e-300;
    }
    out[blockIdx.x * 256 + threadIdx.x] = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7])) + acc + c[1];
}

template <int FORM>
void run(double *out, const char *name) {
    const int steps = 20000;  // ~2 ms per launch
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    long launches = 0, timed = 0;
    double timed_s = 0;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    while (std::chrono::duration<double>(clk::now() - t0).count() < 1.0) {
        k<FORM><<<1024, 256>>>(out, steps);
        hipDeviceSynchronize();
        ++launches;
    }
    hipEventRecord(a);
    while (std::chrono::duration<double>(clk::now() - t0).count() < 3.0) {
        for (int i = 0; i < 20; ++i) k<FORM><<<1024, 256>>>(out, steps);
        timed += 20;
        hipDeviceSynchronize();
    }
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    timed_s = ms * 1e-3;
    const double pairs = (double)timed * 1024.0 * 4.0 * steps * 256.0;
    printf("%-28s %6ld launches timed, %.4f ms per launch, %.2f G pair-equivalents/s\n", name, timed, ms / timed, pairs / timed_s * 1e-9);
}

int main() {
    double *out;
    hipMalloc(&out, 1024 * 256 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(out, "form V (48 FMA + 4 gathers)");
        run<1>(out, "form M (MFMA + 32 FMA + ...)");
    }
    return 0;
}

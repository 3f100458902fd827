// How does the shader clock of a WIDE float64 kernel depend on what ran before it?  (round 5: the pair loops of an 8-rank row shard see
// 2.02 GHz inside the iteration and 2.27-2.33 GHz launched back to back -- tools/stamps_shard.py.)
//   loop: [wide: 1024 workgroups x 256 threads of dependent-free float64 FMAs, fixed WORK]  ->  [narrow: ONE wave spinning X us]
//   optionally a "keeper" on a second stream while the narrow kernel runs (event-ordered): sleeping waves / light FMAs / LDS traffic.
// Prints the wide kernel's duration (its own wall-clock stamps, last iteration's median over workgroups) and cycles / wall.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_clock.hip -o tools/bin/ubench_clock && tools/bin/ubench_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                         \
        }                                                                     \
    } while (0)

__global__ __launch_bounds__(256) void wide_kernel(int iters, double *sink, unsigned long long *stamps) {
    const unsigned long long t0 = wall_clock64(), c0 = clock64();
    double a = threadIdx.x, b = a + 1.0, c = a + 2.0, d = a + 3.0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a = __builtin_fma(a, 1.0000001, 1e-9);
            b = __builtin_fma(b, 1.0000001, 1e-9);
            c = __builtin_fma(c, 1.0000001, 1e-9);
            d = __builtin_fma(d, 1.0000001, 1e-9);
        }
    }
    const unsigned long long t1 = wall_clock64(), c1 = clock64();
    if ((a + b) + (c + d) == 12345.678) sink[0] = a;
    if (threadIdx.x == 0) {
        stamps[3 * blockIdx.x] = t0;
        stamps[3 * blockIdx.x + 1] = t1;
        stamps[3 * blockIdx.x + 2] = c1 - c0;
    }
}

__global__ void narrow_kernel(unsigned long long ticks, double *sink) {  // one wave, dependent FMAs (what a single-workgroup solve looks like)
    const unsigned long long t0 = wall_clock64();
    double x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = __builtin_fma(x, 1.0000001, 1e-9);
    }
    if (x == 12345.678) sink[1] = x;
}

// keeper: runs until *flag != 0 (set by the narrow kernel's successor) or `ticks` have passed
__global__ void keeper_kernel(int mode, unsigned long long ticks, double *sink) {
    __shared__ double lds[1024];
    const unsigned long long t0 = wall_clock64();
    double x = threadIdx.x, y = x + 1.0;
    lds[threadIdx.x & 1023] = x;
    while (wall_clock64() - t0 < ticks) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(50);
        } else if (mode == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x = __builtin_fma(x, 1.0000001, 1e-9);
        } else if (mode == 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                x = __builtin_fma(x, 1.0000001, 1e-9);
                y = __builtin_fma(y, 1.0000001, 1e-9);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) x += lds[(threadIdx.x * 17 + k * 33) & 1023];
        }
    }
    if (x + y == 12345.678) sink[2] = x;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1100;  // ~140 us of wide kernel
    double *sink;
    unsigned long long *stamps;
    const int WG = 1024;
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&stamps, 3 * WG * sizeof(unsigned long long)));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::vector<unsigned long long> h(3 * WG);
    struct Keeper {
        const char *name;
        int mode, blocks, threads;
    };
    const Keeper keepers[] = {{"none", -1, 0, 0},          {"sleep 1024x64", 0, 1024, 64}, {"fma 256x64", 1, 256, 64},     {"fma 1024x64", 1, 1024, 64},
                              {"fma2 1024x256", 2, 1024, 256}, {"lds 1024x64", 3, 1024, 64},    {"lds 1024x256", 3, 1024, 256}};
    const double gaps[] = {0.0, 20.0, 60.0, 100.0};
    printf("wide kernel: %d workgroups x 256 threads, 32 FMAs x %d per thread; narrow gap = one wave of dependent FMAs\n", WG, iters);
    printf("%-16s %8s | %10s %10s %10s\n", "keeper", "gap us", "wide us", "GHz", "period us");
    for (const Keeper &kp : keepers) {
        for (double gap : gaps) {
            if (gap == 0.0 && kp.mode >= 0) continue;
            const int reps = 300;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            for (int r = 0; r < reps + 50; ++r) {
                if (r == 50) CK(hipEventRecord(e0, s0));
                hipLaunchKernelGGL(wide_kernel, dim3(WG), dim3(256), 0, s0, iters, sink, stamps);
                if (gap > 0.0) {
                    if (kp.mode >= 0) {
                        CK(hipEventRecord(ev, s0));
                        CK(hipStreamWaitEvent(s1, ev, 0));
                        hipLaunchKernelGGL(keeper_kernel, dim3(kp.blocks), dim3(kp.threads), 0, s1, kp.mode, (unsigned long long)(gap * 100.0), sink);
                    }
                    hipLaunchKernelGGL(narrow_kernel, dim3(1), dim3(64), 0, s0, (unsigned long long)(gap * 100.0), sink);
                }
            }
            CK(hipEventRecord(e1, s0));
            CK(hipStreamSynchronize(s0));
            CK(hipStreamSynchronize(s1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            unsigned long long tmin = ~0ull, tmax = 0;
            std::vector<double> ghz;
            for (int b = 0; b < WG; ++b) {
                tmin = std::min(tmin, h[3 * b]);
                tmax = std::max(tmax, h[3 * b + 1]);
                ghz.push_back((double)h[3 * b + 2] / ((double)(h[3 * b + 1] - h[3 * b]) * 10.0));
            }
            std::sort(ghz.begin(), ghz.end());
            printf("%-16s %8.0f | %10.1f %10.3f %10.1f\n", kp.name, gap, (double)(tmax - tmin) / 100.0, ghz[WG / 2], ms * 1e3 / reps);
            CK(hipEventDestroy(e0));
            CK(hipEventDestroy(e1));
        }
    }
    return 0;
}

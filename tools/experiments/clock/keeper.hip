// Experiment (round 5): does the shader clock of a row shard's iteration follow how "busy" the chip looks between the wide kernels?
// keeper_kernel: `blocks` workgroups whose waves do nothing (mode 0: s_sleep) or issue dependent float64 FMAs (mode 1) for `usec`
// microseconds of the 100 MHz wall clock; launched back to back on a stream of its own next to the registration.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void keeper_kernel(unsigned long long ticks, int mode, double *sink) {
    const unsigned long long t0 = wall_clock64();
    double x = (double)threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(100);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
        }
    }
    if (x == 12345.678) sink[0] = x;
}

static hipStream_t g_stream;
static double *g_sink;

extern "C" int keeper_init() {
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 1;
    if (hipMalloc(&g_sink, 64) != hipSuccess) return 2;
    return 0;
}
// count launches of `usec` each, back to back on the keeper's own stream
extern "C" int keeper_launch(int blocks, int threads, double usec, int mode, int count) {
    for (int i = 0; i < count; ++i) hipLaunchKernelGGL(keeper_kernel, dim3(blocks), dim3(threads), 0, g_stream, (unsigned long long)(usec * 100.0), mode, g_sink);
    return (int)hipGetLastError();
}
extern "C" int keeper_sync() { return (int)hipStreamSynchronize(g_stream); }

// What can be issued in the shadow of v_mfma_f64_16x16x4_f64 on gfx950?  One workgroup per CU, 2 waves per SIMD (512 threads, the
// Gram kernel's shape) or 1 (256 threads); every wave runs ITERS iterations of 14 independent MFMAs with one kind of filler slotted
// between them, and reports shader cycles (s_memtime) per iteration.  If a filler is free, its column equals the MFMA-only column.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_f64_fill.hip -o /tmp/ubench_fill && /tmp/ubench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

// FILL: 0 none | 1 two 32-bit integer VALU per gap | 2 one v_mul_f64 per gap | 3 one ds_write_b64 + one ds_read_b64 per gap
//       4 one s_barrier per iteration (middle) | 5 one global load per second gap (L2 resident) | 6 two v_mov_b64 per gap
//       7 one 64-bit integer multiply-add (v_mad_u64_u32) per gap
template <int FILL, bool WITH_MFMA>
__global__ __launch_bounds__(512) void k(long long *cycles, double *sink, const double *src, int iters) {
    __shared__ double lds[8 * 64 * 2];
    v4f64 acc[14];
#pragma unroll
    for (int q = 0; q < 14; ++q) acc[q] = v4f64{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 1.0 + lane * 1e-3, b = 0.5;
    unsigned i0 = lane, i1 = lane * 3 + 1;
    double f0 = a, f1 = b, m0 = 1.0, m1 = 2.0;
    unsigned long long u0 = lane;
    double *my = lds + wave * 128 + lane;
    const double *gp = src + threadIdx.x;
    double g = 0.0;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 14; ++q) {
            if (WITH_MFMA) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (FILL == 1) {
                asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %0" : "+v"(i0), "+v"(i1));
            } else if (FILL == 2) {
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f0) : "v"(f1));
            } else if (FILL == 3) {
                asm volatile("ds_write_b64 %1, %2\n ds_read_b64 %0, %1 offset:512" : "=v"(m0) : "v"((unsigned)(size_t)my), "v"(m1) : "memory");
            } else if (FILL == 4) {
                if (q == 6) __builtin_amdgcn_s_barrier();
            } else if (FILL == 5) {
                if (q & 1) g += gp[(q >> 1) * 512];
            } else if (FILL == 6) {
                asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %0" : "+v"(m0), "+v"(m1));
            } else if (FILL == 7) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(u0) : "v"(i0) : "vcc");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FILL == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    double s = f0 + m0 + m1 + g + (double)(i0 + i1) + (double)u0;
#pragma unroll
    for (int q = 0; q < 14; ++q) s += acc[q][0] + acc[q][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int FILL, bool WITH_MFMA>
static void run(const char *what, int threads, long long *dc, double *sink, const double *src) {
    const int iters = 200, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FILL, WITH_MFMA>), dim3(blocks), dim3(threads), 0, 0, dc, sink, src, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FILL, WITH_MFMA>), dim3(blocks), dim3(threads), 0, 0, dc, sink, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), dc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    const int waves = threads / 64;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) sum += (double)h[b * 8 + w];
    const double cyc = sum / (blocks * waves) / iters;
    printf("%-44s %s  waves/SIMD %d: %8.1f cycles per iteration of 14 gaps (%6.1f per gap), wall %7.1f ns per iteration => %.2f GHz\n", what,
           WITH_MFMA ? "with 14 MFMA" : "filler only ", threads / 256, cyc, cyc / 14, ms * 1e6 / iters, cyc / (ms * 1e6 / iters));
}

int main() {
    long long *dc;
    double *sink, *src;
    hipMalloc(&dc, 256 * 8 * sizeof(long long));
    hipMalloc(&sink, 256 * 512 * sizeof(double));
    hipMalloc(&src, 8 * 512 * sizeof(double));
    hipMemset(src, 0, 8 * 512 * sizeof(double));
    for (int threads : {256, 512}) {
        run<0, true>("none", threads, dc, sink, src);
        run<1, true>("2 x 32-bit integer VALU per gap", threads, dc, sink, src);
        run<1, false>("2 x 32-bit integer VALU per gap", threads, dc, sink, src);
        run<2, true>("1 x v_mul_f64 per gap", threads, dc, sink, src);
        run<2, false>("1 x v_mul_f64 per gap", threads, dc, sink, src);
        run<3, true>("ds_write_b64 + ds_read_b64 per gap", threads, dc, sink, src);
        run<3, false>("ds_write_b64 + ds_read_b64 per gap", threads, dc, sink, src);
        run<4, true>("one s_barrier per iteration", threads, dc, sink, src);
        run<4, false>("one s_barrier per iteration", threads, dc, sink, src);
        run<5, true>("one global load per second gap", threads, dc, sink, src);
        run<5, false>("one global load per second gap", threads, dc, sink, src);
        run<6, true>("2 x v_mov_b64 per gap", threads, dc, sink, src);
        run<6, false>("2 x v_mov_b64 per gap", threads, dc, sink, src);
        run<7, true>("1 x v_mad_u64_u32 per gap", threads, dc, sink, src);
        run<7, false>("1 x v_mad_u64_u32 per gap", threads, dc, sink, src);
    }
    return 0;
}

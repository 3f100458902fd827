// Instruction-throughput microbenchmark for the float64 VALU ops the affinity kernels are made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_f64.hip -o tools/ubench_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double c = 1.0000001, d = 0.5;
    int e = 1;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    __shared__ double T[64];
    if (threadIdx.x < 64) T[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));) }
        if (OP == 1) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));) }
        if (OP == 2) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 3) { REP8(asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 4) { REP8(asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7\n v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (OP == 5) { REP8(asm volatile("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));) }
        if (OP == 6) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0001f), "v"(0.5f));) }
        if (OP == 7) { REP8(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(0x7fffffff));) }
        if (OP == 8) { REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"((i0 * 40503 & 63) * 8), "v"((i1 * 40503 & 63) * 8), "v"((i2 * 40503 & 63) * 8), "v"((i3 * 40503 & 63) * 8) : "memory");) }
        if (OP == 9) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));) }
        if (OP == 10) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 11) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 12) { REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 13) { REP8(asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));) }
        if (OP == 14) { REP8(asm volatile("v_fract_f64 %0, %0\n v_fract_f64 %1, %1\n v_fract_f64 %2, %2\n v_fract_f64 %3, %3\n v_floor_f64 %4, %4\n v_floor_f64 %5, %5\n v_floor_f64 %6, %6\n v_floor_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + f0 + f1 + f2 + f3;
}

typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kmfma(double *out, int iters) {
    v4f64 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    double a = threadIdx.x, b = 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

template <int OP>
void run(const char *name, int waves_per_simd, double *out) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(out, iters, 1.0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per SIMD: waves_per_simd waves each issuing iters*64 wave-instructions (ds test: 32 + waits)
    const double instr = (double)iters * 64 * waves_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}

int main() {
    double *out;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(double));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, out);
        run<1>("v_add_f64", w, out);
        run<2>("v_mul_f64", w, out);
        run<3>("v_rndne_f64", w, out);
        run<4>("v_cvt_i32_f64", w, out);
        run<5>("v_ldexp_f64", w, out);
        run<6>("v_fma_f32", w, out);
        run<7>("int and/shift b32", w, out);
        run<8>("ds_read_b64 rand (x32+wait)", w, out);
        run<9>("v_pk_fma_f32", w, out);
        run<10>("v_rcp_f64", w, out);
        run<11>("v_exp_f32", w, out);
        run<12>("cvt f32<->f64", w, out);
        run<13>("v_max_f64", w, out);
        run<14>("v_fract/floor_f64", w, out);
    }
    {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (int w : {1, 2}) {
            const int iters = 500;
            kmfma<<<256 * w, 256>>>(out, 2);
            hipDeviceSynchronize();
            hipEventRecord(a);
            kmfma<<<256 * w, 256>>>(out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double n = (double)iters * 64 * w;
            printf("v_mfma_f64_16x16x4 waves/SIMD=%d %.3f ms -> %.2f ns per MFMA per SIMD (= %.1f cycles @2.4GHz) => %.1f TFLOP/s\n", w, ms,
                   ms * 1e6 / n, ms * 1e6 / n * 2.4, 2048.0 * n * 1024 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

// One closing experiment on the CPD pair loops: would v_mfma_f64_4x4x4_4b take the four accumulations of pass 2
// (a1 += p, ax += p x, ay += p y, az += p z: a K-tile x [1 | x | y | z] product with N = 4 exactly) off the vector ALU's issue slots?
// One instruction = 4 blocks x (4 x 4 x 4) = 256 multiply-adds = what FOUR v_fma_f64 do for 64 pairs.  It pays only if it
//   (a) issues in fewer than the 16 cycles the four v_fma_f64 take, or
//   (b) runs beside vector float64 work (the other ~9 instructions of a pair).
// MODE 0: 8 independent v_mfma_f64_4x4x4 per iteration; 1: 32 v_fma_f64; 2: both interleaved; 3: 2 v_mfma_f64_16x16x4 (reference).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_mfma_4x4.hip -o tools/bin/ubench_mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    v4f64 C0 = {0, 0, 0, 0}, C1 = C0;
    double a = threadIdx.x * 1e-3, b = 1.0;
    double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
    const double m = 1.0000001, d = 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) c[4 * h + q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[4 * h + q], 0, 0, 0);
            }
            if (MODE == 3) {
                if (h == 0) C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C0, 0, 0, 0);
                else C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C1, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(m), "v"(d));
            }
        }
    }
    double s = C0[0] + C1[1] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    for (int q = 0; q < 8; ++q) s += c[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
float run(double *out, int waves, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * waves, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256 * waves, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    double *out; hipMalloc(&out, 256 * 256 * 8 * 8);
    const int iters = 4000;
    for (int w : {1, 2, 4}) {
        const float t0 = run<0>(out, w, iters), t1 = run<1>(out, w, iters), t2 = run<2>(out, w, iters), t3 = run<3>(out, w, iters);
        // per SIMD: w waves x iters x 8 MFMA 4x4x4 (MODE 0) resp. 32 v_fma_f64 (MODE 1)
        printf("waves/SIMD=%d: 8 mfma_4x4x4/iter %.3f ms (%.1f ns each per SIMD) | 32 v_fma_f64/iter %.3f ms (%.2f ns each) | both %.3f ms (sum %.3f, max %.3f) | 2 mfma_16x16x4/iter %.3f ms (%.1f ns each)\n",
               w, t0, t0 * 1e6 / (8.0 * iters * w), t1, t1 * 1e6 / (32.0 * iters * w), t2, t0 + t1, t0 > t1 ? t0 : t1, t3, t3 * 1e6 / (2.0 * iters * w));
    }
    printf("four v_fma_f64 (what one mfma_4x4x4 would replace for 64 pairs) = 4 x the v_fma figure above; the replacement pays only if the mfma figure is below that or `both` is near `max`.\n");
    return 0;
}

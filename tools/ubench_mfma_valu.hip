// Does v_mfma_f64_16x16x4_f64 overlap with f64 VALU work on gfx950?  MODE 0: 4 MFMA per iteration; MODE 1: 32 v_fma_f64;
// MODE 2: both interleaved (independent data).  If the pipes are independent, t2 ~ max(t0, t1); if shared, t2 ~ t0 + t1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
    v4f64 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0;
    double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
    const double m = 1.0000001, d = 0.5;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(m), "v"(d));
        }
        if (MODE == 0 || MODE == 2) {
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(m), "v"(d));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int MODE>
float run(double *out, int waves) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * waves, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256 * waves, 256>>>(out, 4000);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    double *out; hipMalloc(&out, 256 * 256 * 8 * 8);
    for (int w : {1, 2, 4}) {
        float t0 = run<0>(out, w), t1 = run<1>(out, w), t2 = run<2>(out, w);
        printf("waves/SIMD=%d: 4 MFMA/iter %.3f ms | 32 v_fma_f64/iter %.3f ms | both %.3f ms  (sum %.3f, max %.3f)\n", w, t0, t1, t2, t0 + t1, t0 > t1 ? t0 : t1);
    }
    return 0;
}

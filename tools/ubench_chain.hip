// Latency / issue cost of the instruction kinds on the sequential chains of the one-workgroup kernels (gp.hip: lds_cholesky), one
// wave per SIMD, shader cycles by s_memtime.   hipcc -O3 --offload-arch=gfx950 tools/ubench_chain.hip -o tools/bin/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__global__ void k(unsigned long long *out, double *sink, int active) {
    if ((int)threadIdx.x >= active) return;
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999, c = 1e-9, d0 = 1.1, d1 = 1.2, d2 = 1.3, d3 = 1.4, d4 = 1.5, d5 = 1.6, d6 = 1.7, d7 = 1.8;
    unsigned long long t[16];
    int n = 0;
#define TIME(body)                                              \
    {                                                           \
        __builtin_amdgcn_s_waitcnt(0);                          \
        const unsigned long long t0 = __builtin_readcyclecounter(); \
        body __builtin_amdgcn_s_waitcnt(0);                     \
        asm volatile("s_nop 0" ::: "memory");                   \
        t[n++] = __builtin_readcyclecounter() - t0;             \
    }
    // 0: dependent v_fma_f64 chain (64)
    TIME(REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));))
    // 1: 8 independent v_fma_f64 chains, 64 instructions
    TIME(REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b), "v"(c));))
    // 2: dependent v_mul_f64 chain (64)
    TIME(REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));))
    // 3: independent v_fmac_f64_dpp row_newbcast stream (64), 8 accumulators
    TIME(REP16(asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));))
    // 4: dependent v_fmac_f64_dpp chain (64): acc feeds the next one's accumulator only
    TIME(REP64(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));))
    // 5: dependent through the DPP source: v_mov_b64_dpp of the value just produced by a v_mul (s_nop 1 in between), 32 pairs
    TIME(REP16(asm volatile("v_mul_f64 %0, %0, %1\n\ts_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mul_f64 %0, %0, %1\n\ts_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));))
    // 6: dependent v_rsq_f64 chain (16)
    TIME(REP16(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));))
    // 7: the Newton chain of one column: rsq, mul, mul, fma, fma, mul, fma, fma, mul, mul (x16)
    TIME(REP16(asm volatile("v_rsq_f64 %1, %0\n\tv_mul_f64 %2, %0, 0.5\n\tv_mul_f64 %3, %2, %1\n\tv_fma_f64 %3, -%3, %1, 0.5\n\tv_fma_f64 %1, %1, %3, %1\n\tv_mul_f64 %3, %2, %1\n\tv_fma_f64 %3, -%3, %1, 0.5\n\tv_fma_f64 %1, %1, %3, %1\n\tv_mul_f64 %0, %0, %1\n\tv_mul_f64 %0, %0, %0" : "+v"(a), "=&v"(d0), "=&v"(d1), "=&v"(d2));))
    // 8: ds_read_b64 dependent round trip (address from the value): 16
    // 9: empty
    TIME(;)
    if (threadIdx.x == 0)
        for (int i = 0; i < n; ++i) out[i] = t[i];
    sink[threadIdx.x] = a + d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}
int main(int argc, char **argv) {
    const int act = argc > 1 ? atoi(argv[1]) : 256;
    printf("active threads %d\n", act);
    unsigned long long *o, h[16];
    double *s;
    hipMalloc(&o, 128);
    hipMalloc(&s, 8 * 256);
    for (int w = 0; w < 2; ++w) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, o, s, act);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, o, 128, hipMemcpyDeviceToHost);
    const char *nm[] = {"dependent v_fma_f64 x64", "8 independent v_fma_f64 chains x64", "dependent v_mul_f64 x64", "independent v_fmac_f64_dpp x64",
                        "v_fmac_f64_dpp chain through the accumulator x64", "v_mul + s_nop 1 + v_mov_b64_dpp dependent x32 pairs",
                        "dependent v_rsq_f64 x16", "rsq + Newton column chain (10 instr) x16", "empty"};
    const int cnt[] = {64, 64, 64, 64, 64, 32, 16, 16, 1};
    for (int i = 0; i < 9; ++i) printf("%-55s %7llu cycles  = %6.1f per item\n", nm[i], h[i], (double)(h[i] - h[8]) / cnt[i]);
    return 0;
}

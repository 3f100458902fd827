// cycles of ONE 16 x 16 diagonal block of lds_cholesky (gp.hip) by itself, cold and warm instruction cache
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I gingr_amd/csrc tools/ubench_diag.hip -o tools/bin/ubench_diag
#include <hip/hip_runtime.h>
#include "gp.hip"
TimerScope::TimerScope(gingr_ctx *c, int w) : ctx(c), which(w) {}
void TimerScope::stop() {}
TimerScope::~TimerScope() {}
#include <cstdio>
__global__ __launch_bounds__(256) void kd(const double *M, unsigned long long *out, double *res) {
    __shared__ double A[32 * 17 + 64];
    __shared__ int bad;
    double *rd = A + 32 * 17;
    for (int pass = 0; pass < 4; ++pass) {
        for (int e = threadIdx.x; e < 256; e += 256) A[(e >> 4) * 17 + (e & 15)] = M[e];
        if (threadIdx.x == 0) bad = 0;
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        lds_cholesky<256>(A, 17, 16, rd, &bad, 0);
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) out[pass] = t1 - t0;
        __syncthreads();
    }
    if (threadIdx.x < 16) res[threadIdx.x] = A[threadIdx.x * 17 + threadIdx.x];
}
int main() {
    double h[256];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double *M, *res;
    unsigned long long *o, ho[4];
    hipMalloc(&M, sizeof(h));
    hipMalloc(&res, 128);
    hipMalloc(&o, 64);
    hipMemcpy(M, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w = 0; w < 3; ++w) {
        hipLaunchKernelGGL(kd, dim3(1), dim3(256), 0, 0, M, o, res);
        hipDeviceSynchronize();
        hipMemcpy(ho, o, 32, hipMemcpyDeviceToHost);
        printf("launch %d: one diagonal block (lds_cholesky n = 16, three barriers): pass 0 %llu, 1 %llu, 2 %llu, 3 %llu cycles\n", w, ho[0], ho[1], ho[2], ho[3]);
    }
    double r[16];
    hipMemcpy(r, res, 128, hipMemcpyDeviceToHost);
    printf("L[0][0] = %.15g (sqrt(21) = 4.58257569495584)  L[15][15] = %.15g\n", r[0], r[15]);
    return 0;
}

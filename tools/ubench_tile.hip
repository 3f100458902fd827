// Where one 16 x 16 tile of the trailing update (gp.hip: lds_cholesky) spends its cycles: one wave, shader cycles by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kt(unsigned long long *out, double *sink, int ld, int active_waves) {
    extern __shared__ double A[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 144 * ld; e += 256) A[e] = 1e-3 * (e % 97);
    __syncthreads();
    if (wave >= active_waves) return;
    const int l15 = lane & 15, l4 = lane >> 4;
    unsigned long long t[8], tstart_keep = 0;
    double acc_out = 0;
    for (int rep = 0; rep < 3; ++rep) {
        const int kb = 0, i0 = 32 + 16 * wave, j0 = 16;
        const double *pa = A + (i0 + l15) * ld + kb + l4, *pb = A + (j0 + l15) * ld + kb + l4;
        double *pc = A + (i0 + l4) * ld + j0 + l15;
        __builtin_amdgcn_s_waitcnt(0);
        t[0] = __builtin_readcyclecounter();
        tstart_keep = t[0];
        v4f64 acc;
        double a[4], b[4];
        for (int g = 0; g < 4; ++g) acc[g] = pc[4 * g * ld];
        for (int q = 0; q < 4; ++q) { a[q] = pa[4 * q]; b[q] = pb[4 * q]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t[1] = __builtin_readcyclecounter();
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[q], b[q], acc, 0, 0, 0);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
        acc_out += acc[0];
        t[2] = __builtin_readcyclecounter();
        for (int g = 0; g < 4; ++g) pc[4 * g * ld] = acc[g];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t[3] = __builtin_readcyclecounter();
        // twelve contiguous 8-byte reads (lane-consecutive doubles: no bank conflict by construction), then six 16-byte reads
        {
            double u[12];
            const double *pl = A + lane;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t[5] = __builtin_readcyclecounter();
            for (int q = 0; q < 12; ++q) u[q] = pl[64 * q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t[6] = __builtin_readcyclecounter();
            for (int q = 0; q < 12; ++q) acc_out += u[q];
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 w[6];
            const d2 *pw = reinterpret_cast<const d2 *>(A) + lane;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t7 = __builtin_readcyclecounter();
            for (int q = 0; q < 6; ++q) w[q] = pw[64 * q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t[7] = __builtin_readcyclecounter() - t7;
            for (int q = 0; q < 6; ++q) acc_out += w[q][0] + w[q][1];
        }
        // fragments as two 16-byte reads each (lane (l15, l4) takes k = 4 l4 .. 4 l4 + 3 of row l15: needs an even ld), C as before
        if ((ld & 1) == 0) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 *qa = reinterpret_cast<const d2 *>(A + (i0 + l15) * ld + kb + 4 * l4), *qb = reinterpret_cast<const d2 *>(A + (j0 + l15) * ld + kb + 4 * l4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t8 = __builtin_readcyclecounter();
            v4f64 cc;
            for (int g = 0; g < 4; ++g) cc[g] = pc[4 * g * ld];
            const d2 a0 = qa[0], a1 = qa[1], b0 = qb[0], b1 = qb[1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t[0] = __builtin_readcyclecounter() - t8;
            acc_out += cc[0] + cc[1] + cc[2] + cc[3] + a0[0] + a0[1] + a1[0] + a1[1] + b0[0] + b0[1] + b1[0] + b1[1];
        } else {
            t[0] = 0;
        }
        // four INDEPENDENT mfmas
        v4f64 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], c3, 0, 0, 0);
        acc_out += c0[0] + c1[1] + c2[2] + c3[3];
        t[4] = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) out[i] = t[i];
        out[8] = tstart_keep;
    sink[threadIdx.x] = acc_out;
}
int main() {
    unsigned long long *o, h[16];
    double *s;
    hipMalloc(&o, 128);
    hipMalloc(&s, 8 * 256);
    const int lds = 144 * 126 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&kt), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int aw = 4; aw <= 4; aw += 3)
        for (int ld = 112; ld <= 126; ++ld) {
            hipLaunchKernelGGL(kt, dim3(1), dim3(256), lds, 0, o, s, ld, aw);
            hipDeviceSynchronize();
            hipMemcpy(h, o, 72, hipMemcpyDeviceToHost);
            printf("waves %d ld %d: 12 LDS reads %llu | 4 dependent MFMAs (+ result) %llu | 4 LDS writes %llu | 4 independent MFMAs (+ result) %llu cycles\n", aw, ld,
                   h[1] - h[8], h[2] - h[1], h[3] - h[2], h[4] - h[3]);
            printf("      12 contiguous 8-byte reads %llu | 6 contiguous 16-byte reads %llu | tile with 16-byte fragment reads (4 + 4 reads) %llu cycles\n", h[6] - h[5], h[7], h[0]);
        }
    return 0;
}

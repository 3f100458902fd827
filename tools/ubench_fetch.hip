// Calibration of the FETCH_SIZE counter for the access widths this library uses (MI355X_MICROARCH.md: FETCH_SIZE reports half
// the bytes of 16 B / lane streaming reads; "other access widths are uncalibrated").  Each kernel reads the same 256 MiB buffer
// exactly once; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (tools/pmc_fetch_calibration.sh) and compare the counter
// with 262144 KiB.
//   read16_kernel   16 B per lane, contiguous (the basis sweeps)
//   read8_kernel     8 B per lane, contiguous 512 B per wave
//   read8x4_kernel   8 B per lane as four 128-byte row segments per wave instruction (the Gram kernel's fragment loads)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read16_kernel(const double2 *p, size_t n, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = p[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}
__global__ void read8_kernel(const double *p, size_t n, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}
// rows of `rp` doubles; a wave reads rows r..r+3, columns c0..c0+15 (lane = 16 * row + column), then the next 16 columns
__global__ void read8x4_kernel(const double *p, size_t rows, int rp, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4, cl = lane & 15;
    const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6), w = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    double s = 0;
    for (size_t r = 4 * w; r + 3 < rows; r += 4 * waves)
        for (int t = 0; t < rp / 16; ++t) s += p[(r + kq) * rp + 16 * t + cl];
    if (s == 12345.678) out[0] = s;
}

int main() {
    const size_t bytes = 256ull << 20, n = bytes / 8;
    double *p, *out;
    hipMalloc(&p, bytes);
    hipMalloc(&out, 8);
    hipMemset(p, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read16_kernel, dim3(2048), dim3(256), 0, 0, (const double2 *)p, n / 2, out);
        hipLaunchKernelGGL(read8_kernel, dim3(2048), dim3(256), 0, 0, p, n, out);
        hipLaunchKernelGGL(read8x4_kernel, dim3(2048), dim3(256), 0, 0, p, n / 112, 112, out);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}

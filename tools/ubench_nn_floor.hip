// What a ONE-round launch costs at the size of BASELINE config 2 (stateless closest point, 5 000 x 5 000) -- the floor under any exact
// scheme on this grid.  Same grid as nn_small_kernel's first pass (10 query groups x 76 target slices, 256
// threads, 3 workgroups per compute unit):
//   K0  empty                                   -> dispatch + ramp of 760 workgroups
//   K1  slice -> LDS, barrier, one store        -> + one global round trip per workgroup
//   K2  K1 + the pair loop (9 instructions per pair, two queries per lane, no index)  = the first pass itself
//   K3  K2 with the index tracked (compare + two selects per pair)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/ubench_nn_floor.hip -o tools/bin/ubench_nn_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
__device__ __forceinline__ double norm2_exact(double dx, double dy, double dz) { return (dx * dx + dy * dy) + dz * dz; }
template <int MODE>
__global__ __launch_bounds__(256) void k(const double *qx, const double *qy, const double *qz, int M, const double *tx, const double *ty,
                                         const double *tz, int n_targets, int slice_len, double *pd2, int *pidx) {
    if (MODE == 0) return;
    extern __shared__ double sh[];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.y * slice_len, n = min(slice_len, n_targets - j0);
    double *sx = sh, *sy = sh + slice_len, *sz = sh + 2 * slice_len;
    for (int kk = tid; kk < n; kk += 256) sx[kk] = tx[j0 + kk], sy[kk] = ty[j0 + kk], sz[kk] = tz[j0 + kk];
    const int ia = blockIdx.x * 512 + tid, ib = ia + 256;
    const bool oka = ia < M, okb = ib < M;
    const double ax = oka ? qx[ia] : 0.0, ay = oka ? qy[ia] : 0.0, az = oka ? qz[ia] : 0.0;
    const double bx = okb ? qx[ib] : 0.0, by = okb ? qy[ib] : 0.0, bz = okb ? qz[ib] : 0.0;
    __syncthreads();
    double besta = __builtin_huge_val(), bestb = besta;
    int ja = -1, jb = -1;
    if (MODE >= 2) {
        for (int j = 0; j < n; ++j) {
            const double x = sx[j], y = sy[j], z = sz[j];
            const double da = norm2_exact(x - ax, y - ay, z - az), db = norm2_exact(x - bx, y - by, z - bz);
            if (MODE == 3) {
                if (da < besta) besta = da, ja = j0 + j;
                if (db < bestb) bestb = db, jb = j0 + j;
            } else {
                besta = fmin(besta, da);
                bestb = fmin(bestb, db);
            }
        }
    } else {
        besta = sx[tid % (n > 0 ? n : 1)];
    }
    if (oka) pd2[(size_t)blockIdx.y * M + ia] = besta;
    if (okb) pd2[(size_t)blockIdx.y * M + ib] = bestb;
    if (MODE == 3) {
        if (oka) pidx[(size_t)blockIdx.y * M + ia] = ja;
        if (okb) pidx[(size_t)blockIdx.y * M + ib] = jb;
    }
}
// K4: four queries per lane (1 024 per workgroup), unrolled by four targets, no index
__global__ __launch_bounds__(256) void k4(const double *qx, const double *qy, const double *qz, int M, const double *tx, const double *ty,
                                          const double *tz, int n_targets, int slice_len, double *pd2) {
    extern __shared__ double sh[];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.y * slice_len, n = min(slice_len, n_targets - j0);
    double *sx = sh, *sy = sh + slice_len, *sz = sh + 2 * slice_len;
    for (int kk = tid; kk < n; kk += 256) sx[kk] = tx[j0 + kk], sy[kk] = ty[j0 + kk], sz[kk] = tz[j0 + kk];
    double ax[4], ay[4], az[4], best[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = blockIdx.x * 1024 + tid + 256 * u;
        const bool ok = i < M;
        ax[u] = ok ? qx[i] : 0.0, ay[u] = ok ? qy[i] : 0.0, az[u] = ok ? qz[i] : 0.0;
        best[u] = __builtin_huge_val();
    }
    __syncthreads();
    int j = 0;
    for (; j + 2 <= n; j += 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const double x = sx[j + kk], y = sy[j + kk], z = sz[j + kk];
#pragma unroll
            for (int u = 0; u < 4; ++u) best[u] = fmin(best[u], norm2_exact(x - ax[u], y - ay[u], z - az[u]));
        }
    }
    for (; j < n; ++j) {
        const double x = sx[j], y = sy[j], z = sz[j];
#pragma unroll
        for (int u = 0; u < 4; ++u) best[u] = fmin(best[u], norm2_exact(x - ax[u], y - ay[u], z - az[u]));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = blockIdx.x * 1024 + tid + 256 * u;
        if (i < M) pd2[(size_t)blockIdx.y * M + i] = best[u];
    }
}
float run4(const double *q, const double *t, int M, int N, double *pd2, int wgs) {
    const int groups = (M + 1023) / 1024, ns = wgs / groups, len = (N + ns - 1) / ns, nslices = (N + len - 1) / len;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 200;
    float ms = 0;
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL(k4, dim3(groups, nslices), dim3(256), (size_t)3 * len * 8, 0, q, q + M, q + 2 * M, M, t, t + N, t + 2 * N, N, len, pd2);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    return ms * 1e3f / reps;
}
template <int MODE>
float run(const double *q, const double *t, int M, int N, double *pd2, int *pidx) {
    const int groups = (M + 511) / 512, ns = 768 / groups, len = (N + ns - 1) / ns, nslices = (N + len - 1) / len;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 200;
    float ms = 0;
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL(k<MODE>, dim3(groups, nslices), dim3(256), (size_t)3 * len * 8, 0, q, q + M, q + 2 * M, M, t, t + N, t + 2 * N, N, len, pd2, pidx);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    return ms * 1e3f / reps;
}
int main() {
    const int M = 5000, N = 5000;
    std::mt19937_64 rng(7); std::normal_distribution<double> nd(0, 50);
    std::vector<double> h(3 * M); for (auto &v : h) v = nd(rng);
    double *q, *t, *pd2; int *pidx;
    hipMalloc(&q, 3 * M * 8); hipMalloc(&t, 3 * N * 8); hipMalloc(&pd2, (size_t)512 * M * 8); hipMalloc(&pidx, (size_t)512 * M * 4);
    hipMemcpy(q, h.data(), 3 * M * 8, hipMemcpyHostToDevice);
    for (auto &v : h) v = nd(rng);
    hipMemcpy(t, h.data(), 3 * N * 8, hipMemcpyHostToDevice);
    printf("5 000 x 5 000, grid 10 x 76 x 256 threads, back-to-back launches on one stream (us per launch):\n");
    printf("  K0 empty                         %6.2f\n", run<0>(q, t, M, N, pd2, pidx));
    printf("  K1 slice -> LDS, barrier, store  %6.2f\n", run<1>(q, t, M, N, pd2, pidx));
    printf("  K2 + pair loop, no index         %6.2f   (the first pass of gingr_nn at this size)\n", run<2>(q, t, M, N, pd2, pidx));
    printf("  K3 + pair loop, index tracked    %6.2f\n", run<3>(q, t, M, N, pd2, pidx));
    for (int wgs : {512, 768, 1024, 1536})
        printf("  K4 four queries per lane, %4d workgroups  %6.2f\n", wgs, run4(q, t, M, N, pd2, wgs));
    return 0;
}

// The reference's classic rigid / similarity ICP baseline (G/other/algorithms/icp/RigidICP.scala:24-84, ICPFactory.scala:28-38,
// G/other/algorithms/RigidICPRegistration.scala:24-46, G/other/utils/PoseRegistrator.scala:27-43) on the device: one iteration =
// exact closest target point of every template point (the nearest-neighbour kernel of the GiNGR ICP path: lowest index on ties),
// the least-squares rigid / similarity transform of the pairs (LandmarkRegistration.rigid3D / similarity3DLandmarkRegistration
// about the origin, with scalismo's Euler round trip of the rotation), and its application to the template.  The template stays
// in HBM between iterations; one scalar (the mean distance) comes back per iteration for the caller's convergence test.
#include "common.h"
#include "svd3.h"

#include <cmath>
#include <vector>

struct gingr_rigid_icp {
    gingr_ctx *ctx = nullptr;
    int32_t kind = 0;  // 0 rigid (RigidRegistrator3D), 1 similarity (AffineRegistrator3D)
    int64_t M = 0, N = 0;
    DevBuf stage, tpl, tgt, torig, tboxes, ws, idx, d2, part, tr;
    double c0[3] = {0, 0, 0};  // fixed centring point of the sums (centroid of the initial template)
    double last_distance = 0.0;
    bool warm = false;  // idx holds the previous iteration's matches
    NNGrid tgrid;       // uniform grid over the target (nn_grid.hip)
    ~gingr_rigid_icp() { nn_grid_free(&tgrid); }
};

namespace {

constexpr int kIcpBlocks = 256;
constexpr int kSums = 17;  // sum x~ (3), sum y~ (3), sum y~ x~^T (9), sum |x~|^2, sum |x - y|

// idx: positions in the (Morton-ordered) target cloud t0
__global__ __launch_bounds__(256) void icp_sums_kernel(Cloud p, Cloud t0, const int32_t *__restrict__ idx, const double *__restrict__ d2,
                                                       double c0x, double c0y, double c0z, double *__restrict__ part) {
    __shared__ double sh[256];
    double s[kSums];
    for (int q = 0; q < kSums; ++q) s[q] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (int64_t)kIcpBlocks * 256) {
        const int64_t j = idx[i];
        if (j < 0 || j >= t0.n) {  // the scan leaves -1 for a query whose distances are all NaN (non-finite coordinates): no read out
            s[0] += __builtin_nan("");  // of bounds; the NaN makes the transform non-finite, which the caller reports
            continue;
        }
        const double x0 = p.x[i] - c0x, x1 = p.y[i] - c0y, x2 = p.z[i] - c0z;
        const double y0 = t0.x[j] - c0x, y1 = t0.y[j] - c0y, y2 = t0.z[j] - c0z;
        s[0] += x0; s[1] += x1; s[2] += x2;
        s[3] += y0; s[4] += y1; s[5] += y2;
        s[6] += y0 * x0; s[7] += y0 * x1; s[8] += y0 * x2;
        s[9] += y1 * x0; s[10] += y1 * x1; s[11] += y1 * x2;
        s[12] += y2 * x0; s[13] += y2 * x1; s[14] += y2 * x2;
        s[15] += x0 * x0 + x1 * x1 + x2 * x2;
        s[16] += sqrt(d2[i]);  // (pt - closestPoint).norm, RigidICP.scala:67
    }
    for (int q = 0; q < kSums; ++q) {
        sh[threadIdx.x] = s[q];
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * kSums + q] = sh[0];
        __syncthreads();
    }
}

// tr[0] = scale, tr[1..9] = R (row-major, after the Euler round trip), tr[10..12] = t, tr[13] = mean distance of the pairs,
// tr[14] = 1 when the transform is finite
__global__ void icp_transform_kernel(const double *__restrict__ part, double n, double c0x, double c0y, double c0z, int similarity,
                                     double *__restrict__ tr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[kSums];
    for (int q = 0; q < kSums; ++q) {
        double a = 0.0;
        for (int b = 0; b < kIcpBlocks; ++b) a += part[(int64_t)b * kSums + q];  // fixed order
        s[q] = a;
    }
    const double c0[3] = {c0x, c0y, c0z};
    double mux[3], muy[3];
    for (int a = 0; a < 3; ++a) {
        mux[a] = s[a] / n;
        muy[a] = s[3 + a] / n;
    }
    double Sxy[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Sxy[a * 3 + b] = s[6 + a * 3 + b] / n - muy[a] * mux[b];
    const double sig2x = s[15] / n - (mux[0] * mux[0] + mux[1] * mux[1] + mux[2] * mux[2]);
    double R[9], trace_ds = 0.0;
    if (!polar3_rotation(Sxy, R, &trace_ds)) {
        double U[9], D[3], V[9];
        svd3(Sxy, U, D, V);
        const double det = Sxy[0] * (Sxy[4] * Sxy[8] - Sxy[5] * Sxy[7]) - Sxy[1] * (Sxy[3] * Sxy[8] - Sxy[5] * Sxy[6]) +
                           Sxy[2] * (Sxy[3] * Sxy[7] - Sxy[4] * Sxy[6]);
        const double s3 = det < 0 ? -1.0 : 1.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) R[a * 3 + b] = U[a * 3] * V[b * 3] + U[a * 3 + 1] * V[b * 3 + 1] + s3 * U[a * 3 + 2] * V[b * 3 + 2];
        trace_ds = D[0] + D[1] + s3 * D[2];
    }
    const double c = similarity ? trace_ds / sig2x : 1.0;
    // the registration result carries its rotation as Euler angles (rigid3DLandmarkRegistration builds Rotation3D from them)
    double e[3], Re[9];
    rot_to_euler(R, e);
    euler_to_rot(e, Re);
    double mxa[3], mya[3];
    for (int a = 0; a < 3; ++a) {
        mxa[a] = mux[a] + c0[a];
        mya[a] = muy[a] + c0[a];
    }
    tr[0] = c;
    bool fin = fabs(c) <= 1.79769313486231570815e308;
    for (int q = 0; q < 9; ++q) {
        tr[1 + q] = Re[q];
        fin = fin && fabs(Re[q]) <= 1.79769313486231570815e308;
    }
    for (int a = 0; a < 3; ++a) tr[10 + a] = mya[a] - c * (R[a * 3] * mxa[0] + R[a * 3 + 1] * mxa[1] + R[a * 3 + 2] * mxa[2]);
    tr[13] = s[16] / n;
    tr[14] = fin ? 1.0 : 0.0;
}

// p <- s R p + t   (TranslationAfterScalingAfterRotation about the origin; s = 1 for the rigid registrator)
__global__ __launch_bounds__(256) void icp_apply_kernel(int64_t n, double *__restrict__ px, double *__restrict__ py,
                                                        double *__restrict__ pz, const double *__restrict__ tr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double s = tr[0], x = px[i], y = py[i], z = pz[i];
    px[i] = s * (tr[1] * x + tr[2] * y + tr[3] * z) + tr[10];
    py[i] = s * (tr[4] * x + tr[5] * y + tr[6] * z) + tr[11];
    pz[i] = s * (tr[7] * x + tr[8] * y + tr[9] * z) + tr[12];
}

int check(gingr_ctx *ctx) {
    HIP_TRY(ctx, hipGetLastError());
    return GINGR_OK;
}

}  // namespace

extern "C" {

int gingr_rigid_icp_create(gingr_ctx *ctx, int32_t kind, int64_t M, const double *moving_xyz, int64_t N, const double *target_xyz,
                           gingr_rigid_icp **out) {
    if (!ctx || !out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if ((kind != 0 && kind != 1) || M < 1 || N < 1 || N > INT32_MAX || !moving_xyz || !target_xyz)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "rigid_icp_create: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gingr_rigid_icp *h = new gingr_rigid_icp();
    h->ctx = ctx, h->kind = kind, h->M = M, h->N = N;
    auto fail = [&](int rc) {
        delete h;
        return rc;
    };
    const size_t big = (size_t)3 * (M > N ? M : N) * sizeof(double);
    if (h->stage.alloc(big) != hipSuccess || h->tpl.alloc((size_t)3 * M * sizeof(double)) != hipSuccess ||
        h->tgt.alloc((size_t)3 * N * sizeof(double)) != hipSuccess ||
        h->torig.alloc((size_t)N * sizeof(int32_t)) != hipSuccess ||
        h->tboxes.alloc((size_t)30 * ceil_div(N, 256) * sizeof(double)) != hipSuccess ||
        h->ws.alloc((size_t)nn_ws_bytes(M, N)) != hipSuccess || h->idx.alloc((size_t)M * sizeof(int32_t)) != hipSuccess ||
        h->d2.alloc((size_t)M * sizeof(double)) != hipSuccess || h->part.alloc((size_t)kIcpBlocks * kSums * sizeof(double)) != hipSuccess ||
        h->tr.alloc(16 * sizeof(double)) != hipSuccess)
        return fail(gingr_set_error(ctx, GINGR_ERR_HIP, "rigid_icp_create: out of device memory"));
    for (int64_t i = 0; i < M; ++i)
        for (int d = 0; d < 3; ++d) h->c0[d] += moving_xyz[3 * i + d];
    for (int d = 0; d < 3; ++d) h->c0[d] /= (double)M;
    // template as an SoA cloud; the target in Morton order with its tile boxes for the nearest-first pruning of the closest-point
    // kernel (which reports POSITIONS in that order and breaks exact ties by the lowest original index)
    (void)hipMemcpyAsync(h->stage.p, moving_xyz, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    launch_aos_to_soa(ctx, h->stage.as<double>(), M, h->tpl.as<double>());
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpyAsync(h->stage.p, target_xyz, (size_t)3 * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    std::vector<int32_t> order;
    morton_order(target_xyz, N, order);
    (void)hipMemcpyAsync(h->torig.p, order.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    launch_aos_to_soa(ctx, h->stage.as<double>(), N, h->tgt.as<double>(), h->torig.as<int32_t>());
    const double *t = h->tgt.as<double>();
    launch_tile_bbox(ctx, Cloud{t, t + N, t + 2 * N, N}, h->tboxes.as<double>());
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(gingr_set_error(ctx, GINGR_ERR_HIP, "rigid_icp_create: set-up kernels failed"));
    if (nn_grid_build(ctx, target_xyz, N, order.data(), M, &h->tgrid) != GINGR_OK) return fail(GINGR_ERR_HIP);  // (message set)
    *out = h;
    return GINGR_OK;
}

void gingr_rigid_icp_destroy(gingr_rigid_icp *h) {
    if (!h) return;
    if (h->ctx) (void)hipSetDevice(h->ctx->device);
    delete h;
}

/* n_iterations of RigidICP.Iteration; distances[k] (nullable, n_iterations entries) = the mean closest-point distance the k-th
 * iteration measured BEFORE it moved the template (RigidICP.scala:79-82).  Synchronises. */
int gingr_rigid_icp_iterate(gingr_rigid_icp *h, int32_t n_iterations, double *distances) {
    if (!h) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = h->ctx;
    if (n_iterations < 0) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "rigid_icp_iterate: negative iteration count");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = h->M, N = h->N;
    double *p = h->tpl.as<double>();
    const double *t = h->tgt.as<double>();
    const Cloud cp{p, p + M, p + 2 * M, M}, ct{t, t + N, t + 2 * N, N};
    for (int32_t k = 0; k < n_iterations; ++k) {
        const int32_t *warm = h->warm ? h->idx.as<int32_t>() : nullptr;
        if (ctx->nn_grid && h->tgrid.ready && ctx->cull) {  // grid search, then the masked tile scan for what it flagged (fitter.hip, ICP)
            if (!launch_nn_grid(ctx, cp, ct, h->torig.as<int32_t>(), h->tgrid, warm, h->idx.as<int32_t>(), h->d2.as<double>()))
                launch_nn(ctx, cp, ct, h->torig.as<int32_t>(), h->tboxes.as<double>(), h->ws.p, h->idx.as<int32_t>(), h->d2.as<double>(),
                      h->idx.as<int32_t>(), h->tgrid.flag, h->tgrid.cur_nflag());
        } else {
            launch_nn(ctx, cp, ct, h->torig.as<int32_t>(), h->tboxes.as<double>(), h->ws.p, h->idx.as<int32_t>(), h->d2.as<double>(), warm);
        }
        h->warm = true;  // the next iteration starts every query from this one's match
        hipLaunchKernelGGL(icp_sums_kernel, dim3(kIcpBlocks), dim3(256), 0, ctx->stream, cp, ct, h->idx.as<int32_t>(), h->d2.as<double>(),
                           h->c0[0], h->c0[1], h->c0[2], h->part.as<double>());
        hipLaunchKernelGGL(icp_transform_kernel, dim3(1), dim3(64), 0, ctx->stream, h->part.as<double>(), (double)M, h->c0[0], h->c0[1],
                           h->c0[2], (int)h->kind, h->tr.as<double>());
        hipLaunchKernelGGL(icp_apply_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, M, p, p + M, p + 2 * M,
                           h->tr.as<double>());
        GINGR_TRY(check(ctx));
        double res[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(res, h->tr.as<double>() + 13, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (res[1] != 1.0) return gingr_set_error(ctx, GINGR_ERR_NONFINITE, "rigid_icp_iterate: non-finite transform");
        h->last_distance = res[0];
        if (distances) distances[k] = res[0];
    }
    return GINGR_OK;
}

/* current template points [3M] (nullable) and the LAST iteration's transform {s, R[9] row-major, t[3]} (nullable) */
int gingr_rigid_icp_get(gingr_rigid_icp *h, double *points_xyz, double *transform13) {
    if (!h) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = h->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (points_xyz) {
        launch_soa_to_aos(ctx, h->tpl.as<double>(), h->M, h->stage.as<double>());
        HIP_TRY(ctx, hipMemcpyAsync(points_xyz, h->stage.p, (size_t)3 * h->M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (transform13) HIP_TRY(ctx, hipMemcpyAsync(transform13, h->tr.p, 13 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GINGR_OK;
}

/* replace the template points (RigidICP.Iteration is called with the caller's current template) */
int gingr_rigid_icp_set(gingr_rigid_icp *h, const double *points_xyz) {
    if (!h || !points_xyz) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = h->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(h->stage.p, points_xyz, (size_t)3 * h->M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, h->stage.as<double>(), h->M, h->tpl.as<double>());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GINGR_OK;
}

}  // extern "C"

// Context, error handling, timing hooks and the stateless all-pairs operators of the C ABI (include/gingr_hip.h).
#include "common.h"

#include <map>
#include <mutex>
void set_dynamic_lds(const void *func, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> limit;  // (device, function) -> bytes already granted
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    int &have = limit[std::make_pair(dev, func)];
    if (have >= bytes) return;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess)
        have = bytes;
    else
        (void)hipGetLastError();
}


int gingr_set_error(gingr_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

// ------------------------------------------------------------------------------------------------ timing hooks
TimerScope::TimerScope(gingr_ctx *c, int w) : ctx(c), which(w) {
    if (!ctx->timing) return;
    auto get = [&]() {
        hipEvent_t e = nullptr;
        if (!ctx->pool.empty()) {
            e = ctx->pool.back();
            ctx->pool.pop_back();
        } else {
            (void)hipEventCreate(&e);
        }
        return e;
    };
    a = get();
    b = get();
    (void)hipEventRecord(a, ctx->stream);
}

void TimerScope::stop() {
    if (!a) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->spans.push_back({a, b, which});
    a = nullptr;
}

TimerScope::~TimerScope() { stop(); }

static void timing_resolve(gingr_ctx *ctx) {
    for (auto &s : ctx->spans) {
        (void)hipEventSynchronize(s.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            ctx->t_ms[s.which] += ms;
            ctx->t_n[s.which] += 1;
        }
        ctx->pool.push_back(s.a);
        ctx->pool.push_back(s.b);
    }
    ctx->spans.clear();
}

extern "C" {

int gingr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *gingr_build_info(void) { return "libgingr_hip gfx950 (CDNA4) f64; built " __DATE__ " " __TIME__; }

int gingr_ctx_create(int device, gingr_ctx **out) {
    if (!out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GINGR_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return GINGR_ERR_BAD_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return GINGR_ERR_HIP;
    gingr_ctx *ctx = new gingr_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return GINGR_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    void *hp = nullptr, *dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
        memset(hp, 0, 64);
        if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            ctx->regime_host = static_cast<int32_t *>(hp);
            ctx->regime_dev = static_cast<int32_t *>(dp);
        } else {
            (void)hipHostFree(hp);
        }
    }
    (void)hipGetLastError();
    *out = ctx;
    return GINGR_OK;
}

void gingr_ctx_destroy(gingr_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    timing_resolve(ctx);
    gingr_ctx_rccl_release(ctx);
    for (auto e : ctx->pool) (void)hipEventDestroy(e);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->nn_tests) (void)hipFree(ctx->nn_tests);
    if (ctx->regime_host) (void)hipHostFree(ctx->regime_host);
    if (ctx->side_stream) {
        (void)hipStreamSynchronize(ctx->side_stream);
        (void)hipStreamDestroy(ctx->side_stream);
    }
    for (hipEvent_t e : ctx->split_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *gingr_last_error(const gingr_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int gingr_ctx_set_stream(gingr_ctx *ctx, void *hip_stream) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return GINGR_OK;
}

void *gingr_ctx_get_stream(gingr_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

int gingr_ctx_synchronize(gingr_ctx *ctx) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GINGR_OK;
}

int gingr_ctx_set_option(gingr_ctx *ctx, int32_t option, int32_t value) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    switch (option) {
        case GINGR_OPT_CULL: ctx->cull = value != 0; return GINGR_OK;
        case GINGR_OPT_FINE_CULL: ctx->fine_override = value < 0 ? -1 : (value != 0); return GINGR_OK;
        case GINGR_OPT_NN_GRID: ctx->nn_grid = value < 0 ? 0 : (value > 2 ? 2 : value); return GINGR_OK;
        case GINGR_OPT_TRI_GRID: ctx->tri_grid = value < 0 ? 0 : (value > 2 ? 2 : value); return GINGR_OK;
        case GINGR_OPT_SPLIT_EXCHANGE: ctx->split_exchange = value != 0; return GINGR_OK;
        case GINGR_OPT_GRAM_DOWNDATE: ctx->gram_downdate = value < 0 ? -1 : (value != 0); return GINGR_OK;
        default: return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "ctx_set_option: unknown option %d", option);
    }
}

int gingr_ctx_get_option(gingr_ctx *ctx, int32_t option, int32_t *value) {
    if (!ctx || !value) return GINGR_ERR_BAD_ARGUMENT;
    switch (option) {
        case GINGR_OPT_CULL: *value = ctx->cull; return GINGR_OK;
        case GINGR_OPT_FINE_CULL: *value = ctx->fine_override; return GINGR_OK;
        case GINGR_OPT_NN_GRID: *value = ctx->nn_grid; return GINGR_OK;
        case GINGR_OPT_TRI_GRID: *value = ctx->tri_grid; return GINGR_OK;
        case GINGR_OPT_SPLIT_EXCHANGE: *value = ctx->split_exchange; return GINGR_OK;
        case GINGR_OPT_GRAM_DOWNDATE: *value = ctx->gram_downdate; return GINGR_OK;
        default: return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "ctx_get_option: unknown option %d", option);
    }
}

int gingr_ctx_timing_enable(gingr_ctx *ctx, int32_t enable) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    ctx->timing = enable != 0;
    return GINGR_OK;
}

int gingr_ctx_timing_read(gingr_ctx *ctx, int32_t which, double *total_ms, int64_t *launches) {
    if (!ctx || which < 0 || which >= GINGR_TIMERS) return GINGR_ERR_BAD_ARGUMENT;
    timing_resolve(ctx);
    if (total_ms) *total_ms = ctx->t_ms[which];
    if (launches) *launches = ctx->t_n[which];
    return GINGR_OK;
}

int gingr_ctx_nn_counting(gingr_ctx *ctx, int32_t enable) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (enable && !ctx->nn_tests) {
        void *p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, sizeof(unsigned long long)));
        ctx->nn_tests = static_cast<unsigned long long *>(p);
    }
    if (ctx->nn_tests) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->nn_tests, 0, sizeof(unsigned long long), ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (!enable && ctx->nn_tests) {
        (void)hipFree(ctx->nn_tests);
        ctx->nn_tests = nullptr;
    }
    return GINGR_OK;
}

int gingr_ctx_nn_tests(gingr_ctx *ctx, int64_t *tests) {
    if (!ctx || !tests) return GINGR_ERR_BAD_ARGUMENT;
    *tests = 0;
    if (!ctx->nn_tests) return gingr_set_error(ctx, GINGR_ERR_STATE, "nn_tests: counting is off (gingr_ctx_nn_counting)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpy(&v, ctx->nn_tests, sizeof(v), hipMemcpyDeviceToHost));
    *tests = (int64_t)v;
    return GINGR_OK;
}

int gingr_ctx_timing_reset(gingr_ctx *ctx) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    timing_resolve(ctx);
    for (int i = 0; i < GINGR_TIMERS; ++i) {
        ctx->t_ms[i] = 0;
        ctx->t_n[i] = 0;
    }
    return GINGR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ helpers
namespace {

// upload interleaved host points into a freshly allocated SoA device cloud
int upload_cloud(gingr_ctx *ctx, int64_t n, const double *host_xyz, DevBuf &staging, DevBuf &soa, Cloud *out) {
    HIP_TRY(ctx, staging.alloc((size_t)n * 3 * sizeof(double)));
    HIP_TRY(ctx, soa.alloc((size_t)n * 3 * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(staging.p, host_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, staging.as<double>(), n, soa.as<double>());
    double *s = soa.as<double>();
    *out = Cloud{s, s + n, s + 2 * n, n};
    return GINGR_OK;
}

int check_launch(gingr_ctx *ctx) {
    HIP_TRY(ctx, hipGetLastError());
    return GINGR_OK;
}

}  // namespace

extern "C" {

int gingr_cpd_stats(gingr_ctx *ctx, int64_t M, const double *fit, int64_t N, const double *target, double sigma2,
                    double w, double *den, double *P1, double *PX, double *Pt1, double *scalars) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if (M < 1 || N < 1 || !fit || !target) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "cpd_stats: M, N must be >= 1");
    if (!(w >= 0.0 && w < 1.0)) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "cpd_stats: w must be in [0,1)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf sf, st, df, dt, dws, dden, dinv, dpt1, dp1, dpx, dsc, daos, dpart;
    Cloud cf, ct;
    GINGR_TRY(upload_cloud(ctx, M, fit, sf, df, &cf));
    GINGR_TRY(upload_cloud(ctx, N, target, st, dt, &ct));
    const int64_t ws1 = cpd_colsum_ws_doubles(M, N), ws2 = cpd_rowstats_ws_doubles(M, N);
    HIP_TRY(ctx, dws.alloc((size_t)(ws1 > ws2 ? ws1 : ws2) * sizeof(double)));
    HIP_TRY(ctx, dden.alloc(N * sizeof(double)));
    HIP_TRY(ctx, dinv.alloc(N * sizeof(double)));
    HIP_TRY(ctx, dpt1.alloc(N * sizeof(double)));
    HIP_TRY(ctx, dp1.alloc(M * sizeof(double)));
    HIP_TRY(ctx, dpx.alloc(3 * M * sizeof(double)));
    HIP_TRY(ctx, daos.alloc(3 * M * sizeof(double)));
    HIP_TRY(ctx, dsc.alloc(16 * sizeof(double)));
    HIP_TRY(ctx, dpart.alloc(GINGR_SCALAR_PART * sizeof(double)));
    HIP_TRY(ctx, hipMemsetAsync(dsc.p, 0, 16 * sizeof(double), ctx->stream));
    double *sc = dsc.as<double>();
    double *s2dev = sc + 8;
    HIP_TRY(ctx, hipMemcpyAsync(s2dev, &sigma2, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    DevBuf daux;
    HIP_TRY(ctx, daux.alloc(GINGR_AUX * sizeof(double)));
    double *absmax = daux.as<double>();
    launch_cloud_centroid(ctx, ct, absmax + 2);
    launch_cloud_absmax(ctx, ct, absmax + 2, absmax);
    launch_cloud_absmax(ctx, cf, absmax + 2, absmax + 1);
    launch_cpd_colsum(ctx, cf, ct, s2dev, absmax, nullptr, dws.as<double>(), dden.as<double>());
    launch_cpd_den_finalize(ctx, ct, s2dev, w, M, dden.as<double>(), dinv.as<double>(), dpt1.as<double>(), nullptr, dpart.as<double>(), sc);
    launch_cpd_rowstats(ctx, cf, ct, s2dev, absmax, dinv.as<double>(), nullptr, nullptr, dws.as<double>(), dp1.as<double>(), dpx.as<double>(), dpart.as<double>(), sc);
    GINGR_TRY(check_launch(ctx));
    launch_soa_to_aos(ctx, dpx.as<double>(), M, daos.as<double>());
    double hsc[8];
    HIP_TRY(ctx, hipMemcpyAsync(hsc, sc, sizeof(hsc), hipMemcpyDeviceToHost, ctx->stream));
    if (den) HIP_TRY(ctx, hipMemcpyAsync(den, dden.p, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (Pt1) HIP_TRY(ctx, hipMemcpyAsync(Pt1, dpt1.p, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (P1) HIP_TRY(ctx, hipMemcpyAsync(P1, dp1.p, M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (PX) HIP_TRY(ctx, hipMemcpyAsync(PX, daos.p, 3 * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (scalars) {
        // sigma2' = (xPx - 2 trPXY + yPy) / (3 Np)       CPD.scala:145
        scalars[0] = hsc[0];
        scalars[1] = hsc[1];
        scalars[2] = hsc[2];
        scalars[3] = hsc[3];
        scalars[4] = (hsc[1] - 2 * hsc[2] + hsc[3]) / (hsc[0] * 3.0);
        scalars[5] = hsc[5];
    }
    return GINGR_OK;
}

int gingr_cpd_initial_sigma2(gingr_ctx *ctx, int64_t M, const double *ref, int64_t N, const double *target,
                             double *sigma2_out) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if (M < 1 || N < 1 || !ref || !target || !sigma2_out)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "cpd_initial_sigma2: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf sa, sb, da, db, dws, dout;
    Cloud ca, cb;
    GINGR_TRY(upload_cloud(ctx, M, ref, sa, da, &ca));
    GINGR_TRY(upload_cloud(ctx, N, target, sb, db, &cb));
    HIP_TRY(ctx, dws.alloc((size_t)sumsq_pairs_ws_doubles(M) * sizeof(double)));
    HIP_TRY(ctx, dout.alloc(sizeof(double)));
    launch_sumsq_pairs(ctx, ca, cb, dws.as<double>(), dout.as<double>());
    GINGR_TRY(check_launch(ctx));
    double tot = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&tot, dout.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *sigma2_out = tot / (3.0 * (double)N * (double)M);
    return GINGR_OK;
}

int gingr_nn(gingr_ctx *ctx, int64_t M, const double *query, int64_t N, const double *target, int32_t *idx, double *d2,
             double *mean_distance) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if (M < 1 || N < 1 || !query || !target) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "nn: M, N must be >= 1");
    if (N > INT32_MAX) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "nn: N exceeds int32 index range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf sq, st, dq, dt, dws, didx, dd2, dperm, dqperm, dboxes;
    Cloud cq, ct;
    // From a few thousand points on both clouds are put into the spatial (k-d leaf) order of the fitter on the way to the device
    // (round 4; the host has the arrays anyway): the tile scan then prunes by the 256-target tile boxes and their quarter boxes against
    // spatially coherent waves of queries, exactly as inside a registration, instead of testing all pairs (5 000 x 5 000: 25 M distance
    // tests in one round of ~80 short workgroups, 29.7 us).  GINGR_OPT_NN_GRID additionally searches a uniform grid of the targets
    // first (nn_grid.hip) and leaves the scan only the queries the grid cannot certify -- the fitter's warm-started search; from a
    // cold start it is slower than the pruned scan (64 us at 5 000 x 5 000), so the stateless call does not use it unless the option
    // is set to 2.  Same distances, same lowest-original-index tie rule in every variant: the indices are bit-identical
    // (tests/test_gpu_nn_grid.py compares them; GINGR_OPT_CULL = 0 is the plain scan of all pairs).
    // Up to 2^26 pairs (8 000 x 8 000) nothing of that pays: nn_small_kernel tests all pairs straight from the caller's order with
    // wave-uniform target loads in ~4 000 single-wave workgroups (5 000 x 5 000: ~10 us for both of its launches against 30 us for
    // either scan).  GINGR_OPT_NN_GRID = 2 and GINGR_OPT_CULL = 0 still select the other variants (timing comparisons, tests).
    const bool small = ctx->cull && ctx->nn_grid != 2 && nn_small_applies(M, N);
    const bool ordered = !small && ctx->cull && N >= 2048 && M >= 256;
    const bool use_grid = ordered && ctx->nn_grid == 2;
    std::vector<int32_t> perm, qperm;
    NNGrid grid;
    struct GridGuard {  // the per-call grid's device arrays go away on every way out (the early returns of HIP_TRY included) ...
        gingr_ctx *ctx;
        NNGrid *g;
        ~GridGuard() {
            if (g->ready || g->cell_start || g->pts || g->flag || g->nflag) {
                (void)hipStreamSynchronize(ctx->stream);  // ... once nothing enqueued can still read them
                nn_grid_free(g);
            }
        }
    } grid_guard{ctx, &grid};
    if (ordered) {
        morton_order(target, N, perm);
        morton_order(query, M, qperm);
        HIP_TRY(ctx, st.alloc((size_t)N * 3 * sizeof(double)));
        HIP_TRY(ctx, dt.alloc((size_t)N * 3 * sizeof(double)));
        HIP_TRY(ctx, sq.alloc((size_t)M * 3 * sizeof(double)));
        HIP_TRY(ctx, dq.alloc((size_t)M * 3 * sizeof(double)));
        HIP_TRY(ctx, dperm.alloc((size_t)N * sizeof(int32_t)));
        HIP_TRY(ctx, dqperm.alloc((size_t)M * sizeof(int32_t)));
        HIP_TRY(ctx, dboxes.alloc((size_t)ceil_div(N, 256) * 30 * sizeof(double)));
        HIP_TRY(ctx, hipMemcpyAsync(st.p, target, (size_t)N * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(sq.p, query, (size_t)M * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dperm.p, perm.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dqperm.p, qperm.data(), (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        launch_aos_to_soa(ctx, st.as<double>(), N, dt.as<double>(), dperm.as<int32_t>());
        launch_aos_to_soa(ctx, sq.as<double>(), M, dq.as<double>(), dqperm.as<int32_t>());
        double *sp = dt.as<double>(), *qp = dq.as<double>();
        ct = Cloud{sp, sp + N, sp + 2 * N, N};
        cq = Cloud{qp, qp + M, qp + 2 * M, M};
        launch_tile_bbox(ctx, ct, dboxes.as<double>());
        if (use_grid) {
            GINGR_TRY(nn_grid_build(ctx, target, N, perm.data(), M, &grid));
        }
    } else {
        GINGR_TRY(upload_cloud(ctx, M, query, sq, dq, &cq));
        GINGR_TRY(upload_cloud(ctx, N, target, st, dt, &ct));
    }
    HIP_TRY(ctx, dws.alloc((size_t)(small ? nn_small_ws_bytes(M, N) : nn_ws_bytes(M, N))));
    HIP_TRY(ctx, didx.alloc(M * sizeof(int32_t)));
    HIP_TRY(ctx, dd2.alloc(M * sizeof(double)));
    if (small) {
        launch_nn_small(ctx, cq, ct, dws.p, didx.as<int32_t>(), dd2.as<double>());
    } else if (grid.ready) {
        if (!launch_nn_grid(ctx, cq, ct, dperm.as<int32_t>(), grid, nullptr, didx.as<int32_t>(), dd2.as<double>()))
            launch_nn(ctx, cq, ct, dperm.as<int32_t>(), dboxes.as<double>(), dws.p, didx.as<int32_t>(), dd2.as<double>(), nullptr, grid.flag,
                  grid.cur_nflag());
    } else if (ordered) {
        launch_nn(ctx, cq, ct, dperm.as<int32_t>(), dboxes.as<double>(), dws.p, didx.as<int32_t>(), dd2.as<double>());
    } else {
        launch_nn(ctx, cq, ct, nullptr, nullptr, dws.p, didx.as<int32_t>(), dd2.as<double>());
    }
    GINGR_TRY(check_launch(ctx));
    std::vector<double> hd2((size_t)M);
    std::vector<int32_t> hidx((size_t)M);
    HIP_TRY(ctx, hipMemcpyAsync(hidx.data(), didx.p, M * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hd2.data(), dd2.p, M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ordered) {  // device positions of queries and targets -> the caller's numbering
        std::vector<double> t2((size_t)M);
        std::vector<int32_t> ti((size_t)M);
        for (int64_t s2 = 0; s2 < M; ++s2) {
            const int32_t row = qperm[(size_t)s2], pos = hidx[(size_t)s2];
            ti[(size_t)row] = (pos >= 0 && pos < N) ? perm[(size_t)pos] : -1;
            t2[(size_t)row] = hd2[(size_t)s2];
        }
        hidx.swap(ti);
        hd2.swap(t2);
    }
    if (idx) memcpy(idx, hidx.data(), M * sizeof(int32_t));
    if (d2) memcpy(d2, hd2.data(), M * sizeof(double));
    if (mean_distance) {
        // distance += (p - closestPoint).norm in index order; / numberOfPoints    ClosestPointRegistrator.scala:143,146
        double s = 0.0;
        for (int64_t i = 0; i < M; ++i) s += sqrt(hd2[(size_t)i]);
        *mean_distance = s / (double)M;
    }
    return GINGR_OK;
}

int gingr_gauss_block(gingr_ctx *ctx, int64_t na, const double *A, int64_t nb, const double *B, double sigma,
                      double scaling, double *out) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if (na < 1 || nb < 1 || !A || !B || !out || !(sigma > 0))
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gauss_block: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf sa, sb, da, db, dout;
    Cloud ca, cb;
    GINGR_TRY(upload_cloud(ctx, na, A, sa, da, &ca));
    GINGR_TRY(upload_cloud(ctx, nb, B, sb, db, &cb));
    HIP_TRY(ctx, dout.alloc((size_t)na * nb * sizeof(double)));
    launch_gauss_block(ctx, ca, cb, sigma, scaling, dout.as<double>());
    GINGR_TRY(check_launch(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, (size_t)na * nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GINGR_OK;
}

}  // extern "C"

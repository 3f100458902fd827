// Device group: the row-sharded GiNGR update across several GPUs of one node, driven from ONE host process through the C ABI
// (gingr_group_* in include/gingr_hip.h) -- what SURVEY.md section 8b calls gingr_group_create: a JVM / C host has no
// torch.distributed, so the exchange lives in the library.
//
// Partitioning (BASELINE.json north_star): contiguous row shards of the reference / fit points, one
// gingr_ctx + gingr_model + gingr_fitter per device; target cloud and all r-sized state replicated.  Per iteration the two
// exchange segments of fitter.hip (CPD column sums den[N]; Gram + right-hand side + sigma2 sums) are summed across shards
// by a ONE-SHOT all-reduce over peer pointers: every shard writes its partial segment into its own send buffer, records an
// event, waits for the events of all peers and then sums the n send buffers -- its own and, through the xGMI peer
// mapping, the remote ones -- in rank order into its exchange buffer.  Both messages are small (400 KB and 100 KB at
// 50k points, rank 100), so the exchange is latency bound and one hop beats a ring; every shard adds the
// same numbers in the same order, so the replicated r x r solve sees bit-identical inputs on every device and the results do
// not depend on timing.
//
// Host side: one worker thread per device enqueues that device's kernels (one thread serving 8 devices would be launch
// bound: ~20 launches x 8 devices per iteration against ~0.4 ms of device time); the workers meet at a host barrier once per
// exchange, because hipStreamWaitEvent only orders against an event that has already been RECORDED.  Nothing waits for the
// GPU inside an update: n iterations are enqueued back to back.
//
// Send buffers are double buffered by iteration parity: a shard may run ahead of a peer by less than one exchange (it waits
// for the peer's NEXT event before it can finish its own next all-reduce), so the buffer it overwrites two iterations later is
// no longer being read.  They are fine-grained device allocations (coherent across devices) and the events release at system
// scope.  Several shards may live on ONE device (devices = {0, 0}): that is how the protocol is tested on a one-GPU box.
#include "gp.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace {

constexpr int kMaxGroup = 16;

struct PeerPtrs {
    const double *p[kMaxGroup];
};

// out[i] = p[0][i] + p[1][i] + ... in rank order (left to right); two doubles per thread (16-byte accesses)
__global__ __launch_bounds__(256) void peer_sum_kernel(double *__restrict__ out, PeerPtrs src, int n, int64_t count) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < count) {
        double2 s = *reinterpret_cast<const double2 *>(src.p[0] + i);
        for (int q = 1; q < n; ++q) {
            const double2 v = *reinterpret_cast<const double2 *>(src.p[q] + i);
            s.x += v.x;
            s.y += v.y;
        }
        *reinterpret_cast<double2 *>(out + i) = s;
    } else if (i < count) {
        double s = src.p[0][i];
        for (int q = 1; q < n; ++q) s += src.p[q][i];
        out[i] = s;
    }
}

// The gather of the fit (exchange segment GINGR_SEGMENT_FULLFIT): every shard has written its rows into ITS send buffer ([3][M_total]
// planes, original point order); out[d][g] is read from the buffer of the shard that owns row g -- one remote read per element, not a
// sum over n buffers of which n - 1 hold a zero there (round 4).  bounds.b[q] = first row of shard q, b[n] = M_total.
struct RowBounds {
    int64_t b[kMaxGroup + 1];
};
__global__ __launch_bounds__(256) void peer_gather_rows_kernel(double *__restrict__ out, PeerPtrs src, RowBounds bounds, int n, int64_t M_total) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= M_total) return;
    int q = 0;
    while (q + 1 < n && g >= bounds.b[q + 1]) ++q;
    const double *p = src.p[q];
#pragma unroll
    for (int d = 0; d < 3; ++d) out[d * M_total + g] = p[d * M_total + g];
}

// restores the calling thread's current device: the group's host-side set-up selects its devices one after the other, and a JVM / C /
// torch host must find its own device selection untouched when the call returns
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() {
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
        (void)hipGetLastError();
    }
    ~DeviceGuard() {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

struct SpinBarrier {
    std::atomic<int> count{0}, gen{0};
    int n = 1;
    void wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_acq_rel);
        } else {
            while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
        }
    }
};

}  // namespace

struct gingr_group {
    int n = 0;
    std::vector<int> dev;
    std::vector<gingr_ctx *> ctx;
    std::vector<gingr_model *> model;
    std::vector<gingr_fitter *> fit;
    std::vector<int64_t> begin, end;
    int64_t M_total = 0, N = 0;
    int32_t rank = 0;
    // exchange
    std::vector<double *> send[2];                              // [parity][shard]: partial segments, layout of the fitter's xch
    std::vector<hipEvent_t> ready[2][GINGR_NUM_SEGMENTS];       // [parity][segment][shard]
    std::vector<double *> xch;                                  // the fitters' exchange buffers
    int64_t off[GINGR_NUM_SEGMENTS] = {0, 0}, cnt[GINGR_NUM_SEGMENTS] = {0, 0};
    // the gather of a sharded surface update (exchange segment GINGR_SEGMENT_FULLFIT): contributions [3 M_total] per shard and parity
    std::vector<double *> sendfit[2];
    std::vector<hipEvent_t> readyfit[2];
    // the per-template-vertex sums of the sharded reversed direction (GINGR_SEGMENT_REVSUM): contributions [4 M_total] per shard and parity
    std::vector<double *> sendrev[2];
    std::vector<hipEvent_t> readyrev[2];
    bool meshes = false;
    bool reversed = false;  // reversed correspondence direction (gingr_group_set_correspondence_direction): gathers the fit like flavour 2
    int64_t iteration = 0;                                      // parity of the send buffers
    bool fine_grained = false;                                  // the send buffers are fine-grained device allocations
    int distinct_devices = 1;
    SpinBarrier bar;
    // workers
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<int(int)> job;
    uint64_t generation = 0;
    int pending = 0;
    bool quit = false;
    std::vector<int> status;
    char err[640] = {0};

    int run(const std::function<int(int)> &fn) {  // fn(shard) on every worker; first non-zero status wins
        {
            std::unique_lock<std::mutex> lk(mu);
            job = fn;
            pending = n;
            ++generation;
        }
        cv_job.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        for (int r = 0; r < n; ++r)
            if (status[(size_t)r] != GINGR_OK) {
                snprintf(err, sizeof(err), "shard %d (device %d): %s", r, dev[(size_t)r], gingr_last_error(ctx[(size_t)r]));
                return status[(size_t)r];
            }
        return GINGR_OK;
    }
};

namespace {

void worker_main(gingr_group *g, int r) {
    (void)hipSetDevice(g->dev[(size_t)r]);
    uint64_t seen = 0;
    for (;;) {
        std::function<int(int)> fn;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->quit || g->generation != seen; });
            if (g->quit) return;
            seen = g->generation;
            fn = g->job;
        }
        const int rc = fn(r);
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->status[(size_t)r] = rc;
            if (--g->pending == 0) g->cv_done.notify_all();
        }
    }
}

int group_fail(gingr_group *g, int code, const char *what) {
    snprintf(g->err, sizeof(g->err), "%s", what);
    return code;
}

void shard_rows(int64_t M, int n, int r, int64_t *b, int64_t *e) {  // the first (M mod n) shards hold one extra row
    const int64_t base = M / n, extra = M % n;
    *b = r * base + (r < extra ? r : extra);
    *e = *b + base + (r < extra ? 1 : 0);
}

void free_exchange(gingr_group *g) {
    DeviceGuard guard;
    for (int p = 0; p < 2; ++p) {
        for (size_t r = 0; r < g->send[p].size(); ++r)
            if (g->send[p][r]) {
                (void)hipSetDevice(g->dev[r]);
                (void)hipFree(g->send[p][r]);
            }
        g->send[p].clear();
        for (int s = 0; s < GINGR_NUM_SEGMENTS; ++s) {
            for (size_t r = 0; r < g->ready[p][s].size(); ++r)
                if (g->ready[p][s][r]) (void)hipEventDestroy(g->ready[p][s][r]);
            g->ready[p][s].clear();
        }
    }
    for (int p = 0; p < 2; ++p) {
        for (size_t r = 0; r < g->sendfit[p].size(); ++r)
            if (g->sendfit[p][r]) {
                (void)hipSetDevice(g->dev[r]);
                (void)hipFree(g->sendfit[p][r]);
            }
        g->sendfit[p].clear();
        for (size_t r = 0; r < g->readyfit[p].size(); ++r)
            if (g->readyfit[p][r]) (void)hipEventDestroy(g->readyfit[p][r]);
        g->readyfit[p].clear();
        for (size_t r = 0; r < g->sendrev[p].size(); ++r)
            if (g->sendrev[p][r]) {
                (void)hipSetDevice(g->dev[r]);
                (void)hipFree(g->sendrev[p][r]);
            }
        g->sendrev[p].clear();
        for (size_t r = 0; r < g->readyrev[p].size(); ++r)
            if (g->readyrev[p][r]) (void)hipEventDestroy(g->readyrev[p][r]);
        g->readyrev[p].clear();
    }
    g->meshes = false;
    g->reversed = false;  // (the fitters forget their direction with their meshes: a new target starts in the forward direction)
    g->xch.clear();
}

void free_fitters(gingr_group *g) {
    for (size_t r = 0; r < g->fit.size(); ++r) {
        if (g->fit[r]) gingr_fitter_destroy(g->fit[r]);
        g->fit[r] = nullptr;
    }
    free_exchange(g);
}

void free_models(gingr_group *g) {
    free_fitters(g);
    for (size_t r = 0; r < g->model.size(); ++r) {
        if (g->model[r]) gingr_model_destroy(g->model[r]);
        g->model[r] = nullptr;
    }
}

// Peer-readable device memory on the CURRENT device.  Fine-grained (coherent for the peers' reads across xGMI).  Plain device memory
// is only acceptable when every shard of the group sits on the same device (logical shards: no peer ever reads across devices);
// with distinct devices a coarse-grained buffer would make the peers' reads non-coherent without any warning, so that is an error.
int alloc_peer_readable(gingr_group *g, size_t bytes, void **out, bool *fine) {
    *out = nullptr;
    if (hipExtMallocWithFlags(out, bytes, hipDeviceMallocFinegrained) == hipSuccess) {
        *fine = true;
        return GINGR_OK;
    }
    (void)hipGetLastError();
    *out = nullptr;
    if (g->distinct_devices > 1)
        return group_fail(g, GINGR_ERR_HIP,
                          "group: fine-grained device memory (hipDeviceMallocFinegrained) is not available; peers on other devices cannot "
                          "read a coarse-grained send buffer coherently");
    if (hipMalloc(out, bytes) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return group_fail(g, GINGR_ERR_HIP, "group: out of device memory (send buffer)");
    }
    *fine = false;
    return GINGR_OK;
}

// after every shard's model exists: sum the one-off basis moments over the shards and finalize.  Same one-shot all-reduce over peer
// pointers as the per-iteration exchange: every shard copies its partial moments into a peer-readable buffer, everybody meets, and
// every shard adds the n buffers in rank order into its own moments (bit-identical on every device, no host copy of the moments).
int finish_models(gingr_group *g) {
    const int n = g->n;
    if (n > 1) {
        DeviceGuard guard;
        int64_t count = 0;
        std::vector<double *> mom((size_t)n, nullptr), tmp((size_t)n, nullptr);
        auto release = [&]() {
            for (int r = 0; r < n; ++r)
                if (tmp[(size_t)r]) {
                    (void)hipSetDevice(g->dev[(size_t)r]);
                    (void)hipFree(tmp[(size_t)r]);
                }
        };
        for (int r = 0; r < n; ++r) {
            void *p = nullptr;
            int64_t c = 0;
            GINGR_TRY(gingr_model_gram_exchange(g->model[(size_t)r], &p, &c));
            if (r == 0) count = c;
            if (c != count) return group_fail(g, GINGR_ERR_STATE, "group: shards disagree on the model rank");
            mom[(size_t)r] = static_cast<double *>(p);
        }
        for (int r = 0; r < n; ++r) {
            void *buf = nullptr;
            bool fine = false;
            if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess) {
                release();
                return group_fail(g, GINGR_ERR_HIP, "group: hipSetDevice failed");
            }
            const int rc = alloc_peer_readable(g, (size_t)count * sizeof(double), &buf, &fine);
            if (rc) {
                release();
                return rc;
            }
            tmp[(size_t)r] = static_cast<double *>(buf);
        }
        // copy (own stream), wait, sum (own stream), wait: a one-off, so the host may wait between the steps
        int rc = g->run([&](int r) {
            gingr_ctx *ctx = g->ctx[(size_t)r];
            if (hipMemcpyAsync(tmp[(size_t)r], mom[(size_t)r], (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                return gingr_set_error(ctx, GINGR_ERR_HIP, "group: moment copy failed");
            return gingr_ctx_synchronize(ctx);
        });
        if (!rc)
            rc = g->run([&](int r) {
                gingr_ctx *ctx = g->ctx[(size_t)r];
                PeerPtrs src;
                for (int q = 0; q < n; ++q) src.p[q] = tmp[(size_t)q];
                hipLaunchKernelGGL(peer_sum_kernel, dim3((unsigned)ceil_div(ceil_div(count, 2), 256)), dim3(256), 0, ctx->stream,
                                   mom[(size_t)r], src, n, count);
                if (hipGetLastError() != hipSuccess) return gingr_set_error(ctx, GINGR_ERR_HIP, "group: moment all-reduce launch failed");
                return gingr_ctx_synchronize(ctx);
            });
        release();
        if (rc) return rc;
        GINGR_TRY(g->run([&](int r) { return gingr_model_finalize(g->ctx[(size_t)r], g->model[(size_t)r]); }));
    }
    g->rank = gingr_model_rank(g->model[0]);
    // one fitter per shard
    g->fit.assign((size_t)n, nullptr);
    return g->run([&](int r) { return gingr_fitter_create(g->ctx[(size_t)r], g->model[(size_t)r], &g->fit[(size_t)r]); });
}

// One-shot all-reduce over peer pointers, called by every worker: the shard has written its contribution to src[r]; it records
// ev[r], everybody meets, the stream waits for the peers' events and one kernel adds the n contributions in rank order into dst.
// `ok` false: the worker still takes part in the barrier (so nobody deadlocks) but enqueues nothing.
int exchange_buffers(gingr_group *g, int r, std::vector<hipEvent_t> &ev, const std::vector<double *> &src_base, int64_t src_off, double *dst,
                     int64_t count, bool ok, bool gather_rows = false) {
    gingr_ctx *ctx = g->ctx[(size_t)r];
    int rc = GINGR_OK;
    if (ok && hipEventRecord(ev[(size_t)r], ctx->stream) != hipSuccess) rc = gingr_set_error(ctx, GINGR_ERR_HIP, "group: hipEventRecord failed");
    g->bar.wait();  // every peer has RECORDED its event (a wait on a never-recorded event would be a no-op)
    if (!ok || rc) return rc;
    PeerPtrs src;
    for (int q = 0; q < g->n; ++q) {
        if (q != r && hipStreamWaitEvent(ctx->stream, ev[(size_t)q], 0) != hipSuccess)
            return gingr_set_error(ctx, GINGR_ERR_HIP, "group: hipStreamWaitEvent failed");
        src.p[q] = src_base[(size_t)q] + src_off;
    }
    if (gather_rows) {  // (count = 3 M_total: every row comes from the one shard that owns it)
        RowBounds rb;
        for (int q = 0; q < g->n; ++q) rb.b[q] = g->begin[(size_t)q];
        rb.b[g->n] = g->M_total;
        hipLaunchKernelGGL(peer_gather_rows_kernel, dim3((unsigned)ceil_div(g->M_total, 256)), dim3(256), 0, ctx->stream, dst, src, rb, g->n, g->M_total);
    } else {
        hipLaunchKernelGGL(peer_sum_kernel, dim3((unsigned)ceil_div(ceil_div(count, 2), 256)), dim3(256), 0, ctx->stream, dst, src, g->n, count);
    }
    if (hipGetLastError() != hipSuccess) return gingr_set_error(ctx, GINGR_ERR_HIP, "group: all-reduce kernel launch failed");
    return GINGR_OK;
}
// all-reduce of exchange segment s (partials in send[parity], sums into the fitters' xch)
int exchange_segment(gingr_group *g, int r, int s, int parity, bool ok) {
    return exchange_buffers(g, r, g->ready[parity][s], g->send[parity], g->off[s], g->xch[(size_t)r] + g->off[s], g->cnt[s], ok);
}
// timer slots of the exchanges (gingr_ctx_timing_read): 6 = segment 0 (column sums), 7 = segment 1 (Gram bundle); the span runs from
// the record of the shard's own event to the end of its sum kernel, i.e. it includes the wait for the slowest peer
int timed_exchange_segment(gingr_group *g, int r, int s, int parity, bool ok) {
    TimerScope ts(g->ctx[(size_t)r], 6 + s);
    return exchange_segment(g, r, s, parity, ok);
}
// the gather of a sharded surface iteration: contributions in sendfit[parity], the full fit into every fitter's own buffer
int exchange_fullfit(gingr_group *g, int r, int parity, bool ok) {
    return exchange_buffers(g, r, g->readyfit[parity], g->sendfit[parity], 0, fitter_fullfit(g->fit[(size_t)r]), 3 * g->M_total, ok, true);
}
// the reversed direction's per-template-vertex sums: contributions in sendrev[parity], the totals into every fitter's own buffer
int exchange_revsum(gingr_group *g, int r, int parity, bool ok) {
    return exchange_buffers(g, r, g->readyrev[parity], g->sendrev[parity], 0, fitter_revsum(g->fit[(size_t)r]), 4 * g->M_total, ok);
}

// flavour 0 CPD, 1 ICP point cloud, 2 ICP surface; z (nullable): the draws of a sampled proposal (one iteration)
int group_update(gingr_group *g, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int32_t n_iterations, const double *z) {
    if (!g || n_iterations < 0 || flavour < 0 || flavour > 2) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0] || g->xch.empty()) return group_fail(g, GINGR_ERR_STATE, "group update: no model / target set");
    const bool gather = flavour == 2 || (flavour == 1 && g->reversed);  // the tests against the template need all of it
    if (gather && !g->meshes) return group_fail(g, GINGR_ERR_STATE, "group update: no meshes set (gingr_group_set_meshes)");
    if (z && n_iterations != 1) return group_fail(g, GINGR_ERR_BAD_ARGUMENT, "group update: a sampled proposal is one iteration");
    if (g->n == 1)
        return g->run([&](int) {
            gingr_fitter *f = g->fit[0];
            if (z) {
                if (flavour == 0) return gingr_fitter_update_cpd_sample_async(f, cp, z);
                return flavour == 1 ? gingr_fitter_update_icp_sample_async(f, ip, z) : gingr_fitter_update_icp_surface_sample_async(f, ip, z);
            }
            if (flavour == 0) return gingr_fitter_update_cpd_async(f, cp, n_iterations);
            return flavour == 1 ? gingr_fitter_update_icp_async(f, ip, n_iterations) : gingr_fitter_update_icp_surface_async(f, ip, n_iterations);
        });
    const int64_t it0 = g->iteration;
    g->iteration += n_iterations;
    return g->run([&](int r) {
        gingr_fitter *f = g->fit[(size_t)r];
        int rc = fitter_set_zrand(f, z);  // (a shard that failed here still takes part in every exchange below, flagged not-ok)
        for (int32_t it = 0; it < n_iterations; ++it) {
            const int parity = (int)((it0 + it) & 1);
            if (!rc) fitter_set_partial_output(f, g->send[parity][(size_t)r]);
            TimerScope ts(g->ctx[(size_t)r], 3);
            if (gather) {  // the whole posed template for the tests against it: gather the shards' rows of the fit
                if (!rc) fitter_set_partial_fullfit(f, g->sendfit[parity][(size_t)r]);
                if (!rc) rc = fitter_run_phase(f, flavour, cp, ip, GINGR_PHASE_GATHER);
                const int xrc = exchange_fullfit(g, r, parity, rc == GINGR_OK);
                if (!rc) rc = xrc;
            }
            if (!rc && g->reversed && flavour != 0) fitter_set_partial_revsum(f, g->sendrev[parity][(size_t)r]);
            for (int ph = 0; ph < GINGR_NUM_PHASES; ++ph) {
                if (!rc) rc = fitter_run_phase(f, flavour, cp, ip, ph);
                // ICP: the correspondence phase has nothing to exchange (rows are independent) ...
                if (ph < GINGR_NUM_SEGMENTS && !(flavour != 0 && ph == 0)) {
                    const int xrc = timed_exchange_segment(g, r, ph, parity, rc == GINGR_OK);
                    if (!rc) rc = xrc;
                }
                // ... except in the reversed direction: every shard scanned its range of the target queries, the sums are totalled
                if (ph == 0 && flavour != 0 && g->reversed) {
                    const int xrc = exchange_revsum(g, r, parity, rc == GINGR_OK);
                    if (!rc) rc = xrc;
                }
            }
        }
        (void)fitter_set_zrand(f, nullptr);
        return rc;
    });
}

// posterior(state).gp.logpdf(posterior.coefficients(mesh)): phases 0 and 1 with their exchanges, Q0^T e in the tail of segment 1, the
// log-density kernel replicated on every shard (shard 0's value is returned; they are bit-identical)
int group_logpdf(gingr_group *g, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, const double *mesh_xyz_full,
                 double *logpdf) {
    if (!g || !mesh_xyz_full || !logpdf || flavour < 0 || flavour > 2) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0] || g->xch.empty()) return group_fail(g, GINGR_ERR_STATE, "group posterior_logpdf: no model / target set");
    const bool gather = flavour == 2 || (flavour == 1 && g->reversed);
    if (gather && !g->meshes) return group_fail(g, GINGR_ERR_STATE, "group posterior_logpdf: no meshes set (gingr_group_set_meshes)");
    if (g->n == 1)
        return g->run([&](int) {
            gingr_fitter *f = g->fit[0];
            if (flavour == 0) return gingr_fitter_posterior_logpdf_cpd(f, cp, mesh_xyz_full, logpdf);
            return flavour == 1 ? gingr_fitter_posterior_logpdf_icp(f, ip, mesh_xyz_full, logpdf)
                                : gingr_fitter_posterior_logpdf_icp_surface(f, ip, mesh_xyz_full, logpdf);
        });
    const int parity = (int)(g->iteration & 1);
    g->iteration += 1;
    std::vector<double> out((size_t)g->n, 0.0);
    GINGR_TRY(g->run([&](int r) {
        int rc = GINGR_OK;
        gingr_fitter *f = g->fit[(size_t)r];
        fitter_set_partial_output(f, g->send[parity][(size_t)r]);
        if (gather) {
            fitter_set_partial_fullfit(f, g->sendfit[parity][(size_t)r]);
            rc = fitter_run_phase(f, flavour, cp, ip, GINGR_PHASE_GATHER);
            const int xrc = exchange_fullfit(g, r, parity, rc == GINGR_OK);
            if (!rc) rc = xrc;
        }
        if (!rc && g->reversed && flavour != 0) fitter_set_partial_revsum(f, g->sendrev[parity][(size_t)r]);
        if (!rc) rc = fitter_run_phase(f, flavour, cp, ip, 0);
        if (flavour == 0) {
            const int xrc = exchange_segment(g, r, 0, parity, rc == GINGR_OK);
            if (!rc) rc = xrc;
        } else if (g->reversed) {
            const int xrc = exchange_revsum(g, r, parity, rc == GINGR_OK);
            if (!rc) rc = xrc;
        }
        if (!rc) rc = fitter_run_phase(f, flavour, cp, ip, 1);
        if (!rc) rc = fitter_logpdf_prepare(f, mesh_xyz_full);
        const int xrc = exchange_segment(g, r, 1, parity, rc == GINGR_OK);
        if (!rc) rc = xrc;
        if (!rc) rc = fitter_logpdf_finish(f, &out[(size_t)r]);
        return rc;
    }));
    *logpdf = out[0];
    return GINGR_OK;
}

}  // namespace

extern "C" {

int gingr_group_create(int32_t ndev, const int32_t *devices, gingr_group **out) {
    if (!out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (ndev < 1 || ndev > kMaxGroup || !devices) return GINGR_ERR_BAD_ARGUMENT;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return GINGR_ERR_NO_DEVICE;
    for (int r = 0; r < ndev; ++r)
        if (devices[r] < 0 || devices[r] >= have) return GINGR_ERR_BAD_ARGUMENT;
    gingr_group *g = new gingr_group();
    g->n = ndev;
    g->dev.assign(devices, devices + ndev);
    g->ctx.assign((size_t)ndev, nullptr);
    g->model.assign((size_t)ndev, nullptr);
    g->fit.assign((size_t)ndev, nullptr);
    g->begin.assign((size_t)ndev, 0);
    g->end.assign((size_t)ndev, 0);
    g->status.assign((size_t)ndev, GINGR_OK);
    g->bar.n = ndev;
    {
        std::vector<int> uniq(g->dev);
        std::sort(uniq.begin(), uniq.end());
        g->distinct_devices = (int)(std::unique(uniq.begin(), uniq.end()) - uniq.begin());
    }
    DeviceGuard guard;
    for (int r = 0; r < ndev; ++r) {
        const int rc = gingr_ctx_create(devices[r], &g->ctx[(size_t)r]);
        if (rc) {
            for (auto c : g->ctx)
                if (c) gingr_ctx_destroy(c);
            delete g;
            return rc;
        }
    }
    // peer mappings between distinct devices (a kernel on device a reads the send buffers that live on device b)
    for (int a = 0; a < ndev; ++a)
        for (int b = 0; b < ndev; ++b) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, devices[a], devices[b]);
            if (!can) {
                for (auto c : g->ctx) gingr_ctx_destroy(c);
                delete g;
                return GINGR_ERR_HIP;  // no peer path between two devices of the group
            }
            (void)hipSetDevice(devices[a]);
            const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                for (auto c : g->ctx) gingr_ctx_destroy(c);
                delete g;
                return GINGR_ERR_HIP;
            }
            (void)hipGetLastError();
        }
    for (int r = 0; r < ndev; ++r) g->workers.emplace_back(worker_main, g, r);
    *out = g;
    return GINGR_OK;
}

void gingr_group_destroy(gingr_group *g) {
    if (!g) return;
    (void)gingr_group_synchronize(g);
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->quit = true;
    }
    g->cv_job.notify_all();
    for (auto &t : g->workers) t.join();
    free_models(g);
    for (auto c : g->ctx)
        if (c) gingr_ctx_destroy(c);
    delete g;
}

int32_t gingr_group_size(const gingr_group *g) { return g ? g->n : 0; }
const char *gingr_group_last_error(const gingr_group *g) { return g ? g->err : "null group"; }
gingr_ctx *gingr_group_ctx(gingr_group *g, int32_t shard) { return (g && shard >= 0 && shard < g->n) ? g->ctx[(size_t)shard] : nullptr; }

int gingr_group_shard_rows(const gingr_group *g, int32_t shard, int64_t *row_begin, int64_t *row_end) {
    if (!g || shard < 0 || shard >= g->n || g->M_total <= 0) return GINGR_ERR_BAD_ARGUMENT;
    if (row_begin) *row_begin = g->begin[(size_t)shard];
    if (row_end) *row_end = g->end[(size_t)shard];
    return GINGR_OK;
}

int gingr_group_model_upload(gingr_group *g, int64_t M_total, int32_t rank, const double *ref, const double *mean,
                             const double *basis_colmajor, const double *variance) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (M_total < g->n) return group_fail(g, GINGR_ERR_BAD_ARGUMENT, "group: fewer points than shards");
    free_models(g);
    g->M_total = M_total;
    for (int r = 0; r < g->n; ++r) shard_rows(M_total, g->n, r, &g->begin[(size_t)r], &g->end[(size_t)r]);
    GINGR_TRY(g->run([&](int r) {
        return gingr_model_upload(g->ctx[(size_t)r], M_total, rank, ref, mean, basis_colmajor, variance, g->begin[(size_t)r],
                                  g->end[(size_t)r], &g->model[(size_t)r]);
    }));
    return finish_models(g);
}

int gingr_group_gpmm_build_gaussian(gingr_group *g, int64_t M_total, const double *ref, int32_t n_kernels, const double *sigmas,
                                    const double *scalings, double relative_tolerance, int32_t max_rank) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (M_total < g->n) return group_fail(g, GINGR_ERR_BAD_ARGUMENT, "group: fewer points than shards");
    free_models(g);
    g->M_total = M_total;
    for (int r = 0; r < g->n; ++r) shard_rows(M_total, g->n, r, &g->begin[(size_t)r], &g->end[(size_t)r]);
    GINGR_TRY(g->run([&](int r) {
        return gingr_gpmm_build_gaussian(g->ctx[(size_t)r], M_total, ref, n_kernels, sigmas, scalings, relative_tolerance, max_rank,
                                         g->begin[(size_t)r], g->end[(size_t)r], &g->model[(size_t)r]);
    }));
    return finish_models(g);
}

int32_t gingr_group_model_rank(const gingr_group *g) { return g ? g->rank : 0; }

int gingr_group_set_target(gingr_group *g, int64_t N, const double *target_xyz) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0]) return group_fail(g, GINGR_ERR_STATE, "group set_target: no model");
    GINGR_TRY(gingr_group_synchronize(g));
    free_exchange(g);
    GINGR_TRY(g->run([&](int r) { return gingr_fitter_set_target(g->fit[(size_t)r], N, target_xyz); }));
    g->N = N;
    g->xch.assign((size_t)g->n, nullptr);
    for (int r = 0; r < g->n; ++r) {
        void *p = nullptr;
        int64_t off[GINGR_NUM_SEGMENTS], cnt[GINGR_NUM_SEGMENTS];
        GINGR_TRY(gingr_fitter_exchange(g->fit[(size_t)r], &p, off, cnt));
        g->xch[(size_t)r] = static_cast<double *>(p);
        for (int s = 0; s < GINGR_NUM_SEGMENTS; ++s) {
            g->off[s] = off[s];
            g->cnt[s] = cnt[s];
        }
    }
    if (g->n == 1) return GINGR_OK;
    const int64_t total = round_up(g->off[GINGR_NUM_SEGMENTS - 1] + g->cnt[GINGR_NUM_SEGMENTS - 1], 32);
    DeviceGuard guard;
    // any failure below leaves NO exchange behind (free_exchange): group_update then refuses to run instead of handing null send
    // buffers / events to the kernels
    auto fail = [&](int code, const char *what) {
        free_exchange(g);
        return what ? group_fail(g, code, what) : code;
    };
    g->fine_grained = true;
    for (int p = 0; p < 2; ++p) {
        g->send[p].assign((size_t)g->n, nullptr);
        for (int s = 0; s < GINGR_NUM_SEGMENTS; ++s) g->ready[p][s].assign((size_t)g->n, nullptr);
        for (int r = 0; r < g->n; ++r) {
            if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess) return fail(GINGR_ERR_HIP, "group: hipSetDevice failed");
            void *buf = nullptr;
            bool fine = false;
            const int rc = alloc_peer_readable(g, (size_t)total * sizeof(double), &buf, &fine);
            if (rc) return fail(rc, nullptr);
            g->send[p][(size_t)r] = static_cast<double *>(buf);
            g->fine_grained = g->fine_grained && fine;
            // (hipMemset runs on the null stream, which the contexts' non-blocking streams do not order against: wait for it here)
            if (hipMemset(buf, 0, (size_t)total * sizeof(double)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
                return fail(GINGR_ERR_HIP, "group: memset failed");
            for (int s = 0; s < GINGR_NUM_SEGMENTS; ++s)
                if (hipEventCreateWithFlags(&g->ready[p][s][(size_t)r], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess)
                    return fail(GINGR_ERR_HIP, "group: hipEventCreate failed");
        }
    }
    return GINGR_OK;
}

int gingr_group_set_meshes(gingr_group *g, int64_t n_model_triangles, const int32_t *model_triangles, int64_t n_target_triangles,
                           const int32_t *target_triangles) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0] || g->xch.empty()) return group_fail(g, GINGR_ERR_STATE, "group set_meshes: no model / target set");
    GINGR_TRY(gingr_group_synchronize(g));
    // every shard's set_meshes drops its direction and frees its reversed-direction sums (fitter.hip: free_meshes): the group forgets
    // the direction with them, or the next update would hand a null total buffer to the peer-sum kernel
    g->reversed = false;
    GINGR_TRY(g->run([&](int r) {
        return gingr_fitter_set_meshes(g->fit[(size_t)r], n_model_triangles, model_triangles, n_target_triangles, target_triangles);
    }));
    if (g->n == 1 || !g->sendfit[0].empty()) {
        g->meshes = true;
        return GINGR_OK;
    }
    // the contributions to the gathered fit, peer-readable and double buffered like the other send buffers.  `meshes` is set only
    // once every buffer and event exists; a failure on the way leaves none of them behind (group_update then refuses to gather
    // instead of handing null peer pointers to the exchange kernel)
    DeviceGuard guard;
    const size_t bytes = (size_t)3 * g->M_total * sizeof(double);
    auto build = [&]() -> int {
        for (int p = 0; p < 2; ++p) {
            g->sendfit[p].assign((size_t)g->n, nullptr);
            g->readyfit[p].assign((size_t)g->n, nullptr);
            for (int r = 0; r < g->n; ++r) {
                if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess) return group_fail(g, GINGR_ERR_HIP, "group: hipSetDevice failed");
                void *buf = nullptr;
                bool fine = false;
                GINGR_TRY(alloc_peer_readable(g, bytes, &buf, &fine));
                g->sendfit[p][(size_t)r] = static_cast<double *>(buf);
                g->fine_grained = g->fine_grained && fine;
                if (hipMemset(buf, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
                    return group_fail(g, GINGR_ERR_HIP, "group: memset failed");
                if (hipEventCreateWithFlags(&g->readyfit[p][(size_t)r], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess)
                    return group_fail(g, GINGR_ERR_HIP, "group: hipEventCreate failed");
            }
        }
        return GINGR_OK;
    };
    const int rc = build();
    if (rc != GINGR_OK) {
        for (int p = 0; p < 2; ++p) {
            for (size_t r = 0; r < g->sendfit[p].size(); ++r)
                if (g->sendfit[p][r]) {
                    (void)hipSetDevice(g->dev[r]);
                    (void)hipFree(g->sendfit[p][r]);
                }
            g->sendfit[p].clear();
            for (size_t r = 0; r < g->readyfit[p].size(); ++r)
                if (g->readyfit[p][r]) (void)hipEventDestroy(g->readyfit[p][r]);
            g->readyfit[p].clear();
        }
        return rc;
    }
    g->meshes = true;
    return GINGR_OK;
}

int gingr_group_set_surface_method(gingr_group *g, int32_t method) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0]) return group_fail(g, GINGR_ERR_STATE, "group set_surface_method: no model");
    return g->run([&](int r) { return gingr_fitter_set_surface_method(g->fit[(size_t)r], method); });
}

int gingr_group_set_correspondence_direction(gingr_group *g, int32_t reversed) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0]) return group_fail(g, GINGR_ERR_STATE, "group set_correspondence_direction: no model");
    if (reversed && g->n > 1 && !g->meshes)
        return group_fail(g, GINGR_ERR_STATE, "group set_correspondence_direction: set the meshes first (the reversed direction works on the gathered template)");
    GINGR_TRY(g->run([&](int r) { return gingr_fitter_set_correspondence_direction(g->fit[(size_t)r], reversed); }));
    if (reversed && g->n > 1 && g->sendrev[0].empty()) {
        // the shards' contributions to the per-template-vertex sums, peer-readable and double buffered like the other send buffers;
        // `reversed` is set only once all of them exist
        DeviceGuard guard;
        const size_t bytes = (size_t)4 * g->M_total * sizeof(double);
        auto build = [&]() -> int {
            for (int p = 0; p < 2; ++p) {
                g->sendrev[p].assign((size_t)g->n, nullptr);
                g->readyrev[p].assign((size_t)g->n, nullptr);
                for (int r = 0; r < g->n; ++r) {
                    if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess) return group_fail(g, GINGR_ERR_HIP, "group: hipSetDevice failed");
                    void *buf = nullptr;
                    bool fine = false;
                    GINGR_TRY(alloc_peer_readable(g, bytes, &buf, &fine));
                    g->sendrev[p][(size_t)r] = static_cast<double *>(buf);
                    g->fine_grained = g->fine_grained && fine;
                    if (hipMemset(buf, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
                        return group_fail(g, GINGR_ERR_HIP, "group: memset failed");
                    if (hipEventCreateWithFlags(&g->readyrev[p][(size_t)r], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess)
                        return group_fail(g, GINGR_ERR_HIP, "group: hipEventCreate failed");
                }
            }
            return GINGR_OK;
        };
        const int rc = build();
        if (rc != GINGR_OK) {
            for (int p = 0; p < 2; ++p) {
                for (size_t r = 0; r < g->sendrev[p].size(); ++r)
                    if (g->sendrev[p][r]) {
                        (void)hipSetDevice(g->dev[r]);
                        (void)hipFree(g->sendrev[p][r]);
                    }
                g->sendrev[p].clear();
                for (size_t r = 0; r < g->readyrev[p].size(); ++r)
                    if (g->readyrev[p][r]) (void)hipEventDestroy(g->readyrev[p][r]);
                g->readyrev[p].clear();
            }
            return rc;
        }
    }
    g->reversed = reversed != 0;
    return GINGR_OK;
}

int gingr_group_update_async(gingr_group *g, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int32_t n_iterations,
                             const double *z) {
    return group_update(g, flavour, cp, ip, n_iterations, z);
}

int gingr_group_posterior_logpdf(gingr_group *g, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                 const double *mesh_xyz_full, double *logpdf) {
    return group_logpdf(g, flavour, cp, ip, mesh_xyz_full, logpdf);
}

int gingr_group_exchange_info(const gingr_group *g, int32_t *distinct_devices, int32_t *fine_grained) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (distinct_devices) *distinct_devices = g->distinct_devices;
    if (fine_grained) *fine_grained = (g->n > 1 && !g->send[0].empty() && g->fine_grained) ? 1 : 0;
    return GINGR_OK;
}

int gingr_group_set_landmarks(gingr_group *g, int32_t n_lm, const int32_t *lm_pid, const double *lm_xyz, const double *lm_cov) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0]) return group_fail(g, GINGR_ERR_STATE, "group set_landmarks: no model");
    return g->run([&](int r) { return gingr_fitter_set_landmarks(g->fit[(size_t)r], n_lm, lm_pid, lm_xyz, lm_cov); });
}

int gingr_group_set_options(gingr_group *g, int32_t global_transform, double step_length) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0]) return group_fail(g, GINGR_ERR_STATE, "group set_options: no model");
    for (int r = 0; r < g->n; ++r) GINGR_TRY(gingr_fitter_set_options(g->fit[(size_t)r], global_transform, step_length));
    return GINGR_OK;
}

int gingr_group_set_state(gingr_group *g, const double *alpha, const gingr_state_scalars *s) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0]) return group_fail(g, GINGR_ERR_STATE, "group set_state: no model");
    return g->run([&](int r) { return gingr_fitter_set_state(g->fit[(size_t)r], alpha, s); });
}

int gingr_group_get_state(gingr_group *g, double *alpha, gingr_state_scalars *s, double *fit_xyz) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    if (g->fit.empty() || !g->fit[0]) return group_fail(g, GINGR_ERR_STATE, "group get_state: no model");
    return g->run([&](int r) {
        return gingr_fitter_get_state(g->fit[(size_t)r], r == 0 ? alpha : nullptr, r == 0 ? s : nullptr,
                                      fit_xyz ? fit_xyz + 3 * g->begin[(size_t)r] : nullptr);
    });
}

int gingr_group_update_cpd_async(gingr_group *g, const gingr_cpd_params *p, int32_t n_iterations) {
    if (!g || !p) return GINGR_ERR_BAD_ARGUMENT;
    return group_update(g, 0, p, nullptr, n_iterations, nullptr);
}

int gingr_group_update_icp_async(gingr_group *g, const gingr_icp_params *p, int32_t n_iterations) {
    if (!g || !p) return GINGR_ERR_BAD_ARGUMENT;
    return group_update(g, 1, nullptr, p, n_iterations, nullptr);
}

int gingr_group_synchronize(gingr_group *g) {
    if (!g) return GINGR_ERR_BAD_ARGUMENT;
    for (int r = 0; r < g->n; ++r) GINGR_TRY(gingr_ctx_synchronize(g->ctx[(size_t)r]));
    return GINGR_OK;
}

}  // extern "C"

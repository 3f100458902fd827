// Native RCCL exchange of the row-sharded update (one process per GPU): the library itself enqueues ncclAllReduce on the context's
// stream between the phases of an iteration -- "RCCL all-reduce over xGMI for CPD column sums" (BASELINE.json north_star) with no
// callback into the host, no stream hop and no Python between the kernels.  librccl is bound at run time (dlopen): a single-GPU
// host never needs it, and a host that already carries one (torch ships its own librccl.so.1) shares that copy.
// Reference: none -- the reference is single-process / single-device; the protocol is the one of
// gingr_fitter_update_*_sharded_async (include/gingr_hip.h), the collective is RCCL's.
#include "gp.h"

#include <dlfcn.h>

#include <mutex>

namespace {

// the entry points used, with RCCL's C signatures (rccl.h): ncclResult_t is an int enum (0 = success), ncclComm_t a pointer,
// ncclUniqueId 128 opaque bytes passed BY VALUE, ncclFloat64 = 8, ncclSum = 0
struct UniqueId {
    char internal[GINGR_RCCL_UNIQUE_ID_BYTES];
};
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;  // (optional: the gather falls back to AllReduce)
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    std::string path;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl(gingr_ctx *ctx, const char *path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) {
        // one librccl per process: an explicit request for ANOTHER file is refused, not silently ignored
        if (path && *path && g_rccl.path != path)
            return gingr_set_error(ctx, GINGR_ERR_STATE, "rccl: %s is already bound in this process, cannot bind %s", g_rccl.path.c_str(), path);
        return GINGR_OK;
    }
    void *h = nullptr;
    if (path && *path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) return gingr_set_error(ctx, GINGR_ERR_STATE, "rccl: cannot load %s: %s", path, dlerror());
        g_rccl.path = path;
    } else {
        // the copy the process already carries (matched by SONAME), else the loader's search path
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        g_rccl.path = h ? "librccl.so.1 (already loaded in this process)" : "librccl.so.1";
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return gingr_set_error(ctx, GINGR_ERR_STATE, "rccl: librccl.so.1 not found (%s); pass its path to gingr_rccl_load", dlerror());
    }
    Rccl r;
    r.handle = h;
    r.path = g_rccl.path;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
        dlclose(h);
        return gingr_set_error(ctx, GINGR_ERR_STATE, "rccl: %s lacks an entry point (ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce)",
                               r.path.c_str());
    }
    g_rccl = r;
    return GINGR_OK;
}

int rccl_fail(gingr_ctx *ctx, const char *what, int rc) {
    return gingr_set_error(ctx, GINGR_ERR_HIP, "rccl: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
}

// the gingr_allreduce_fn of the native path: user = the context
int native_allreduce(void *user, int32_t, void *device_ptr, int64_t count) {
    gingr_ctx *ctx = static_cast<gingr_ctx *>(user);
    // (exchange_stream: the first half of a split column-sum exchange runs on the context's second stream, fitter.hip)
    const int rc = g_rccl.AllReduce(device_ptr, device_ptr, (size_t)count, /* ncclFloat64 */ 8, /* ncclSum */ 0, ctx->rccl_comm,
                                    ctx->exchange_stream ? ctx->exchange_stream : ctx->stream);
    if (rc != 0) {
        (void)rccl_fail(ctx, "ncclAllReduce", rc);
        return 1;
    }
    return 0;
}

// the fitter_gather_fn of the native path: the shard's rows into its slot of the staging buffer, ncclAllGather in place (every rank
// sends the same count: slots are padded to ceil(M_total / world) rows), one kernel spreads the slots over the planes of the full
// fit.  Half the wire bytes of the zero-padded all-reduce and no reduction (round 4 spelled the gather as a sum).
int native_gather(void *user, gingr_fitter *f) {
    gingr_ctx *ctx = static_cast<gingr_ctx *>(user);
    if (!g_rccl.AllGather) return 1;  // (the same library on every rank of a node: a uniform answer)
    // Gather or zero-padded all-reduce is a choice between two DIFFERENT collectives on one communicator: it has to come out the same
    // on every rank.  Whether a rank can gather depends on its own rows only (the balanced partition), so the ranks agree once per
    // (meshes, world) with a MIN all-reduce of a flag; a rank that cannot stage after the agreement is an error, never a silent
    // switch to the other collective (the peers would wait in ncclAllGather for ever).
    int agreed = fitter_gather_agreed(f, ctx->rccl_world);
    if (agreed < 0) {
        double flag = fitter_gather_possible(f, ctx->rccl_world, ctx->rccl_rank) ? 1.0 : 0.0, *dflag = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&dflag), sizeof(double)) != hipSuccess ||
            hipMemcpyAsync(dflag, &flag, sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            (void)hipFree(dflag);
            (void)gingr_set_error(ctx, GINGR_ERR_HIP, "rccl gather: no device word for the agreement");
            return -1;
        }
        const int rc = g_rccl.AllReduce(dflag, dflag, 1, /* ncclFloat64 */ 8, /* ncclMin */ 3, ctx->rccl_comm, ctx->stream);
        const bool copied = rc == 0 && hipMemcpyAsync(&flag, dflag, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                            hipStreamSynchronize(ctx->stream) == hipSuccess;
        (void)hipFree(dflag);
        if (rc != 0) {
            (void)rccl_fail(ctx, "ncclAllReduce (gather agreement)", rc);
            return -1;
        }
        if (!copied) {
            (void)gingr_set_error(ctx, GINGR_ERR_HIP, "rccl gather: reading the agreement back failed");
            return -1;
        }
        agreed = flag > 0.5 ? 1 : 0;
        fitter_set_gather_agreed(f, ctx->rccl_world, agreed);
    }
    if (!agreed) return 1;  // every rank falls back to the zero-padded all-reduce
    void *send = nullptr, *recv = nullptr;
    int64_t count = 0;
    if (gingr_fitter_gather_stage(f, ctx->rccl_world, ctx->rccl_rank, &send, &recv, &count) != GINGR_OK) return -1;  // (the error text is set)
    const int rc = g_rccl.AllGather(send, recv, (size_t)count, /* ncclFloat64 */ 8, ctx->rccl_comm, ctx->stream);
    if (rc != 0) {
        (void)rccl_fail(ctx, "ncclAllGather", rc);
        return -1;
    }
    return gingr_fitter_gather_finish(f, ctx->rccl_world) == GINGR_OK ? 0 : -1;
}

}  // namespace

void gingr_ctx_rccl_release(gingr_ctx *ctx) {  // gingr_ctx_destroy
    if (ctx->rccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
    ctx->rccl_world = 0;
}

extern "C" {

int gingr_rccl_load(gingr_ctx *ctx, const char *librccl_path) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    return load_rccl(ctx, librccl_path);
}

int gingr_rccl_unique_id(gingr_ctx *ctx, void *id_bytes) {
    if (!ctx || !id_bytes) return GINGR_ERR_BAD_ARGUMENT;
    GINGR_TRY(load_rccl(ctx, nullptr));
    UniqueId id;
    memset(&id, 0, sizeof(id));
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return rccl_fail(ctx, "ncclGetUniqueId", rc);
    memcpy(id_bytes, id.internal, sizeof(id.internal));
    return GINGR_OK;
}

int gingr_ctx_rccl_init(gingr_ctx *ctx, const void *id_bytes, int32_t world, int32_t rank) {
    if (!ctx || !id_bytes || world < 1 || rank < 0 || rank >= world)
        return ctx ? gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "rccl_init: need 0 <= rank < world") : GINGR_ERR_BAD_ARGUMENT;
    if (ctx->rccl_comm) return gingr_set_error(ctx, GINGR_ERR_STATE, "rccl_init: this context already has a communicator");
    GINGR_TRY(load_rccl(ctx, nullptr));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    UniqueId id;
    memcpy(id.internal, id_bytes, sizeof(id.internal));
    void *comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, (int)world, id, (int)rank);
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    ctx->rccl_comm = comm;
    ctx->rccl_world = world;
    ctx->rccl_rank = rank;
    return GINGR_OK;
}

int gingr_ctx_rccl_info(gingr_ctx *ctx, int32_t *world, int32_t *rank, int32_t *version, char *library_path, int32_t path_capacity) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if (world) *world = ctx->rccl_comm ? ctx->rccl_world : 0;
    if (rank) *rank = ctx->rccl_comm ? ctx->rccl_rank : -1;
    if (version) {
        int v = 0;
        if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
        *version = v;
    }
    if (library_path && path_capacity > 0) snprintf(library_path, (size_t)path_capacity, "%s", g_rccl.path.c_str());
    return GINGR_OK;
}

int gingr_ctx_rccl_allreduce_async(gingr_ctx *ctx, void *device_ptr, int64_t count) {
    if (!ctx || !device_ptr || count < 0) return GINGR_ERR_BAD_ARGUMENT;
    if (!ctx->rccl_comm) return gingr_set_error(ctx, GINGR_ERR_STATE, "rccl_allreduce: no communicator (gingr_ctx_rccl_init)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return native_allreduce(ctx, 0, device_ptr, count) == 0 ? GINGR_OK : GINGR_ERR_HIP;
}

int gingr_fitter_update_cpd_rccl_async(gingr_fitter *f, const gingr_cpd_params *p, int32_t n_iterations) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = fitter_ctx(f);
    if (!ctx->rccl_comm) return gingr_set_error(ctx, GINGR_ERR_STATE, "update_cpd_rccl: no communicator (gingr_ctx_rccl_init)");
    return fitter_sharded_update(f, 0, p, nullptr, n_iterations, nullptr, native_allreduce, ctx, nullptr, ctx->split_exchange != 0);
}

int gingr_fitter_update_icp_rccl_async(gingr_fitter *f, const gingr_icp_params *p, int32_t n_iterations) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = fitter_ctx(f);
    if (!ctx->rccl_comm) return gingr_set_error(ctx, GINGR_ERR_STATE, "update_icp_rccl: no communicator (gingr_ctx_rccl_init)");
    return gingr_fitter_update_icp_sharded_async(f, p, n_iterations, native_allreduce, ctx);
}

int gingr_fitter_update_rccl_async(gingr_fitter *f, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int32_t n_iterations,
                                   const double *z) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = fitter_ctx(f);
    if (!ctx->rccl_comm) return gingr_set_error(ctx, GINGR_ERR_STATE, "update_rccl: no communicator (gingr_ctx_rccl_init)");
    return fitter_sharded_update(f, flavour, cp, ip, n_iterations, z, native_allreduce, ctx, native_gather, ctx->split_exchange != 0);
}

int gingr_fitter_posterior_logpdf_rccl(gingr_fitter *f, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                       const double *mesh_xyz_full, double *logpdf) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = fitter_ctx(f);
    if (!ctx->rccl_comm) return gingr_set_error(ctx, GINGR_ERR_STATE, "posterior_logpdf_rccl: no communicator (gingr_ctx_rccl_init)");
    return fitter_sharded_logpdf(f, flavour, cp, ip, mesh_xyz_full, native_allreduce, ctx, logpdf, native_gather);
}

}  // extern "C"

/* One RANK of a row-sharded CPD registration driven from plain C through the library's own RCCL exchange (no Python, no torch in
 * the process): what a JVM host does per GPU -- rank 0 creates the ncclUniqueId and hands its 128 bytes to the others through a
 * FILE, every rank builds its communicator on its own device (gingr_ctx_rccl_init), all-reduces the one-off basis moments and runs
 * the fused sharded update (gingr_fitter_update_cpd_rccl_async: ncclAllReduce enqueued by the library between the phases).
 *
 *   cabi_rccl_rank <rank> <world> <device> <idfile> <input.bin> <output.txt>
 *
 * input.bin: the header and arrays of tests/c/cabi_fitter_driver.c (triangles ignored).  output.txt: "alpha ...", "scalars ...",
 * "probe ..." (%.17g).  Every rank must end in the same replicated state, equal to the single-shard run.
 * Runs wherever `world` GPUs are visible (tests/test_gpu_multi_device.py skips it on a one-GPU box; with world = 1 it is the
 * one-rank communicator and runs anywhere). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "gingr_hip.h"

#define CHECK(call)                                                                                              \
    do {                                                                                                         \
        int rc__ = (call);                                                                                       \
        if (rc__ != GINGR_OK) {                                                                                  \
            fprintf(stderr, "rank %d %s:%d %s -> %d (%s)\n", rank, __FILE__, __LINE__, #call, rc__, ctx ? gingr_last_error(ctx) : ""); \
            return 10;                                                                                           \
        }                                                                                                        \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 7) return 2;
    const int rank = atoi(argv[1]), world = atoi(argv[2]), device = atoi(argv[3]);
    const char *idfile = argv[4];
    FILE *in = fopen(argv[5], "rb");
    if (!in) return 2;
    int64_t hdr[6];
    if (fread(hdr, sizeof(int64_t), 6, in) != 6) return 2;
    const int64_t M = hdr[0], N = hdr[1], r = hdr[2];
    const int32_t n_iter = (int32_t)hdr[5];
    double *ref = malloc(sizeof(double) * 3 * M), *mean = malloc(sizeof(double) * 3 * M), *basis = malloc(sizeof(double) * 3 * M * r);
    double *var = malloc(sizeof(double) * r), *target = malloc(sizeof(double) * 3 * N), s2w[2];
    if (!ref || !mean || !basis || !var || !target) return 9;
    if (fread(ref, 8, 3 * M, in) != (size_t)(3 * M) || fread(mean, 8, 3 * M, in) != (size_t)(3 * M) ||
        fread(basis, 8, 3 * M * r, in) != (size_t)(3 * M * r) || fread(var, 8, r, in) != (size_t)r ||
        fread(target, 8, 3 * N, in) != (size_t)(3 * N) || fread(s2w, 8, 2, in) != 2)
        return 2;
    fclose(in);

    gingr_ctx *ctx = NULL;
    CHECK(gingr_ctx_create(device, &ctx));
    /* the id: rank 0 writes <idfile>.tmp and renames it (readers never see half a file) */
    unsigned char id[GINGR_RCCL_UNIQUE_ID_BYTES];
    if (rank == 0) {
        CHECK(gingr_rccl_unique_id(ctx, id));
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 3;
        fclose(f);
        if (rename(tmp, idfile) != 0) return 3;
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 6000 && !(f = fopen(idfile, "rb")); ++tries) usleep(10000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 3;
        fclose(f);
    }
    CHECK(gingr_ctx_rccl_init(ctx, id, world, rank));
    int32_t w_info = 0, r_info = -1, version = 0;
    char lib[512];
    CHECK(gingr_ctx_rccl_info(ctx, &w_info, &r_info, &version, lib, (int32_t)sizeof lib));
    if (w_info != world || r_info != rank) return 4;

    /* contiguous balanced rows, like gingr_amd.sharded.shard_rows */
    const int64_t base = M / world, extra = M % world;
    const int64_t b = rank * base + (rank < extra ? rank : extra), e = b + base + (rank < extra ? 1 : 0);
    gingr_model *model = NULL;
    gingr_fitter *f = NULL;
    CHECK(gingr_model_upload(ctx, M, (int32_t)r, ref, mean, basis, var, b, e, &model));
    void *gptr = NULL;
    int64_t gcount = 0;
    CHECK(gingr_model_gram_exchange(model, &gptr, &gcount));
    CHECK(gingr_ctx_rccl_allreduce_async(ctx, gptr, gcount));   /* one-off: the basis moments of all shards */
    CHECK(gingr_ctx_synchronize(ctx));
    CHECK(gingr_model_finalize(ctx, model));
    CHECK(gingr_fitter_create(ctx, model, &f));
    CHECK(gingr_fitter_set_target(f, N, target));
    CHECK(gingr_fitter_set_options(f, GINGR_RIGID_TRANSFORMS, 1.0));
    double *alpha = calloc((size_t)r, sizeof(double)), *fit = malloc(sizeof(double) * 3 * (e - b));
    gingr_state_scalars s;
    memset(&s, 0, sizeof s);
    s.scale = 1.0;
    s.sigma2 = s2w[0];
    CHECK(gingr_fitter_set_state(f, alpha, &s));
    const gingr_cpd_params cp = {s2w[1], 1.0};
    CHECK(gingr_fitter_update_cpd_rccl_async(f, &cp, n_iter));
    CHECK(gingr_ctx_synchronize(ctx));
    CHECK(gingr_fitter_get_state(f, alpha, &s, fit));

    FILE *out = fopen(argv[6], "w");
    if (!out) return 2;
    fprintf(out, "rccl %d %d %d %s\n", w_info, r_info, version, lib);
    fprintf(out, "rows %lld %lld\n", (long long)b, (long long)e);
    fprintf(out, "alpha");
    for (int64_t i = 0; i < r; ++i) fprintf(out, " %.17g", alpha[i]);
    fprintf(out, "\nscalars %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %d\n", s.euler[0], s.euler[1], s.euler[2], s.translation[0],
            s.translation[1], s.translation[2], s.scale, s.sigma2, (int)s.iteration, (int)s.status);
    fprintf(out, "fit");
    for (int64_t i = 0; i < 3 * (e - b); ++i) fprintf(out, " %.17g", fit[i]);
    fprintf(out, "\n");
    fclose(out);
    gingr_fitter_destroy(f);
    gingr_model_destroy(model);
    gingr_ctx_destroy(ctx);
    return 0;
}

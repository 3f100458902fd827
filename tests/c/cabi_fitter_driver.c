/* Plain-C consumer of the FUSED path of the boundary (no Python, no torch in the process): what the JNI shim of jvm/ does, in C.
 *
 *   cabi_fitter_driver <input.bin> <output.txt>
 *
 * input.bin  (little endian): int64 M, N, r, T_model, T_target, n_iter; then float64 ref[3M], mean[3M], basis[3M*r] (column
 *            major), variance[r], target[3N], sigma2, w; then int32 model_triangles[3*T_model], target_triangles[3*T_target].
 * output.txt: one "name v0 v1 ..." line per result (%.17g); compared with the oracle by tests/test_gpu_cabi_from_c.py.
 *
 * Sections (every group of entry points of include/gingr_hip.h is called at least once):
 *   A  model upload, fitter, n_iter fused CPD updates, state + statistics + retry counter + timing hooks
 *   B  the same iterations on TWO row shards with the 3-phase / 2-segment protocol, the exchange segments summed ON THE HOST
 *      in C (hipMemcpy is the only HIP call this program makes itself)
 *   C  the same iterations through the in-library device group (gingr_group_*), two logical shards
 *   D  stateless model operators (instance, coefficients, posterior mean), GPMM built on the device + download, point-set extrema,
 *      Gaussian kernel block, initial sigma2
 *   E  ICP: point cloud (fused + indices), surface flavour (+ correspondences, distance statistics), probabilistic proposal and
 *      log transition density
 *   F  classic CPD handle (rigid), mesh distance statistics
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "gingr_hip.h"

static FILE *out;

#define CHECK(call)                                                                            \
    do {                                                                                       \
        int rc__ = (call);                                                                     \
        if (rc__ != GINGR_OK) {                                                                \
            fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc__, ctx ? gingr_last_error(ctx) : ""); \
            return 10;                                                                         \
        }                                                                                      \
    } while (0)

static void put(const char *name, const double *v, long n) {
    fprintf(out, "%s", name);
    for (long i = 0; i < n; ++i) fprintf(out, " %.17g", v[i]);
    fprintf(out, "\n");
}

static void put_state(const char *tag, const double *alpha, long r, const gingr_state_scalars *s, const double *fit, long M) {
    char name[64];
    snprintf(name, sizeof name, "%s_alpha", tag);
    put(name, alpha, r);
    double sc[13] = {s->euler[0], s->euler[1], s->euler[2], s->center[0], s->center[1], s->center[2], s->translation[0],
                     s->translation[1], s->translation[2], s->scale, s->sigma2, (double)s->iteration, (double)s->status};
    snprintf(name, sizeof name, "%s_scalars", tag);
    put(name, sc, 13);
    snprintf(name, sizeof name, "%s_fit", tag);
    put(name, fit, 3 * M);
}

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "out of memory\n");
        exit(9);
    }
    return p;
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *in = fopen(argv[1], "rb");
    out = fopen(argv[2], "w");
    if (!in || !out) return 2;
    int64_t hdr[6];
    if (fread(hdr, sizeof(int64_t), 6, in) != 6) return 2;
    const int64_t M = hdr[0], N = hdr[1], r = hdr[2], Tm = hdr[3], Tt = hdr[4];
    const int32_t n_iter = (int32_t)hdr[5];
    double *ref = xmalloc(sizeof(double) * 3 * M), *mean = xmalloc(sizeof(double) * 3 * M);
    double *basis = xmalloc(sizeof(double) * 3 * M * r), *var = xmalloc(sizeof(double) * r), *target = xmalloc(sizeof(double) * 3 * N);
    double s2w[2];
    int32_t *mtri = xmalloc(sizeof(int32_t) * 3 * Tm), *ttri = xmalloc(sizeof(int32_t) * 3 * Tt);
    if (fread(ref, 8, 3 * M, in) != (size_t)(3 * M) || fread(mean, 8, 3 * M, in) != (size_t)(3 * M) ||
        fread(basis, 8, 3 * M * r, in) != (size_t)(3 * M * r) || fread(var, 8, r, in) != (size_t)r ||
        fread(target, 8, 3 * N, in) != (size_t)(3 * N) || fread(s2w, 8, 2, in) != 2 ||
        fread(mtri, 4, 3 * Tm, in) != (size_t)(3 * Tm) || fread(ttri, 4, 3 * Tt, in) != (size_t)(3 * Tt))
        return 2;
    fclose(in);
    const double sigma2 = s2w[0], w = s2w[1];

    gingr_ctx *ctx = NULL;
    CHECK(gingr_ctx_create(0, &ctx));
    fprintf(out, "build %d devices; %s\n", gingr_device_count(), gingr_build_info());

    double *alpha = xmalloc(sizeof(double) * r), *fit = xmalloc(sizeof(double) * 3 * M), *zero = calloc((size_t)r, sizeof(double));
    gingr_state_scalars s0, s;
    memset(&s0, 0, sizeof s0);
    s0.scale = 1.0;
    s0.sigma2 = sigma2;
    const gingr_cpd_params cp = {w, 1.0};

    /* ---------------------------------------------------------------- A: fused single-shard updates */
    gingr_model *model = NULL;
    gingr_fitter *f = NULL;
    CHECK(gingr_model_upload(ctx, M, (int32_t)r, ref, mean, basis, var, 0, M, &model));
    if (gingr_model_num_points(model) != M || gingr_model_rank(model) != r) return 11;
    CHECK(gingr_fitter_create(ctx, model, &f));
    CHECK(gingr_fitter_set_target(f, N, target));
    CHECK(gingr_fitter_set_landmarks(f, 0, NULL, NULL, NULL));
    CHECK(gingr_fitter_set_options(f, GINGR_RIGID_TRANSFORMS, 1.0));
    CHECK(gingr_fitter_set_state(f, zero, &s0));
    CHECK(gingr_ctx_timing_enable(ctx, 1));
    CHECK(gingr_ctx_timing_reset(ctx));
    CHECK(gingr_fitter_update_cpd_async(f, &cp, n_iter));
    CHECK(gingr_ctx_synchronize(ctx));
    CHECK(gingr_fitter_get_state(f, alpha, &s, fit));
    put_state("A", alpha, r, &s, fit, M);
    {
        double *P1 = xmalloc(8 * M), *PX = xmalloc(8 * 3 * M), *den = xmalloc(8 * N), sc6[6];
        CHECK(gingr_fitter_get_cpd_stats(f, P1, PX, den, sc6));
        put("A_P1", P1, M);
        put("A_scalars6", sc6, 6);
        free(P1); free(PX); free(den);
        double ms = 0;
        int64_t launches = 0;
        CHECK(gingr_ctx_timing_read(ctx, 3, &ms, &launches));
        double t[2] = {ms, (double)launches};
        put("A_timing_update", t, 2);
        CHECK(gingr_ctx_timing_enable(ctx, 0));
        int32_t retry = -1;
        CHECK(gingr_fitter_retry_counter(f, -1, &retry));
        double rr = retry;
        put("A_retry", &rr, 1);
        if (gingr_ctx_get_stream(ctx) == NULL) return 12;
        CHECK(gingr_ctx_set_stream(ctx, NULL));
    }

    /* ---------------------------------------------------------------- B: two row shards, exchange summed on the host */
    {
        const int64_t h = M / 2 + 1;
        const int64_t b[2] = {0, h}, e[2] = {h, M};
        gingr_model *ms[2] = {NULL, NULL};
        gingr_fitter *fs[2] = {NULL, NULL};
        for (int k = 0; k < 2; ++k) CHECK(gingr_model_upload(ctx, M, (int32_t)r, ref, mean, basis, var, b[k], e[k], &ms[k]));
        {   /* one-off moments: sum over the shards, then finalize */
            void *p[2];
            int64_t cnt[2];
            for (int k = 0; k < 2; ++k) CHECK(gingr_model_gram_exchange(ms[k], &p[k], &cnt[k]));
            double *a = xmalloc(8 * cnt[0]), *c = xmalloc(8 * cnt[0]);
            if (hipMemcpy(a, p[0], 8 * cnt[0], hipMemcpyDeviceToHost) != hipSuccess) return 13;
            if (hipMemcpy(c, p[1], 8 * cnt[0], hipMemcpyDeviceToHost) != hipSuccess) return 13;
            for (int64_t i = 0; i < cnt[0]; ++i) a[i] += c[i];
            for (int k = 0; k < 2; ++k)
                if (hipMemcpy(p[k], a, 8 * cnt[0], hipMemcpyHostToDevice) != hipSuccess) return 13;
            free(a); free(c);
            for (int k = 0; k < 2; ++k) CHECK(gingr_model_finalize(ctx, ms[k]));
        }
        void *xp[2];
        int64_t off[GINGR_NUM_SEGMENTS], cnt[GINGR_NUM_SEGMENTS];
        for (int k = 0; k < 2; ++k) {
            CHECK(gingr_fitter_create(ctx, ms[k], &fs[k]));
            CHECK(gingr_fitter_set_target(fs[k], N, target));
            CHECK(gingr_fitter_set_options(fs[k], GINGR_RIGID_TRANSFORMS, 1.0));
            CHECK(gingr_fitter_set_state(fs[k], zero, &s0));
            CHECK(gingr_fitter_exchange(fs[k], &xp[k], off, cnt));
        }
        int64_t big = cnt[0] > cnt[1] ? cnt[0] : cnt[1];
        double *h0 = xmalloc(8 * big), *h1 = xmalloc(8 * big);
        for (int it = 0; it < n_iter; ++it)
            for (int ph = 0; ph < GINGR_NUM_PHASES; ++ph) {
                for (int k = 0; k < 2; ++k) CHECK(gingr_fitter_cpd_phase_async(fs[k], &cp, ph));
                if (ph < GINGR_NUM_SEGMENTS) {
                    CHECK(gingr_ctx_synchronize(ctx));
                    double *d0 = (double *)xp[0] + off[ph], *d1 = (double *)xp[1] + off[ph];
                    if (hipMemcpy(h0, d0, 8 * cnt[ph], hipMemcpyDeviceToHost) != hipSuccess) return 13;
                    if (hipMemcpy(h1, d1, 8 * cnt[ph], hipMemcpyDeviceToHost) != hipSuccess) return 13;
                    for (int64_t i = 0; i < cnt[ph]; ++i) h0[i] += h1[i];
                    if (hipMemcpy(d0, h0, 8 * cnt[ph], hipMemcpyHostToDevice) != hipSuccess) return 13;
                    if (hipMemcpy(d1, h0, 8 * cnt[ph], hipMemcpyHostToDevice) != hipSuccess) return 13;
                }
            }
        for (int k = 0; k < 2; ++k) CHECK(gingr_fitter_get_state(fs[k], alpha, &s, fit + 3 * b[k]));
        put_state("B", alpha, r, &s, fit, M);
        free(h0); free(h1);
        for (int k = 0; k < 2; ++k) {
            gingr_fitter_destroy(fs[k]);
            gingr_model_destroy(ms[k]);
        }
    }

    /* ---------------------------------------------------------------- C: in-library device group, two logical shards */
    {
        const int32_t devs[2] = {0, gingr_device_count() > 1 ? 1 : 0};
        gingr_group *g = NULL;
        CHECK(gingr_group_create(2, devs, &g));
        if (gingr_group_size(g) != 2 || gingr_group_ctx(g, 0) == NULL) return 14;
#define GCHECK(call)                                                                            \
    do {                                                                                        \
        int rc__ = (call);                                                                      \
        if (rc__ != GINGR_OK) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc__, gingr_group_last_error(g)); \
            return 15;                                                                          \
        }                                                                                       \
    } while (0)
        GCHECK(gingr_group_model_upload(g, M, (int32_t)r, ref, mean, basis, var));
        int64_t rb = -1, re = -1;
        GCHECK(gingr_group_shard_rows(g, 1, &rb, &re));
        if (re != M || gingr_group_model_rank(g) != r) return 14;
        GCHECK(gingr_group_set_target(g, N, target));
        {   /* how the group exchanges: logical shards on one device may live in plain device memory, two devices must not */
            int32_t nd = -1, fg = -1;
            GCHECK(gingr_group_exchange_info(g, &nd, &fg));
            if (nd != (devs[0] == devs[1] ? 1 : 2) || (nd > 1 && fg != 1)) return 14;
            const double info[2] = {(double)nd, (double)fg};
            put("C_exchange_info", info, 2);
        }
        GCHECK(gingr_group_set_landmarks(g, 0, NULL, NULL, NULL));
        GCHECK(gingr_group_set_options(g, GINGR_RIGID_TRANSFORMS, 1.0));
        GCHECK(gingr_group_set_state(g, zero, &s0));
        GCHECK(gingr_group_update_cpd_async(g, &cp, n_iter));
        GCHECK(gingr_group_synchronize(g));
        GCHECK(gingr_group_get_state(g, alpha, &s, fit));
        put_state("C", alpha, r, &s, fit, M);
        /* ICP through the group: one iteration from the CPD result */
        const gingr_icp_params ipg = {s.sigma2, 1.0, 10};
        GCHECK(gingr_group_update_icp_async(g, &ipg, 1));
        GCHECK(gingr_group_get_state(g, alpha, &s, fit));
        put_state("C_icp", alpha, r, &s, fit, M);
        gingr_group_destroy(g);
    }

    /* ---------------------------------------------------------------- D: stateless operators */
    {
        double euler[3] = {0.05, -0.02, 0.03}, center[3] = {1, 2, 3}, tr[3] = {0.5, -0.25, 0.75};
        for (int64_t k = 0; k < r; ++k) alpha[k] = 0.1 * (double)((k % 5) - 2);
        double *inst = xmalloc(8 * 3 * M), *coef = xmalloc(8 * r), *wgt = xmalloc(8 * M), *pm = xmalloc(8 * 3 * M);
        CHECK(gingr_model_instance(ctx, model, alpha, euler, center, tr, 1.0, inst));
        put("D_instance", inst, 3 * M);
        CHECK(gingr_model_coefficients(ctx, model, euler, center, tr, inst, coef));
        put("D_coefficients", coef, r);
        for (int64_t i = 0; i < M; ++i) wgt[i] = (i % 3 == 0) ? 0.0 : 2.0;
        CHECK(gingr_model_posterior_mean(ctx, model, euler, center, tr, inst, wgt, 0, NULL, NULL, NULL, pm, coef));
        put("D_posterior_coeffs", coef, r);
        double s2i = 0;
        CHECK(gingr_cpd_initial_sigma2(ctx, M, ref, N, target, &s2i));
        put("D_initial_sigma2", &s2i, 1);
        double *gb = xmalloc(8 * 4 * 5);
        CHECK(gingr_gauss_block(ctx, 4, ref, 5, target, 30.0, 2.0, gb));
        put("D_gauss_block", gb, 20);
        double ext[2];
        CHECK(gingr_pointset_distance_extrema(ctx, ref, M, &ext[0], &ext[1]));
        put("D_extrema", ext, 2);
        gingr_model *built = NULL;
        const double sg[1] = {60.0}, scl[1] = {30.0};
        CHECK(gingr_gpmm_build_gaussian(ctx, M, ref, 1, sg, scl, 0.0, 12, 0, M, &built));
        double *bvar = xmalloc(8 * 12);
        if (gingr_model_rank(built) != 12) return 16;
        CHECK(gingr_model_download(ctx, built, NULL, NULL, NULL, bvar));
        put("D_built_variance", bvar, 12);
        gingr_model_destroy(built);
        free(inst); free(coef); free(wgt); free(pm); free(gb); free(bvar);
    }

    /* ---------------------------------------------------------------- E: ICP flavours, probabilistic proposal */
    {
        const gingr_icp_params ip = {25.0, 1.0, 10};
        s0.sigma2 = 25.0;
        CHECK(gingr_fitter_set_state(f, zero, &s0));
        CHECK(gingr_fitter_update_icp_async(f, &ip, 2));
        CHECK(gingr_fitter_get_state(f, alpha, &s, fit));
        put_state("E_icp", alpha, r, &s, fit, M);
        int32_t *idx = xmalloc(4 * M);
        double *d2 = xmalloc(8 * M), *tmp = xmalloc(8 * M);
        CHECK(gingr_fitter_get_icp_idx(f, idx, d2));
        for (int64_t i = 0; i < M; ++i) tmp[i] = idx[i];
        put("E_icp_idx", tmp, M);
        /* surface correspondence */
        CHECK(gingr_fitter_set_meshes(f, Tm, mtri, Tt, ttri));
        CHECK(gingr_fitter_set_surface_method(f, 0));
        CHECK(gingr_fitter_set_correspondence_direction(f, 0));
        CHECK(gingr_fitter_set_state(f, zero, &s0));
        CHECK(gingr_fitter_icp_surface_phase_async(f, &ip, 0));
        double *cpx = xmalloc(8 * 3 * M), *w01 = xmalloc(8 * M);
        CHECK(gingr_fitter_get_surface_correspondence(f, cpx, w01));
        put("E_surface_weights", w01, M);
        put("E_surface_cp", cpx, 3 * M);
        CHECK(gingr_fitter_set_state(f, zero, &s0));
        CHECK(gingr_fitter_update_icp_surface_async(f, &ip, 1));
        CHECK(gingr_fitter_get_state(f, alpha, &s, fit));
        put_state("E_surface", alpha, r, &s, fit, M);
        double st4[4];
        CHECK(gingr_fitter_surface_distance_stats(f, 0, 0, NULL, 0, 2.0, st4));
        put("E_surface_stats", st4, 4);
        /* probabilistic proposal + log transition density (CPD flavour) */
        double *z = xmalloc(8 * r);
        for (int64_t k = 0; k < r; ++k) z[k] = 0.3 * (double)((k * 7) % 5 - 2);
        s0.sigma2 = sigma2;
        CHECK(gingr_fitter_set_state(f, zero, &s0));
        CHECK(gingr_fitter_update_cpd_sample_async(f, &cp, z));
        CHECK(gingr_fitter_get_state(f, alpha, &s, fit));
        put_state("E_sample", alpha, r, &s, fit, M);
        CHECK(gingr_fitter_set_state(f, zero, &s0));
        double lp = 0;
        CHECK(gingr_fitter_posterior_logpdf_cpd(f, &cp, fit, &lp));
        put("E_logpdf", &lp, 1);
        free(idx); free(d2); free(tmp); free(cpx); free(w01); free(z);
    }

    /* ---------------------------------------------------------------- F: classic CPD, stand-alone mesh statistics */
    {
        gingr_classic_cpd *h = NULL;
        CHECK(gingr_classic_cpd_create(ctx, 0, M, ref, N, target, 2.0, 2.0, 0.0, &h));
        CHECK(gingr_classic_cpd_iterate(h, 3));
        double s2c = 0, tr13[13];
        double *ty = xmalloc(8 * 3 * M);
        CHECK(gingr_classic_cpd_get(h, ty, &s2c, tr13, NULL));
        put("F_classic_sigma2", &s2c, 1);
        put("F_classic_ty", ty, 3 * M);
        CHECK(gingr_classic_cpd_set(h, ref, s2c));
        gingr_classic_cpd_destroy(h);
        double st4[4];
        CHECK(gingr_mesh_distance_stats(ctx, M, ref, N, target, Tt, ttri, 0, 0.0, st4));
        put("F_mesh_stats", st4, 4);
        free(ty);
    }

    /* ---------------------------------------------------------------- G: per-coordinate GPMM kernels, closest surface points,
     *                                                                      model transfer, classic rigid ICP */
    {
        const double sg[1] = {60.0}, scl[1] = {30.0};
        gingr_scalar_kernel kx, kyz, kdot;
        memset(&kx, 0, sizeof(kx));
        kx.kind = GINGR_KERNEL_GAUSSIAN_MIXTURE, kx.n_kernels = 1, kx.sigmas = sg, kx.scalings = scl, kx.mirror = -1.0;
        kyz = kx;
        kyz.mirror = 1.0;
        memset(&kdot, 0, sizeof(kdot));
        kdot.kind = GINGR_KERNEL_DOT, kdot.scaling = 0.01;
        gingr_model *sym = NULL, *dot = NULL;
        CHECK(gingr_gpmm_build_diagonal(ctx, M, ref, &kx, &kyz, &kyz, 0.0, 14, 0, M, &sym));
        if (gingr_model_rank(sym) != 14) return 17;
        double *sv = xmalloc(8 * 14);
        CHECK(gingr_model_download(ctx, sym, NULL, NULL, NULL, sv));
        put("G_sym_variance", sv, 14);
        CHECK(gingr_gpmm_build_diagonal(ctx, M, ref, &kdot, &kdot, &kdot, 0.0, 9, 0, M, &dot));
        double dv[9];
        const int32_t dr = gingr_model_rank(dot);
        CHECK(gingr_model_download(ctx, dot, NULL, NULL, NULL, dv));
        put("G_dot_variance", dv, dr);
        gingr_model_destroy(dot);
        /* closest points of the model reference on the target surface, then the symmetric model carried to 40 new points */
        double *cpp = xmalloc(8 * 3 * M), *d2 = xmalloc(8 * M), *bary = xmalloc(8 * 3 * M);
        int32_t *tid = xmalloc(4 * M);
        CHECK(gingr_mesh_closest_points(ctx, M, ref, N, target, Tt, ttri, cpp, d2, tid, bary));
        put("G_cp", cpp, 3 * M);
        put("G_bary", bary, 3 * M);
        const int64_t Mn = 40;
        int32_t *ids = xmalloc(4 * 3 * Mn);
        double *wts = xmalloc(8 * 3 * Mn), *nref = xmalloc(8 * 3 * Mn), *nb = xmalloc(8 * 3 * Mn * 14);
        for (int64_t i = 0; i < Mn; ++i) {
            for (int k = 0; k < 3; ++k) ids[3 * i + k] = (int32_t)((7 * i + 3 * k) % M);
            wts[3 * i] = 0.5, wts[3 * i + 1] = 0.3, wts[3 * i + 2] = 0.2;
            for (int d = 0; d < 3; ++d)
                nref[3 * i + d] = 0.5 * ref[3 * ids[3 * i] + d] + 0.3 * ref[3 * ids[3 * i + 1] + d] + 0.2 * ref[3 * ids[3 * i + 2] + d];
        }
        gingr_model *moved = NULL;
        CHECK(gingr_model_new_reference(ctx, sym, Mn, nref, ids, wts, 0, Mn, &moved));
        CHECK(gingr_model_download(ctx, moved, NULL, NULL, nb, NULL));
        put("G_new_basis", nb, 3 * Mn * 14);
        double *ob = xmalloc(8 * 3 * M * 14);
        CHECK(gingr_model_download(ctx, sym, NULL, NULL, ob, NULL));
        put("G_sym_basis", ob, 3 * M * 14);
        gingr_model_destroy(moved);
        gingr_model_destroy(sym);
        /* the nearest-neighbour scan with its diagnostic counter: the stateless entry point scans every pair */
        {
            int32_t *nidx = xmalloc(4 * M);
            int64_t tests = -1;
            double md = 0.0;
            CHECK(gingr_ctx_nn_counting(ctx, 1));
            CHECK(gingr_nn(ctx, M, ref, N, target, nidx, NULL, &md));
            CHECK(gingr_ctx_nn_tests(ctx, &tests));
            CHECK(gingr_ctx_nn_counting(ctx, 0));
            if (tests < M * N) return 18;          /* 64 lanes per wave: M rounded up to whole waves */
            const double nt[2] = {(double)tests, md};
            put("G_nn_tests", nt, 2);
            free(nidx);
        }
        /* two iterations of the classic rigid ICP */
        gingr_rigid_icp *icp = NULL;
        double dist[2], tr13[13];
        double *pts = xmalloc(8 * 3 * M);
        CHECK(gingr_rigid_icp_create(ctx, 0, M, ref, N, target, &icp));
        CHECK(gingr_rigid_icp_iterate(icp, 2, dist));
        CHECK(gingr_rigid_icp_get(icp, pts, tr13));
        put("G_icp_dist", dist, 2);
        put("G_icp_points", pts, 3 * M);
        CHECK(gingr_rigid_icp_set(icp, ref));
        gingr_rigid_icp_destroy(icp);
        /* one optimal-step non-rigid ICP iteration (N-ICP-T, then N-ICP-A) of the model reference against the target surface:
           correspondence of explicit points through the fitter, edges of the triangulation, the least-squares step */
        {
            int32_t *edges = xmalloc(4 * 2 * 3 * Tm);
            int64_t ne = 0;
            for (int64_t t = 0; t < Tm; ++t)
                for (int k = 0; k < 3; ++k) {
                    int32_t a = mtri[3 * t + k], b = mtri[3 * t + (k + 1) % 3];
                    if (a > b) { int32_t tmp = a; a = b; b = tmp; }
                    int dup = 0;
                    for (int64_t e = 0; e < ne && !dup; ++e) dup = edges[2 * e] == a && edges[2 * e + 1] == b;
                    if (!dup) { edges[2 * ne] = a; edges[2 * ne + 1] = b; ++ne; }
                }
            gingr_icp_params ip2 = {1.0, 1.0, 1};
            double *cps = xmalloc(8 * 3 * M), *ws = xmalloc(8 * M), *moved_t = xmalloc(8 * 3 * M), *moved_a = xmalloc(8 * 3 * M);
            const int32_t lmid[2] = {5, 17};
            double lmt[6], lmo[6];
            for (int l = 0; l < 2; ++l)
                for (int d = 0; d < 3; ++d) lmt[3 * l + d] = target[3 * (int64_t)(11 + 7 * l) + d];
            CHECK(gingr_fitter_set_surface_method(f, 0));
            CHECK(gingr_fitter_set_correspondence_direction(f, 0));
            CHECK(gingr_fitter_set_state(f, zero, &s0));
            CHECK(gingr_fitter_set_fit_points(f, ref));
            CHECK(gingr_fitter_icp_surface_phase_async(f, &ip2, 0));
            CHECK(gingr_fitter_get_surface_correspondence(f, cps, ws));
            CHECK(gingr_nicp_solve(ctx, 0, M, ref, ne, edges, ws, cps, 2, lmid, lmt, 10.0, 5.0, 1.0, moved_t, NULL));
            CHECK(gingr_nicp_solve(ctx, 1, M, ref, ne, edges, ws, cps, 2, lmid, lmt, 10.0, 5.0, 0.5, moved_a, lmo));
            put("G_nicp_w", ws, M);
            put("G_nicp_t", moved_t, 3 * M);
            put("G_nicp_a", moved_a, 3 * M);
            put("G_nicp_lm", lmo, 6);
            free(edges); free(cps); free(ws); free(moved_t); free(moved_a);
        }
        free(sv); free(cpp); free(d2); free(bary); free(tid); free(ids); free(wts); free(nref); free(nb); free(ob); free(pts);
    }

    gingr_fitter_destroy(f);
    gingr_model_destroy(model);
    gingr_ctx_destroy(ctx);
    fclose(out);
    free(ref); free(mean); free(basis); free(var); free(target); free(mtri); free(ttri); free(alpha); free(fit); free(zero);
    return 0;
}

/* Plain-C consumer of the boundary (what a cgo / JNI / FFI shim does): links libgingr_hip.so, no Python, no torch.
 * Reads "M N" and M + N points from stdin, prints the nearest-neighbour index of every query (gingr_nn) and the CPD statistics
 * Np and sigma2' (gingr_cpd_stats, sigma2 = 4, w = 0.1).  Built and run by tests/test_gpu_cabi_from_c.py. */
#include <stdio.h>
#include <stdlib.h>

#include "gingr_hip.h"

int main(void) {
    long M, N;
    if (scanf("%ld %ld", &M, &N) != 2) return 2;
    double *q = malloc(sizeof(double) * 3 * M), *t = malloc(sizeof(double) * 3 * N), *d2 = malloc(sizeof(double) * M);
    int32_t *idx = malloc(sizeof(int32_t) * M);
    for (long i = 0; i < 3 * M; ++i)
        if (scanf("%lf", &q[i]) != 1) return 2;
    for (long i = 0; i < 3 * N; ++i)
        if (scanf("%lf", &t[i]) != 1) return 2;
    gingr_ctx *ctx = NULL;
    int rc = gingr_ctx_create(0, &ctx);
    if (rc != GINGR_OK) {
        fprintf(stderr, "gingr_ctx_create: %d\n", rc);
        return 3;
    }
    double mean = 0.0;
    rc = gingr_nn(ctx, M, q, N, t, idx, d2, &mean);
    if (rc != GINGR_OK) {
        fprintf(stderr, "gingr_nn: %s\n", gingr_last_error(ctx));
        return 4;
    }
    for (long i = 0; i < M; ++i) printf("%d\n", idx[i]);
    double sc[8] = {0};
    rc = gingr_cpd_stats(ctx, M, q, N, t, 4.0, 0.1, NULL, NULL, NULL, NULL, sc);
    if (rc != GINGR_OK) {
        fprintf(stderr, "gingr_cpd_stats: %s\n", gingr_last_error(ctx));
        return 5;
    }
    printf("%.17g %.17g %.17g\n", sc[0], sc[4], mean);
    gingr_ctx_destroy(ctx);
    free(q); free(t); free(d2); free(idx);
    return 0;
}

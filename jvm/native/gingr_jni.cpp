// JNI shim: gingr.hip.GingrHipNative -> the C ABI of libgingr_hip.so (include/gingr_hip.h).
// Build (on a machine with a JDK; none exists in this repository's image, so the file is compile-guarded):
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include gingr_jni.cpp \
//       -L../../gingr_amd -lgingr_hip -Wl,-rpath,'$ORIGIN' -o libgingr_jni.so
// Arrays are pinned with GetPrimitiveArrayCritical for the duration of one C call; the library never keeps a host
// pointer after the call returns, so no JVM memory is referenced asynchronously.
#if __has_include(<jni.h>)
#include <jni.h>

#include "gingr_hip.h"

namespace {
struct Pin {  // RAII critical-section pin of a primitive array (nullptr-safe)
    JNIEnv *env;
    jarray arr;
    void *p;
    jint mode;
    Pin(JNIEnv *e, jarray a, bool readonly) : env(e), arr(a), p(a ? e->GetPrimitiveArrayCritical(a, nullptr) : nullptr),
                                                 mode(readonly ? JNI_ABORT : 0) {}
    ~Pin() {
        if (p) env->ReleasePrimitiveArrayCritical(arr, p, mode);
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};
inline jlong H(void *p) { return reinterpret_cast<jlong>(p); }
template <typename T>
inline T *P(jlong h) { return reinterpret_cast<T *>(h); }
}  // namespace

#define JFN(ret, name) extern "C" JNIEXPORT ret JNICALL Java_gingr_hip_GingrHipNative_##name

JFN(jint, deviceCount)(JNIEnv *, jclass) { return gingr_device_count(); }

JFN(jlong, ctxCreate)(JNIEnv *, jclass, jint device) {
    gingr_ctx *c = nullptr;
    return gingr_ctx_create(device, &c) == GINGR_OK ? H(c) : 0;
}
JFN(void, ctxDestroy)(JNIEnv *, jclass, jlong ctx) { gingr_ctx_destroy(P<gingr_ctx>(ctx)); }
JFN(jstring, lastError)(JNIEnv *env, jclass, jlong ctx) { return env->NewStringUTF(gingr_last_error(P<gingr_ctx>(ctx))); }

JFN(jint, cpdStats)(JNIEnv *env, jclass, jlong ctx, jdoubleArray fit, jdoubleArray target, jdouble sigma2, jdouble w,
                    jdoubleArray den, jdoubleArray p1, jdoubleArray px, jdoubleArray pt1, jdoubleArray sc) {
    const jsize M = env->GetArrayLength(fit) / 3, N = env->GetArrayLength(target) / 3;
    Pin a(env, fit, true), b(env, target, true), c(env, den, false), d(env, p1, false), e(env, px, false), f(env, pt1, false),
        g(env, sc, false);
    return gingr_cpd_stats(P<gingr_ctx>(ctx), M, a.as<double>(), N, b.as<double>(), sigma2, w, c.as<double>(), d.as<double>(),
                           e.as<double>(), f.as<double>(), g.as<double>());
}
JFN(jint, cpdInitialSigma2)(JNIEnv *env, jclass, jlong ctx, jdoubleArray ref, jdoubleArray target, jdoubleArray out) {
    const jsize M = env->GetArrayLength(ref) / 3, N = env->GetArrayLength(target) / 3;
    Pin a(env, ref, true), b(env, target, true), c(env, out, false);
    return gingr_cpd_initial_sigma2(P<gingr_ctx>(ctx), M, a.as<double>(), N, b.as<double>(), c.as<double>());
}
JFN(jint, nn)(JNIEnv *env, jclass, jlong ctx, jdoubleArray q, jdoubleArray target, jintArray idx, jdoubleArray d2, jdoubleArray md) {
    const jsize M = env->GetArrayLength(q) / 3, N = env->GetArrayLength(target) / 3;
    Pin a(env, q, true), b(env, target, true), c(env, idx, false), d(env, d2, false), e(env, md, false);
    return gingr_nn(P<gingr_ctx>(ctx), M, a.as<double>(), N, b.as<double>(), c.as<int32_t>(), d.as<double>(), e.as<double>());
}
JFN(jint, gaussBlock)(JNIEnv *env, jclass, jlong ctx, jdoubleArray A, jdoubleArray B, jdouble sigma, jdouble scaling, jdoubleArray out) {
    const jsize na = env->GetArrayLength(A) / 3, nb = env->GetArrayLength(B) / 3;
    Pin a(env, A, true), b(env, B, true), c(env, out, false);
    return gingr_gauss_block(P<gingr_ctx>(ctx), na, a.as<double>(), nb, b.as<double>(), sigma, scaling, c.as<double>());
}

JFN(jlong, modelUpload)(JNIEnv *env, jclass, jlong ctx, jlong mTotal, jint rank, jdoubleArray ref, jdoubleArray mean,
                        jdoubleArray basis, jdoubleArray variance, jlong rowBegin, jlong rowEnd) {
    Pin a(env, ref, true), b(env, mean, true), c(env, basis, true), d(env, variance, true);
    gingr_model *m = nullptr;
    const int rc = gingr_model_upload(P<gingr_ctx>(ctx), mTotal, rank, a.as<double>(), b.as<double>(), c.as<double>(),
                                      d.as<double>(), rowBegin, rowEnd, &m);
    return rc == GINGR_OK ? H(m) : 0;
}
JFN(void, modelDestroy)(JNIEnv *, jclass, jlong m) { gingr_model_destroy(P<gingr_model>(m)); }

JFN(jlong, fitterCreate)(JNIEnv *, jclass, jlong ctx, jlong model) {
    gingr_fitter *f = nullptr;
    return gingr_fitter_create(P<gingr_ctx>(ctx), P<gingr_model>(model), &f) == GINGR_OK ? H(f) : 0;
}
JFN(void, fitterDestroy)(JNIEnv *, jclass, jlong f) { gingr_fitter_destroy(P<gingr_fitter>(f)); }
JFN(jint, fitterSetTarget)(JNIEnv *env, jclass, jlong f, jdoubleArray target) {
    const jsize N = env->GetArrayLength(target) / 3;
    Pin a(env, target, true);
    return gingr_fitter_set_target(P<gingr_fitter>(f), N, a.as<double>());
}
JFN(jint, fitterSetLandmarks)(JNIEnv *env, jclass, jlong f, jintArray pid, jdoubleArray xyz, jdoubleArray cov) {
    const jsize n = pid ? env->GetArrayLength(pid) : 0;
    Pin a(env, pid, true), b(env, xyz, true), c(env, cov, true);
    return gingr_fitter_set_landmarks(P<gingr_fitter>(f), n, a.as<int32_t>(), b.as<double>(), c.as<double>());
}
JFN(jint, fitterSetOptions)(JNIEnv *, jclass, jlong f, jint gt, jdouble step) {
    return gingr_fitter_set_options(P<gingr_fitter>(f), gt, step);
}
JFN(jint, fitterSetState)(JNIEnv *env, jclass, jlong f, jdoubleArray alpha, jdoubleArray pose, jint iteration, jint status) {
    Pin a(env, alpha, true), b(env, pose, true);
    const double *p = b.as<double>();
    gingr_state_scalars s;
    for (int q = 0; q < 3; ++q) {
        s.euler[q] = p[q];
        s.center[q] = p[3 + q];
        s.translation[q] = p[6 + q];
    }
    s.scale = p[9];
    s.sigma2 = p[10];
    s.iteration = iteration;
    s.status = status;
    return gingr_fitter_set_state(P<gingr_fitter>(f), a.as<double>(), &s);
}
JFN(jint, fitterUpdateCpd)(JNIEnv *, jclass, jlong f, jdouble w, jdouble lambda, jint n) {
    gingr_cpd_params p{w, lambda};
    return gingr_fitter_update_cpd_async(P<gingr_fitter>(f), &p, n);
}
JFN(jint, fitterUpdateIcp)(JNIEnv *, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations, jint n) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    return gingr_fitter_update_icp_async(P<gingr_fitter>(f), &p, n);
}
JFN(jint, fitterUpdateCpdSample)(JNIEnv *env, jclass, jlong f, jdouble w, jdouble lambda, jdoubleArray z) {
    gingr_cpd_params p{w, lambda};
    Pin a(env, z, true);
    return gingr_fitter_update_cpd_sample_async(P<gingr_fitter>(f), &p, a.as<double>());
}
JFN(jint, fitterUpdateIcpSample)(JNIEnv *env, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations,
                                 jdoubleArray z) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    Pin a(env, z, true);
    return gingr_fitter_update_icp_sample_async(P<gingr_fitter>(f), &p, a.as<double>());
}
JFN(jint, fitterPosteriorLogpdfCpd)(JNIEnv *env, jclass, jlong f, jdouble w, jdouble lambda, jdoubleArray mesh, jdoubleArray out) {
    gingr_cpd_params p{w, lambda};
    Pin a(env, mesh, true), b(env, out, false);
    return gingr_fitter_posterior_logpdf_cpd(P<gingr_fitter>(f), &p, a.as<double>(), b.as<double>());
}
JFN(jint, fitterPosteriorLogpdfIcp)(JNIEnv *env, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations,
                                    jdoubleArray mesh, jdoubleArray out) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    Pin a(env, mesh, true), b(env, out, false);
    return gingr_fitter_posterior_logpdf_icp(P<gingr_fitter>(f), &p, a.as<double>(), b.as<double>());
}
JFN(jint, fitterGetState)(JNIEnv *env, jclass, jlong f, jdoubleArray alpha, jdoubleArray pose, jintArray iterStatus, jdoubleArray fit) {
    gingr_state_scalars s;
    int rc;
    {
        Pin a(env, alpha, false), c(env, fit, false);
        rc = gingr_fitter_get_state(P<gingr_fitter>(f), a.as<double>(), &s, c.as<double>());
    }
    if (rc != GINGR_OK) return rc;
    Pin b(env, pose, false), d(env, iterStatus, false);
    double *p = b.as<double>();
    for (int q = 0; q < 3; ++q) {
        p[q] = s.euler[q];
        p[3 + q] = s.center[q];
        p[6 + q] = s.translation[q];
    }
    p[9] = s.scale;
    p[10] = s.sigma2;
    d.as<int32_t>()[0] = s.iteration;
    d.as<int32_t>()[1] = s.status;
    return rc;
}
JFN(jint, fitterSetMeshes)(JNIEnv *env, jclass, jlong f, jintArray modelTri, jintArray targetTri) {
    const jlong nm = env->GetArrayLength(modelTri) / 3, nt = env->GetArrayLength(targetTri) / 3;
    Pin a(env, modelTri, true), b(env, targetTri, true);
    return gingr_fitter_set_meshes(P<gingr_fitter>(f), nm, a.as<int32_t>(), nt, b.as<int32_t>());
}
JFN(jint, fitterSetSurfaceMethod)(JNIEnv *, jclass, jlong f, jint method) {
    return gingr_fitter_set_surface_method(P<gingr_fitter>(f), method);
}
JFN(jint, fitterSetCorrespondenceDirection)(JNIEnv *, jclass, jlong f, jint reversed) {
    return gingr_fitter_set_correspondence_direction(P<gingr_fitter>(f), reversed);
}
JFN(jint, fitterGetReversedCorrespondence)(JNIEnv *env, jclass, jlong f, jintArray ids, jdoubleArray w) {
    Pin a(env, ids, false), b(env, w, false);
    return gingr_fitter_get_reversed_correspondence(P<gingr_fitter>(f), a.as<int32_t>(), b.as<double>());
}
JFN(jint, fitterUpdateIcpSurface)(JNIEnv *, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations, jint n) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    return gingr_fitter_update_icp_surface_async(P<gingr_fitter>(f), &p, n);
}
JFN(jint, fitterUpdateIcpSurfaceSample)(JNIEnv *env, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations,
                                        jdoubleArray z) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    Pin a(env, z, true);
    return gingr_fitter_update_icp_surface_sample_async(P<gingr_fitter>(f), &p, a.as<double>());
}
JFN(jint, fitterPosteriorLogpdfIcpSurface)(JNIEnv *env, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations,
                                           jdoubleArray mesh, jdoubleArray out) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    Pin a(env, mesh, true), b(env, out, false);
    return gingr_fitter_posterior_logpdf_icp_surface(P<gingr_fitter>(f), &p, a.as<double>(), b.as<double>());
}
JFN(jint, fitterGetSurfaceCorrespondence)(JNIEnv *env, jclass, jlong f, jdoubleArray cp, jdoubleArray w) {
    Pin a(env, cp, false), b(env, w, false);
    return gingr_fitter_get_surface_correspondence(P<gingr_fitter>(f), a.as<double>(), b.as<double>());
}
JFN(jint, fitterSurfaceDistanceStats)(JNIEnv *env, jclass, jlong f, jint direction, jlong nPoints, jdoubleArray pts, jint boundaryAware,
                                      jdouble sdev, jdoubleArray out4) {
    Pin o(env, out4, false);
    if (!pts) return gingr_fitter_surface_distance_stats(P<gingr_fitter>(f), direction, nPoints, nullptr, boundaryAware, sdev, o.as<double>());
    Pin a(env, pts, true);
    return gingr_fitter_surface_distance_stats(P<gingr_fitter>(f), direction, env->GetArrayLength(pts) / 3, a.as<double>(), boundaryAware,
                                               sdev, o.as<double>());
}
JFN(jlong, classicCpdCreate)(JNIEnv *env, jclass, jlong ctx, jint kind, jdoubleArray tmpl, jdoubleArray target, jdouble lambda,
                             jdouble beta, jdouble w) {
    const jlong m = env->GetArrayLength(tmpl) / 3, n = env->GetArrayLength(target) / 3;
    Pin a(env, tmpl, true), b(env, target, true);
    gingr_classic_cpd *h = nullptr;
    if (gingr_classic_cpd_create(P<gingr_ctx>(ctx), kind, m, a.as<double>(), n, b.as<double>(), lambda, beta, w, &h) != GINGR_OK) return 0;
    return reinterpret_cast<jlong>(h);
}
JFN(void, classicCpdDestroy)(JNIEnv *, jclass, jlong h) { gingr_classic_cpd_destroy(P<gingr_classic_cpd>(h)); }
JFN(jint, classicCpdIterate)(JNIEnv *, jclass, jlong h, jint n) { return gingr_classic_cpd_iterate(P<gingr_classic_cpd>(h), n); }
JFN(jint, classicCpdGet)(JNIEnv *env, jclass, jlong h, jdoubleArray ty, jdoubleArray s2, jdoubleArray tr, jdoubleArray w) {
    Pin a(env, ty, false), b(env, s2, false), c(env, tr, false), d(env, w, false);   // a null array pins to a null pointer
    return gingr_classic_cpd_get(P<gingr_classic_cpd>(h), a.as<double>(), b.as<double>(), c.as<double>(), d.as<double>());
}
JFN(jint, classicCpdSet)(JNIEnv *env, jclass, jlong h, jdoubleArray ty, jdouble s2) {
    Pin a(env, ty, true);
    return gingr_classic_cpd_set(P<gingr_classic_cpd>(h), a.as<double>(), s2);
}
JFN(jint, meshDistanceStats)(JNIEnv *env, jclass, jlong ctx, jdoubleArray pts, jdoubleArray verts, jintArray tris, jint boundaryAware,
                             jdouble sdev, jdoubleArray out4) {
    const jlong np = env->GetArrayLength(pts) / 3, nv = env->GetArrayLength(verts) / 3, nt = env->GetArrayLength(tris) / 3;
    Pin a(env, pts, true), b(env, verts, true), c(env, tris, true), o(env, out4, false);
    return gingr_mesh_distance_stats(P<gingr_ctx>(ctx), np, a.as<double>(), nv, b.as<double>(), nt, c.as<int32_t>(), boundaryAware, sdev,
                                     o.as<double>());
}
JFN(jlong, gpmmBuildGaussian)(JNIEnv *env, jclass, jlong ctx, jlong mTotal, jdoubleArray ref, jdoubleArray sigmas,
                               jdoubleArray scalings, jdouble relTol, jint maxRank, jlong rowBegin, jlong rowEnd) {
    const jint nk = env->GetArrayLength(sigmas);
    Pin a(env, ref, true), b(env, sigmas, true), c(env, scalings, true);
    gingr_model *m = nullptr;
    if (gingr_gpmm_build_gaussian(P<gingr_ctx>(ctx), mTotal, a.as<double>(), nk, b.as<double>(), c.as<double>(), relTol, maxRank,
                                  rowBegin, rowEnd, &m) != GINGR_OK)
        return 0;
    return reinterpret_cast<jlong>(m);
}
JFN(jint, pointsetDistanceExtrema)(JNIEnv *env, jclass, jlong ctx, jdoubleArray xyz, jdoubleArray out2) {
    const jlong n = env->GetArrayLength(xyz) / 3;
    Pin a(env, xyz, true), b(env, out2, false);
    return gingr_pointset_distance_extrema(P<gingr_ctx>(ctx), a.as<double>(), n, b.as<double>(), b.as<double>() + 1);
}
JFN(jint, modelDownload)(JNIEnv *env, jclass, jlong ctx, jlong model, jdoubleArray ref, jdoubleArray mean, jdoubleArray basis,
                         jdoubleArray variance) {
    Pin a(env, ref, false), b(env, mean, false), c(env, basis, false), d(env, variance, false);
    return gingr_model_download(P<gingr_ctx>(ctx), P<gingr_model>(model), a.as<double>(), b.as<double>(), c.as<double>(),
                                d.as<double>());
}
JFN(jint, modelRank)(JNIEnv *, jclass, jlong model) { return gingr_model_rank(P<gingr_model>(model)); }
#else
// No JDK headers on this machine: the shim is not built (the C ABI it wraps is still covered by the Python tests).
#endif

// JNI shim: gingr.hip.GingrHipNative -> the C ABI of libgingr_hip.so (include/gingr_hip.h).
// Build (on a machine with a JDK; none exists in this repository's image, so the file is compile-guarded):
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include gingr_jni.cpp \
//       -L../../gingr_amd -lgingr_hip -Wl,-rpath,'$ORIGIN' -o libgingr_jni.so
// Arrays cross the boundary by COPY (Get/Set<Type>ArrayRegion into a native staging vector), never by
// GetPrimitiveArrayCritical: every C call below uploads, launches kernels and synchronises the GPU (some also sort on the host),
// and the JNI specification forbids blocking or long-running work inside a critical region -- it would hold the GC lock of the
// whole JVM (GCLocker) for milliseconds and can stall or deadlock a host that runs several MH chains in threads.  The staging
// copy is a memcpy of at most 3 M doubles per call (1.2 MB at 50k points), small against the PCIe transfer that follows.  The
// library never keeps a host pointer after a call returns, so the staging vectors die with the call.
#if __has_include(<jni.h>)
#include <jni.h>

#include <cstdint>
#include <vector>

#include "gingr_hip.h"

namespace {
// Native staging copy of a primitive Java array (nullptr-safe).  readonly: copied in at construction.  Otherwise: sized like the
// Java array, handed to the C call as an output buffer and copied back at destruction (after the C call has returned).
template <typename T>
struct ArrTraits;
template <>
struct ArrTraits<double> {
    static void get(JNIEnv *e, jarray a, jsize n, double *p) { e->GetDoubleArrayRegion(static_cast<jdoubleArray>(a), 0, n, p); }
    static void set(JNIEnv *e, jarray a, jsize n, const double *p) { e->SetDoubleArrayRegion(static_cast<jdoubleArray>(a), 0, n, p); }
};
template <>
struct ArrTraits<int32_t> {
    static void get(JNIEnv *e, jarray a, jsize n, int32_t *p) { e->GetIntArrayRegion(static_cast<jintArray>(a), 0, n, reinterpret_cast<jint *>(p)); }
    static void set(JNIEnv *e, jarray a, jsize n, const int32_t *p) {
        e->SetIntArrayRegion(static_cast<jintArray>(a), 0, n, reinterpret_cast<const jint *>(p));
    }
};
template <typename T>
struct Arr {
    JNIEnv *env;
    jarray arr;
    bool readonly;
    std::vector<T> buf;
    Arr(JNIEnv *e, jarray a, bool ro) : env(e), arr(a), readonly(ro) {
        if (!a) return;
        const jsize n = e->GetArrayLength(a);
        buf.resize((size_t)n);
        if (ro && n > 0) ArrTraits<T>::get(e, a, n, buf.data());
    }
    Arr(const Arr &) = delete;
    Arr &operator=(const Arr &) = delete;
    ~Arr() {
        if (arr && !readonly && !buf.empty()) ArrTraits<T>::set(env, arr, (jsize)buf.size(), buf.data());
    }
    T *ptr() { return arr ? buf.data() : nullptr; }
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
    Arr<double> a(env, fit, true); Arr<double> b(env, target, true); Arr<double> c(env, den, false); Arr<double> d(env, p1, false); Arr<double> e(env, px, false); Arr<double> f(env, pt1, false); Arr<double> g(env, sc, false);
    return gingr_cpd_stats(P<gingr_ctx>(ctx), M, a.ptr(), N, b.ptr(), sigma2, w, c.ptr(), d.ptr(),
                           e.ptr(), f.ptr(), g.ptr());
}
JFN(jint, cpdInitialSigma2)(JNIEnv *env, jclass, jlong ctx, jdoubleArray ref, jdoubleArray target, jdoubleArray out) {
    const jsize M = env->GetArrayLength(ref) / 3, N = env->GetArrayLength(target) / 3;
    Arr<double> a(env, ref, true); Arr<double> b(env, target, true); Arr<double> c(env, out, false);
    return gingr_cpd_initial_sigma2(P<gingr_ctx>(ctx), M, a.ptr(), N, b.ptr(), c.ptr());
}
JFN(jint, nn)(JNIEnv *env, jclass, jlong ctx, jdoubleArray q, jdoubleArray target, jintArray idx, jdoubleArray d2, jdoubleArray md) {
    const jsize M = env->GetArrayLength(q) / 3, N = env->GetArrayLength(target) / 3;
    Arr<double> a(env, q, true); Arr<double> b(env, target, true); Arr<int32_t> c(env, idx, false); Arr<double> d(env, d2, false); Arr<double> e(env, md, false);
    return gingr_nn(P<gingr_ctx>(ctx), M, a.ptr(), N, b.ptr(), c.ptr(), d.ptr(), e.ptr());
}
JFN(jint, gaussBlock)(JNIEnv *env, jclass, jlong ctx, jdoubleArray A, jdoubleArray B, jdouble sigma, jdouble scaling, jdoubleArray out) {
    const jsize na = env->GetArrayLength(A) / 3, nb = env->GetArrayLength(B) / 3;
    Arr<double> a(env, A, true); Arr<double> b(env, B, true); Arr<double> c(env, out, false);
    return gingr_gauss_block(P<gingr_ctx>(ctx), na, a.ptr(), nb, b.ptr(), sigma, scaling, c.ptr());
}

JFN(jlong, modelUpload)(JNIEnv *env, jclass, jlong ctx, jlong mTotal, jint rank, jdoubleArray ref, jdoubleArray mean,
                        jdoubleArray basis, jdoubleArray variance, jlong rowBegin, jlong rowEnd) {
    Arr<double> a(env, ref, true); Arr<double> b(env, mean, true); Arr<double> c(env, basis, true); Arr<double> d(env, variance, true);
    gingr_model *m = nullptr;
    const int rc = gingr_model_upload(P<gingr_ctx>(ctx), mTotal, rank, a.ptr(), b.ptr(), c.ptr(),
                                      d.ptr(), rowBegin, rowEnd, &m);
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
    Arr<double> a(env, target, true);
    return gingr_fitter_set_target(P<gingr_fitter>(f), N, a.ptr());
}
JFN(jint, fitterSetLandmarks)(JNIEnv *env, jclass, jlong f, jintArray pid, jdoubleArray xyz, jdoubleArray cov) {
    const jsize n = pid ? env->GetArrayLength(pid) : 0;
    Arr<int32_t> a(env, pid, true); Arr<double> b(env, xyz, true); Arr<double> c(env, cov, true);
    return gingr_fitter_set_landmarks(P<gingr_fitter>(f), n, a.ptr(), b.ptr(), c.ptr());
}
JFN(jint, fitterSetOptions)(JNIEnv *, jclass, jlong f, jint gt, jdouble step) {
    return gingr_fitter_set_options(P<gingr_fitter>(f), gt, step);
}
JFN(jint, fitterSetStopThreshold)(JNIEnv *, jclass, jlong f, jdouble threshold) {
    return gingr_fitter_set_stop_threshold(P<gingr_fitter>(f), threshold);
}
JFN(jint, fitterStopRuleHit)(JNIEnv *, jclass, jlong f) {  // >= 0: the mark; < 0: -(error code)
    int32_t hit = 0;
    const int rc = gingr_fitter_stop_rule_hit(P<gingr_fitter>(f), &hit);
    return rc == GINGR_OK ? hit : -rc;
}
JFN(jint, fitterSetState)(JNIEnv *env, jclass, jlong f, jdoubleArray alpha, jdoubleArray pose, jint iteration, jint status) {
    Arr<double> a(env, alpha, true); Arr<double> b(env, pose, true);
    const double *p = b.ptr();
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
    return gingr_fitter_set_state(P<gingr_fitter>(f), a.ptr(), &s);
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
    Arr<double> a(env, z, true);
    return gingr_fitter_update_cpd_sample_async(P<gingr_fitter>(f), &p, a.ptr());
}
JFN(jint, fitterUpdateIcpSample)(JNIEnv *env, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations,
                                 jdoubleArray z) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    Arr<double> a(env, z, true);
    return gingr_fitter_update_icp_sample_async(P<gingr_fitter>(f), &p, a.ptr());
}
JFN(jint, fitterPosteriorLogpdfCpd)(JNIEnv *env, jclass, jlong f, jdouble w, jdouble lambda, jdoubleArray mesh, jdoubleArray out) {
    gingr_cpd_params p{w, lambda};
    Arr<double> a(env, mesh, true); Arr<double> b(env, out, false);
    return gingr_fitter_posterior_logpdf_cpd(P<gingr_fitter>(f), &p, a.ptr(), b.ptr());
}
JFN(jint, fitterPosteriorLogpdfIcp)(JNIEnv *env, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations,
                                    jdoubleArray mesh, jdoubleArray out) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    Arr<double> a(env, mesh, true); Arr<double> b(env, out, false);
    return gingr_fitter_posterior_logpdf_icp(P<gingr_fitter>(f), &p, a.ptr(), b.ptr());
}
JFN(jint, fitterGetState)(JNIEnv *env, jclass, jlong f, jdoubleArray alpha, jdoubleArray pose, jintArray iterStatus, jdoubleArray fit) {
    gingr_state_scalars s;
    int rc;
    {
        Arr<double> a(env, alpha, false); Arr<double> c(env, fit, false);
        rc = gingr_fitter_get_state(P<gingr_fitter>(f), a.ptr(), &s, c.ptr());
    }
    if (rc != GINGR_OK) return rc;
    Arr<double> b(env, pose, false); Arr<int32_t> d(env, iterStatus, false);
    double *p = b.ptr();
    for (int q = 0; q < 3; ++q) {
        p[q] = s.euler[q];
        p[3 + q] = s.center[q];
        p[6 + q] = s.translation[q];
    }
    p[9] = s.scale;
    p[10] = s.sigma2;
    d.ptr()[0] = s.iteration;
    d.ptr()[1] = s.status;
    return rc;
}
JFN(jint, fitterSetMeshes)(JNIEnv *env, jclass, jlong f, jintArray modelTri, jintArray targetTri) {
    const jlong nm = env->GetArrayLength(modelTri) / 3, nt = env->GetArrayLength(targetTri) / 3;
    Arr<int32_t> a(env, modelTri, true); Arr<int32_t> b(env, targetTri, true);
    return gingr_fitter_set_meshes(P<gingr_fitter>(f), nm, a.ptr(), nt, b.ptr());
}
JFN(jint, fitterSetSurfaceMethod)(JNIEnv *, jclass, jlong f, jint method) {
    return gingr_fitter_set_surface_method(P<gingr_fitter>(f), method);
}
JFN(jint, fitterSetCorrespondenceDirection)(JNIEnv *, jclass, jlong f, jint reversed) {
    return gingr_fitter_set_correspondence_direction(P<gingr_fitter>(f), reversed);
}
JFN(jint, fitterGetReversedCorrespondence)(JNIEnv *env, jclass, jlong f, jintArray ids, jdoubleArray w) {
    Arr<int32_t> a(env, ids, false); Arr<double> b(env, w, false);
    return gingr_fitter_get_reversed_correspondence(P<gingr_fitter>(f), a.ptr(), b.ptr());
}
JFN(jint, fitterUpdateIcpSurface)(JNIEnv *, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations, jint n) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    return gingr_fitter_update_icp_surface_async(P<gingr_fitter>(f), &p, n);
}
JFN(jint, fitterUpdateIcpSurfaceSample)(JNIEnv *env, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations,
                                        jdoubleArray z) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    Arr<double> a(env, z, true);
    return gingr_fitter_update_icp_surface_sample_async(P<gingr_fitter>(f), &p, a.ptr());
}
JFN(jint, fitterPosteriorLogpdfIcpSurface)(JNIEnv *env, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations,
                                           jdoubleArray mesh, jdoubleArray out) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    Arr<double> a(env, mesh, true); Arr<double> b(env, out, false);
    return gingr_fitter_posterior_logpdf_icp_surface(P<gingr_fitter>(f), &p, a.ptr(), b.ptr());
}
JFN(jint, fitterGetSurfaceCorrespondence)(JNIEnv *env, jclass, jlong f, jdoubleArray cp, jdoubleArray w) {
    Arr<double> a(env, cp, false); Arr<double> b(env, w, false);
    return gingr_fitter_get_surface_correspondence(P<gingr_fitter>(f), a.ptr(), b.ptr());
}
JFN(jint, fitterSurfaceDistanceStats)(JNIEnv *env, jclass, jlong f, jint direction, jlong nPoints, jdoubleArray pts, jint boundaryAware,
                                      jdouble sdev, jdoubleArray out4) {
    Arr<double> o(env, out4, false);
    if (!pts) return gingr_fitter_surface_distance_stats(P<gingr_fitter>(f), direction, nPoints, nullptr, boundaryAware, sdev, o.ptr());
    Arr<double> a(env, pts, true);
    return gingr_fitter_surface_distance_stats(P<gingr_fitter>(f), direction, env->GetArrayLength(pts) / 3, a.ptr(), boundaryAware,
                                               sdev, o.ptr());
}
JFN(jlong, classicCpdCreate)(JNIEnv *env, jclass, jlong ctx, jint kind, jdoubleArray tmpl, jdoubleArray target, jdouble lambda,
                             jdouble beta, jdouble w) {
    const jlong m = env->GetArrayLength(tmpl) / 3, n = env->GetArrayLength(target) / 3;
    Arr<double> a(env, tmpl, true); Arr<double> b(env, target, true);
    gingr_classic_cpd *h = nullptr;
    if (gingr_classic_cpd_create(P<gingr_ctx>(ctx), kind, m, a.ptr(), n, b.ptr(), lambda, beta, w, &h) != GINGR_OK) return 0;
    return reinterpret_cast<jlong>(h);
}
JFN(void, classicCpdDestroy)(JNIEnv *, jclass, jlong h) { gingr_classic_cpd_destroy(P<gingr_classic_cpd>(h)); }
JFN(jint, classicCpdIterate)(JNIEnv *, jclass, jlong h, jint n) { return gingr_classic_cpd_iterate(P<gingr_classic_cpd>(h), n); }
JFN(jint, classicCpdGet)(JNIEnv *env, jclass, jlong h, jdoubleArray ty, jdoubleArray s2, jdoubleArray tr, jdoubleArray w) {
    Arr<double> a(env, ty, false); Arr<double> b(env, s2, false); Arr<double> c(env, tr, false); Arr<double> d(env, w, false);   // a null array gives a null pointer
    return gingr_classic_cpd_get(P<gingr_classic_cpd>(h), a.ptr(), b.ptr(), c.ptr(), d.ptr());
}
JFN(jint, classicCpdSet)(JNIEnv *env, jclass, jlong h, jdoubleArray ty, jdouble s2) {
    Arr<double> a(env, ty, true);
    return gingr_classic_cpd_set(P<gingr_classic_cpd>(h), a.ptr(), s2);
}
JFN(jint, meshDistanceStats)(JNIEnv *env, jclass, jlong ctx, jdoubleArray pts, jdoubleArray verts, jintArray tris, jint boundaryAware,
                             jdouble sdev, jdoubleArray out4) {
    const jlong np = env->GetArrayLength(pts) / 3, nv = env->GetArrayLength(verts) / 3, nt = env->GetArrayLength(tris) / 3;
    Arr<double> a(env, pts, true); Arr<double> b(env, verts, true); Arr<int32_t> c(env, tris, true); Arr<double> o(env, out4, false);
    return gingr_mesh_distance_stats(P<gingr_ctx>(ctx), np, a.ptr(), nv, b.ptr(), nt, c.ptr(), boundaryAware, sdev,
                                     o.ptr());
}
JFN(jlong, gpmmBuildGaussian)(JNIEnv *env, jclass, jlong ctx, jlong mTotal, jdoubleArray ref, jdoubleArray sigmas,
                               jdoubleArray scalings, jdouble relTol, jint maxRank, jlong rowBegin, jlong rowEnd) {
    const jint nk = env->GetArrayLength(sigmas);
    Arr<double> a(env, ref, true); Arr<double> b(env, sigmas, true); Arr<double> c(env, scalings, true);
    gingr_model *m = nullptr;
    if (gingr_gpmm_build_gaussian(P<gingr_ctx>(ctx), mTotal, a.ptr(), nk, b.ptr(), c.ptr(), relTol, maxRank,
                                  rowBegin, rowEnd, &m) != GINGR_OK)
        return 0;
    return reinterpret_cast<jlong>(m);
}
JFN(jlong, gpmmBuildDiagonal)(JNIEnv *env, jclass, jlong ctx, jlong mTotal, jdoubleArray ref, jintArray kind3, jobjectArray sigmas3,
                               jobjectArray scalings3, jdoubleArray mirror3, jdoubleArray scaling3, jobjectArray lookup3, jdouble relTol,
                               jint maxRank, jlong rowBegin, jlong rowEnd) {
    Arr<double> a(env, ref, true); Arr<int32_t> kd(env, kind3, true); Arr<double> mi(env, mirror3, true); Arr<double> sc(env, scaling3, true);
    if (kd.buf.size() != 3 || mi.buf.size() != 3 || sc.buf.size() != 3) return 0;
    std::vector<double> sig[3], scl[3], lut[3];
    gingr_scalar_kernel k[3];
    auto copy = [&](jobjectArray outer, int d, std::vector<double> &dst) {
        if (!outer) return;
        jdoubleArray inner = static_cast<jdoubleArray>(env->GetObjectArrayElement(outer, d));
        if (!inner) return;
        const jsize n = env->GetArrayLength(inner);
        dst.resize((size_t)n);
        if (n > 0) env->GetDoubleArrayRegion(inner, 0, n, dst.data());
        env->DeleteLocalRef(inner);
    };
    for (int d = 0; d < 3; ++d) {
        copy(sigmas3, d, sig[d]);
        copy(scalings3, d, scl[d]);
        copy(lookup3, d, lut[d]);
        k[d].kind = kd.buf[(size_t)d];
        k[d].n_kernels = (int32_t)sig[d].size();
        k[d].sigmas = sig[d].empty() ? nullptr : sig[d].data();
        k[d].scalings = scl[d].empty() ? nullptr : scl[d].data();
        k[d].mirror = mi.buf[(size_t)d];
        k[d].scaling = sc.buf[(size_t)d];
        k[d].lookup = lut[d].empty() ? nullptr : lut[d].data();
    }
    // equal kernels -> one factorisation: the library compares contents, but lookup tables by pointer
    for (int d = 1; d < 3; ++d)
        for (int e = 0; e < d; ++e)
            if (k[d].kind == GINGR_KERNEL_LOOKUP && k[e].kind == GINGR_KERNEL_LOOKUP && lut[d] == lut[e]) k[d].lookup = k[e].lookup;
    gingr_model *m = nullptr;
    if (gingr_gpmm_build_diagonal(P<gingr_ctx>(ctx), mTotal, a.ptr(), &k[0], &k[1], &k[2], relTol, maxRank, rowBegin, rowEnd, &m) !=
        GINGR_OK)
        return 0;
    return reinterpret_cast<jlong>(m);
}
JFN(jint, meshClosestPoints)(JNIEnv *env, jclass, jlong ctx, jdoubleArray pts, jdoubleArray verts, jintArray tris, jdoubleArray cp,
                              jdoubleArray d2, jintArray triId, jdoubleArray bary) {
    Arr<double> a(env, pts, true); Arr<double> b(env, verts, true); Arr<int32_t> c(env, tris, true);
    Arr<double> o1(env, cp, false); Arr<double> o2(env, d2, false); Arr<int32_t> o3(env, triId, false); Arr<double> o4(env, bary, false);
    return gingr_mesh_closest_points(P<gingr_ctx>(ctx), (int64_t)a.buf.size() / 3, a.ptr(), (int64_t)b.buf.size() / 3, b.ptr(),
                                     (int64_t)c.buf.size() / 3, c.ptr(), o1.ptr(), o2.ptr(), o3.ptr(), o4.ptr());
}
JFN(jlong, modelNewReference)(JNIEnv *env, jclass, jlong ctx, jlong src, jdoubleArray newRef, jintArray ids, jdoubleArray weights,
                               jlong rowBegin, jlong rowEnd) {
    Arr<double> a(env, newRef, true); Arr<int32_t> b(env, ids, true); Arr<double> c(env, weights, true);
    gingr_model *m = nullptr;
    if (gingr_model_new_reference(P<gingr_ctx>(ctx), P<gingr_model>(src), (int64_t)a.buf.size() / 3, a.ptr(), b.ptr(), c.ptr(), rowBegin,
                                  rowEnd, &m) != GINGR_OK)
        return 0;
    return reinterpret_cast<jlong>(m);
}
JFN(jlong, rigidIcpCreate)(JNIEnv *env, jclass, jlong ctx, jint kind, jdoubleArray tpl, jdoubleArray target) {
    Arr<double> a(env, tpl, true); Arr<double> b(env, target, true);
    gingr_rigid_icp *h = nullptr;
    if (gingr_rigid_icp_create(P<gingr_ctx>(ctx), kind, (int64_t)a.buf.size() / 3, a.ptr(), (int64_t)b.buf.size() / 3, b.ptr(), &h) != GINGR_OK)
        return 0;
    return reinterpret_cast<jlong>(h);
}
JFN(void, rigidIcpDestroy)(JNIEnv *, jclass, jlong h) { gingr_rigid_icp_destroy(P<gingr_rigid_icp>(h)); }
JFN(jint, rigidIcpIterate)(JNIEnv *env, jclass, jlong h, jint n, jdoubleArray distances) {
    Arr<double> d(env, distances, false);
    return gingr_rigid_icp_iterate(P<gingr_rigid_icp>(h), n, d.ptr());
}
JFN(jint, rigidIcpGet)(JNIEnv *env, jclass, jlong h, jdoubleArray pts, jdoubleArray tr13) {
    Arr<double> a(env, pts, false); Arr<double> b(env, tr13, false);
    return gingr_rigid_icp_get(P<gingr_rigid_icp>(h), a.ptr(), b.ptr());
}
JFN(jint, rigidIcpSet)(JNIEnv *env, jclass, jlong h, jdoubleArray pts) {
    Arr<double> a(env, pts, true);
    return gingr_rigid_icp_set(P<gingr_rigid_icp>(h), a.ptr());
}
JFN(jint, fitterSetFitPoints)(JNIEnv *env, jclass, jlong f, jdoubleArray pts) {
    Arr<double> a(env, pts, true);
    return gingr_fitter_set_fit_points(P<gingr_fitter>(f), a.ptr());
}
JFN(jint, fitterIcpSurfacePhase)(JNIEnv *, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations, jint phase) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    return gingr_fitter_icp_surface_phase_async(P<gingr_fitter>(f), &p, phase);
}
JFN(jint, nicpSolve)(JNIEnv *env, jclass, jlong ctx, jint kind, jdoubleArray tpl, jintArray edges, jdoubleArray w, jdoubleArray cp,
                     jintArray lmIds, jdoubleArray lmTargets, jdouble alpha, jdouble beta, jdouble gamma, jdoubleArray out, jdoubleArray outLm) {
    const jlong n = env->GetArrayLength(tpl) / 3, ne = edges ? env->GetArrayLength(edges) / 2 : 0;
    const jint nl = lmIds ? env->GetArrayLength(lmIds) : 0;
    Arr<double> a(env, tpl, true); Arr<int32_t> e(env, edges, true); Arr<double> ww(env, w, true); Arr<double> c(env, cp, true);
    Arr<int32_t> li(env, lmIds, true); Arr<double> lt(env, lmTargets, true); Arr<double> o(env, out, false); Arr<double> ol(env, outLm, false);
    return gingr_nicp_solve(P<gingr_ctx>(ctx), kind, n, a.ptr(), ne, e.ptr(), ww.ptr(), c.ptr(), nl, li.ptr(), lt.ptr(), alpha, beta, gamma,
                            o.ptr(), ol.ptr());
}
JFN(jint, pointsetDistanceExtrema)(JNIEnv *env, jclass, jlong ctx, jdoubleArray xyz, jdoubleArray out2) {
    const jlong n = env->GetArrayLength(xyz) / 3;
    Arr<double> a(env, xyz, true); Arr<double> b(env, out2, false);
    return gingr_pointset_distance_extrema(P<gingr_ctx>(ctx), a.ptr(), n, b.ptr(), b.ptr() + 1);
}
JFN(jint, modelDownload)(JNIEnv *env, jclass, jlong ctx, jlong model, jdoubleArray ref, jdoubleArray mean, jdoubleArray basis,
                         jdoubleArray variance) {
    Arr<double> a(env, ref, false); Arr<double> b(env, mean, false); Arr<double> c(env, basis, false); Arr<double> d(env, variance, false);
    return gingr_model_download(P<gingr_ctx>(ctx), P<gingr_model>(model), a.ptr(), b.ptr(), c.ptr(),
                                d.ptr());
}
JFN(jint, modelRank)(JNIEnv *, jclass, jlong model) { return gingr_model_rank(P<gingr_model>(model)); }

JFN(jint, fitterRetryCounter)(JNIEnv *env, jclass, jlong f, jint setTo, jintArray out1) {
    Arr<int32_t> o(env, out1, false);
    return gingr_fitter_retry_counter(P<gingr_fitter>(f), setTo, o.ptr());
}

// ---- device group (gingr_group_*): multi-GPU from the JVM process itself
namespace {
void scalars_from(const double *p, jint iteration, jint status, gingr_state_scalars *s) {
    for (int q = 0; q < 3; ++q) {
        s->euler[q] = p[q];
        s->center[q] = p[3 + q];
        s->translation[q] = p[6 + q];
    }
    s->scale = p[9];
    s->sigma2 = p[10];
    s->iteration = iteration;
    s->status = status;
}
}  // namespace
JFN(jlong, groupCreate)(JNIEnv *env, jclass, jintArray devices) {
    Arr<int32_t> d(env, devices, true);
    gingr_group *g = nullptr;
    return gingr_group_create((int32_t)d.buf.size(), d.ptr(), &g) == GINGR_OK ? H(g) : 0;
}
JFN(void, groupDestroy)(JNIEnv *, jclass, jlong g) { gingr_group_destroy(P<gingr_group>(g)); }
JFN(jstring, groupLastError)(JNIEnv *env, jclass, jlong g) { return env->NewStringUTF(gingr_group_last_error(P<gingr_group>(g))); }
JFN(jint, groupModelUpload)(JNIEnv *env, jclass, jlong g, jlong mTotal, jint rank, jdoubleArray ref, jdoubleArray mean,
                            jdoubleArray basis, jdoubleArray variance) {
    Arr<double> a(env, ref, true); Arr<double> b(env, mean, true); Arr<double> c(env, basis, true); Arr<double> d(env, variance, true);
    return gingr_group_model_upload(P<gingr_group>(g), mTotal, rank, a.ptr(), b.ptr(), c.ptr(), d.ptr());
}
JFN(jint, groupGpmmBuildGaussian)(JNIEnv *env, jclass, jlong g, jlong mTotal, jdoubleArray ref, jdoubleArray sigmas,
                                  jdoubleArray scalings, jdouble relTol, jint maxRank) {
    Arr<double> a(env, ref, true); Arr<double> b(env, sigmas, true); Arr<double> c(env, scalings, true);
    return gingr_group_gpmm_build_gaussian(P<gingr_group>(g), mTotal, a.ptr(), (int32_t)b.buf.size(), b.ptr(), c.ptr(), relTol, maxRank);
}
JFN(jint, groupSetTarget)(JNIEnv *env, jclass, jlong g, jdoubleArray target) {
    Arr<double> a(env, target, true);
    return gingr_group_set_target(P<gingr_group>(g), (int64_t)a.buf.size() / 3, a.ptr());
}
JFN(jint, groupSetLandmarks)(JNIEnv *env, jclass, jlong g, jintArray pid, jdoubleArray xyz, jdoubleArray cov) {
    Arr<int32_t> a(env, pid, true); Arr<double> b(env, xyz, true); Arr<double> c(env, cov, true);
    return gingr_group_set_landmarks(P<gingr_group>(g), (int32_t)a.buf.size(), a.ptr(), b.ptr(), c.ptr());
}
JFN(jint, groupSetOptions)(JNIEnv *, jclass, jlong g, jint gt, jdouble step) {
    return gingr_group_set_options(P<gingr_group>(g), gt, step);
}
JFN(jint, groupSetState)(JNIEnv *env, jclass, jlong g, jdoubleArray alpha, jdoubleArray pose, jint iteration, jint status) {
    Arr<double> a(env, alpha, true); Arr<double> b(env, pose, true);
    gingr_state_scalars s;
    scalars_from(b.ptr(), iteration, status, &s);
    return gingr_group_set_state(P<gingr_group>(g), a.ptr(), &s);
}
JFN(jint, groupGetState)(JNIEnv *env, jclass, jlong g, jdoubleArray alpha, jdoubleArray pose, jintArray iterStatus, jdoubleArray fit) {
    gingr_state_scalars s;
    Arr<double> a(env, alpha, false); Arr<double> b(env, pose, false); Arr<int32_t> d(env, iterStatus, false); Arr<double> c(env, fit, false);
    const int rc = gingr_group_get_state(P<gingr_group>(g), a.ptr(), &s, c.ptr());
    if (rc != GINGR_OK) return rc;
    double *p = b.ptr();
    for (int q = 0; q < 3; ++q) {
        p[q] = s.euler[q];
        p[3 + q] = s.center[q];
        p[6 + q] = s.translation[q];
    }
    p[9] = s.scale;
    p[10] = s.sigma2;
    d.ptr()[0] = s.iteration;
    d.ptr()[1] = s.status;
    return rc;
}
JFN(jint, groupUpdateCpd)(JNIEnv *, jclass, jlong g, jdouble w, jdouble lambda, jint n) {
    gingr_cpd_params p{w, lambda};
    return gingr_group_update_cpd_async(P<gingr_group>(g), &p, n);
}
JFN(jint, groupUpdateIcp)(JNIEnv *, jclass, jlong g, jdouble initialSigma, jdouble endSigma, jint maxIterations, jint n) {
    gingr_icp_params p{initialSigma, endSigma, maxIterations};
    return gingr_group_update_icp_async(P<gingr_group>(g), &p, n);
}
JFN(jint, groupSynchronize)(JNIEnv *, jclass, jlong g) { return gingr_group_synchronize(P<gingr_group>(g)); }
// out2 = {physical devices behind the shards, 1 if the peer-read send buffers are fine-grained}: what a first multi-GPU run is read by
JFN(jint, groupExchangeInfo)(JNIEnv *env, jclass, jlong g, jintArray out2) {
    int32_t nd = 0, fg = 0;
    const int rc = gingr_group_exchange_info(P<gingr_group>(g), &nd, &fg);
    if (out2 && env->GetArrayLength(out2) >= 2) {
        const jint v[2] = {nd, fg};
        env->SetIntArrayRegion(out2, 0, 2, v);
    }
    return rc;
}

// ---- native RCCL exchange (gingr_ctx_rccl_*): one JVM process per GPU, the library enqueues the collectives itself.  The 128 bytes
// of the ncclUniqueId travel as int[32]; how rank 0's id reaches the other JVMs is the host's business (a socket, a file, MPI ...)
JFN(jint, rcclUniqueId)(JNIEnv *env, jclass, jlong ctx, jintArray id32) {
    Arr<int32_t> id(env, id32, false);
    if (id.buf.size() * sizeof(int32_t) < GINGR_RCCL_UNIQUE_ID_BYTES) return GINGR_ERR_BAD_ARGUMENT;
    return gingr_rccl_unique_id(P<gingr_ctx>(ctx), id.ptr());
}
JFN(jint, ctxRcclInit)(JNIEnv *env, jclass, jlong ctx, jintArray id32, jint world, jint rank) {
    Arr<int32_t> id(env, id32, true);
    if (id.buf.size() * sizeof(int32_t) < GINGR_RCCL_UNIQUE_ID_BYTES) return GINGR_ERR_BAD_ARGUMENT;
    return gingr_ctx_rccl_init(P<gingr_ctx>(ctx), id.ptr(), world, rank);
}
// the one-off sum of the basis moments of a row-sharded model across the ranks (before modelFinalize)
JFN(jint, ctxRcclAllreduceModelMoments)(JNIEnv *, jclass, jlong ctx, jlong model) {
    void *p = nullptr;
    int64_t n = 0;
    const int rc = gingr_model_gram_exchange(P<gingr_model>(model), &p, &n);
    if (rc != GINGR_OK) return rc;
    const int rc2 = gingr_ctx_rccl_allreduce_async(P<gingr_ctx>(ctx), p, n);
    return rc2 != GINGR_OK ? rc2 : gingr_ctx_synchronize(P<gingr_ctx>(ctx));
}
JFN(jint, fitterUpdateCpdRccl)(JNIEnv *, jclass, jlong f, jdouble w, jdouble lambda, jint n) {
    const gingr_cpd_params p{w, lambda};
    return gingr_fitter_update_cpd_rccl_async(P<gingr_fitter>(f), &p, n);
}
JFN(jint, fitterUpdateIcpRccl)(JNIEnv *, jclass, jlong f, jdouble initialSigma, jdouble endSigma, jint maxIterations, jint n) {
    const gingr_icp_params p{initialSigma, endSigma, maxIterations};
    return gingr_fitter_update_icp_rccl_async(P<gingr_fitter>(f), &p, n);
}
JFN(jint, ctxSetOption)(JNIEnv *, jclass, jlong ctx, jint option, jint value) { return gingr_ctx_set_option(P<gingr_ctx>(ctx), option, value); }
// ---- round 4: every flavour of the update (0 CPD, 1 ICP point cloud, 2 ICP surface), the sampled proposal (z = r standard normals,
// null for the mean update) and the transition density on ROW SHARDS -- through the device group and through the context's RCCL
// communicator (gingr_group_update_async / _posterior_logpdf, gingr_fitter_update_rccl_async / _posterior_logpdf_rccl)
JFN(jint, groupSetMeshes)(JNIEnv *env, jclass, jlong g, jintArray modelTriangles, jintArray targetTriangles) {
    Arr<int32_t> a(env, modelTriangles, true), b(env, targetTriangles, true);
    return gingr_group_set_meshes(P<gingr_group>(g), (int64_t)a.buf.size() / 3, a.ptr(), (int64_t)b.buf.size() / 3, b.ptr());
}
JFN(jint, groupSetSurfaceMethod)(JNIEnv *, jclass, jlong g, jint method) { return gingr_group_set_surface_method(P<gingr_group>(g), method); }
JFN(jint, groupSetCorrespondenceDirection)(JNIEnv *, jclass, jlong g, jboolean reversed) {
    return gingr_group_set_correspondence_direction(P<gingr_group>(g), reversed ? 1 : 0);
}
JFN(jint, groupUpdate)(JNIEnv *env, jclass, jlong g, jint flavour, jdouble w, jdouble lambda, jdouble initialSigma, jdouble endSigma,
                       jint maxIterations, jint n, jdoubleArray z) {
    const gingr_cpd_params cp{w, lambda};
    const gingr_icp_params ip{initialSigma, endSigma, maxIterations};
    Arr<double> zz(env, z, true);
    return gingr_group_update_async(P<gingr_group>(g), flavour, &cp, &ip, n, zz.ptr());
}
JFN(jint, groupPosteriorLogpdf)(JNIEnv *env, jclass, jlong g, jint flavour, jdouble w, jdouble lambda, jdouble initialSigma, jdouble endSigma,
                                jint maxIterations, jdoubleArray meshXyzFull, jdoubleArray out1) {
    const gingr_cpd_params cp{w, lambda};
    const gingr_icp_params ip{initialSigma, endSigma, maxIterations};
    Arr<double> m(env, meshXyzFull, true), o(env, out1, false);
    if (o.buf.empty()) return GINGR_ERR_BAD_ARGUMENT;
    return gingr_group_posterior_logpdf(P<gingr_group>(g), flavour, &cp, &ip, m.ptr(), o.ptr());
}
JFN(jint, fitterUpdateRccl)(JNIEnv *env, jclass, jlong f, jint flavour, jdouble w, jdouble lambda, jdouble initialSigma, jdouble endSigma,
                            jint maxIterations, jint n, jdoubleArray z) {
    const gingr_cpd_params cp{w, lambda};
    const gingr_icp_params ip{initialSigma, endSigma, maxIterations};
    Arr<double> zz(env, z, true);
    return gingr_fitter_update_rccl_async(P<gingr_fitter>(f), flavour, &cp, &ip, n, zz.ptr());
}
JFN(jint, fitterPosteriorLogpdfRccl)(JNIEnv *env, jclass, jlong f, jint flavour, jdouble w, jdouble lambda, jdouble initialSigma,
                                     jdouble endSigma, jint maxIterations, jdoubleArray meshXyzFull, jdoubleArray out1) {
    const gingr_cpd_params cp{w, lambda};
    const gingr_icp_params ip{initialSigma, endSigma, maxIterations};
    Arr<double> m(env, meshXyzFull, true), o(env, out1, false);
    if (o.buf.empty()) return GINGR_ERR_BAD_ARGUMENT;
    return gingr_fitter_posterior_logpdf_rccl(P<gingr_fitter>(f), flavour, &cp, &ip, m.ptr(), o.ptr());
}
// ---- one Metropolis-Hastings step per call (gingr_fitter_mh_step): flavour as above; kind 0 = informed proposal with the r draws z,
// kind 1 = the parameters a random-walk proposal chose (alpha, pose11 = euler, center, translation, scale, sigma2; iteration / status).
// alphaOut[r], fitOut[3 M] (nullable), poseOut[11], intOut = {iteration, status, forward status, backward status},
// dblOut = {log value, dist sum, dist max, count, log q forward, log q backward}
JFN(jint, fitterMhStep)(JNIEnv *env, jclass, jlong f, jint flavour, jint kind, jdouble w, jdouble lambda, jdouble initialSigma, jdouble endSigma,
                        jint maxIterations, jdoubleArray z, jdoubleArray alpha, jdoubleArray pose11, jint iteration, jint status,
                        jdouble evalSdev, jlong evalPoints, jboolean needForward, jdoubleArray alphaOut, jdoubleArray fitOut,
                        jdoubleArray poseOut, jintArray intOut, jdoubleArray dblOut) {
    const gingr_cpd_params cp{w, lambda};
    const gingr_icp_params ip{initialSigma, endSigma, maxIterations};
    Arr<double> zz(env, z, true), aa(env, alpha, true), pp(env, pose11, true);
    Arr<double> ao(env, alphaOut, false), fo(env, fitOut, false), po(env, poseOut, false), d(env, dblOut, false);
    Arr<int32_t> io(env, intOut, false);
    if (ao.buf.empty() || po.buf.size() < 11 || io.buf.size() < 4 || d.buf.size() < 6 || (kind == 1 && pp.buf.size() < 11))
        return GINGR_ERR_BAD_ARGUMENT;
    gingr_state_scalars s{};
    if (kind == 1) {
        const double *p = pp.ptr();
        for (int q = 0; q < 3; ++q) s.euler[q] = p[q], s.center[q] = p[3 + q], s.translation[q] = p[6 + q];
        s.scale = p[9], s.sigma2 = p[10], s.iteration = iteration, s.status = status;
    }
    gingr_mh_request req{};
    req.flavour = flavour, req.kind = kind, req.cpd = &cp, req.icp = &ip, req.z = zz.ptr(), req.alpha = aa.ptr();
    req.scalars = kind == 1 ? &s : nullptr;
    req.eval_sdev = evalSdev, req.eval_points = evalPoints, req.need_forward = needForward ? 1 : 0;
    gingr_mh_result res{};
    const int rc = gingr_fitter_mh_step(P<gingr_fitter>(f), &req, ao.ptr(), fo.ptr(), &res);
    if (rc != GINGR_OK) return rc;
    double *p = po.ptr();
    for (int q = 0; q < 3; ++q) p[q] = res.scalars.euler[q], p[3 + q] = res.scalars.center[q], p[6 + q] = res.scalars.translation[q];
    p[9] = res.scalars.scale, p[10] = res.scalars.sigma2;
    io.ptr()[0] = res.scalars.iteration, io.ptr()[1] = res.scalars.status, io.ptr()[2] = res.forward_status, io.ptr()[3] = res.backward_status;
    d.ptr()[0] = res.log_value, d.ptr()[1] = res.dist_sum, d.ptr()[2] = res.dist_max, d.ptr()[3] = (double)res.count;
    d.ptr()[4] = res.log_q_forward, d.ptr()[5] = res.log_q_backward;
    return rc;
}
JFN(jint, fitterMhRestore)(JNIEnv *, jclass, jlong f) { return gingr_fitter_mh_restore(P<gingr_fitter>(f)); }
#else
// No JDK headers on this machine: the shim is not built (the C ABI it wraps is still covered by the Python tests).
#endif

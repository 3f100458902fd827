package gingr.hip;

/**
 * JNI binding of libgingr_hip.so (C ABI: include/gingr_hip.h) through the thin shim jvm/native/gingr_jni.cpp.
 * One static native method per C entry point the Scala plugin needs; only primitive arrays and opaque handles (long)
 * cross the boundary -- no JVM objects, no callbacks into the JVM.  Every int return is a gingr_status (0 = ok).
 *
 * NOT COMPILED IN THIS REPOSITORY'S IMAGE (no JDK there); see INTEGRATION.md for the build recipe.
 */
public final class GingrHipNative {
    static {
        System.loadLibrary("gingr_jni"); // links libgingr_hip.so
    }

    private GingrHipNative() {}

    public static native int deviceCount();
    public static native long ctxCreate(int device);                       // 0 on failure
    public static native void ctxDestroy(long ctx);
    public static native String lastError(long ctx);

    // stateless all-pairs operators (gingr_cpd_stats / gingr_nn / gingr_gauss_block / gingr_cpd_initial_sigma2)
    public static native int cpdStats(long ctx, double[] fitXyz, double[] targetXyz, double sigma2, double w,
                                      double[] den, double[] p1, double[] px, double[] pt1, double[] scalars6);
    public static native int cpdInitialSigma2(long ctx, double[] refXyz, double[] targetXyz, double[] out1);
    public static native int nn(long ctx, double[] queryXyz, double[] targetXyz, int[] idx, double[] d2, double[] meanDistance1);
    public static native int gaussBlock(long ctx, double[] a, double[] b, double sigma, double scaling, double[] out);

    // model resident on the device (gingr_model_upload): basis is Breeze's column-major basisMatrix.data
    public static native long modelUpload(long ctx, long mTotal, int rank, double[] refXyz, double[] meanXyz,
                                          double[] basisColMajor, double[] variance, long rowBegin, long rowEnd);
    public static native void modelDestroy(long model);

    // device-resident registration state (gingr_fitter_*)
    public static native long fitterCreate(long ctx, long model);
    public static native void fitterDestroy(long fitter);
    public static native int fitterSetTarget(long fitter, double[] targetXyz);
    public static native int fitterSetLandmarks(long fitter, int[] pid, double[] xyz, double[] cov9);
    public static native int fitterSetOptions(long fitter, int globalTransform, double stepLength);
    /** The run's stopping rule on the device (GingrAlgorithm.run's dropWhile with CPD's |sigma2 - last| < threshold): an update that moves
     *  sigma2 by less marks the state, later updates leave it as it is, so a whole run can be ONE fitterUpdateCpd call.  threshold < 0: off;
     *  the call also clears the mark.  fitterStopRuleHit: the mark as of the last fitterGetState (negative: an error code). */
    public static native int fitterSetStopThreshold(long fitter, double threshold);
    public static native int fitterStopRuleHit(long fitter);
    /** poseScalars = { phi, theta, psi, cx, cy, cz, tx, ty, tz, scale, sigma2 } */
    public static native int fitterSetState(long fitter, double[] alpha, double[] poseScalars11, int iteration, int status);
    public static native int fitterUpdateCpd(long fitter, double w, double lambda, int nIterations);
    public static native int fitterUpdateIcp(long fitter, double initialSigma, double endSigma, int maxIterations, int nIterations);
    /** probabilistic = true: ONE update whose proposal is posterior.sample(); z = rank standard-normal draws of the JVM's Random */
    public static native int fitterUpdateCpdSample(long fitter, double w, double lambda, double[] z);
    public static native int fitterUpdateIcpSample(long fitter, double initialSigma, double endSigma, int maxIterations, double[] z);
    /** posterior(current state).gp.logpdf(posterior.coefficients(mesh)); out1[0] receives the value */
    public static native int fitterPosteriorLogpdfCpd(long fitter, double w, double lambda, double[] meshXyz, double[] out1);
    public static native int fitterPosteriorLogpdfIcp(long fitter, double initialSigma, double endSigma, int maxIterations,
                                                      double[] meshXyz, double[] out1);
    /** iterStatus2 = { iteration, status }; fitXyz may be null */
    public static native int fitterGetState(long fitter, double[] alpha, double[] poseScalars11, int[] iterStatus2, double[] fitXyz);

    // ---- ICP with the surface correspondence (TriangularClosestPoint): flat triangle id triples of both meshes
    public static native int fitterSetMeshes(long fitter, int[] modelTriangles, int[] targetTriangles);
    /** 0 = TriangularClosestPoint, 1 = AlongNormalClosestPoint */
    public static native int fitterSetSurfaceMethod(long fitter, int method);
    /** reverseCorrespondenceDirection (all ICP flavours); call after fitterSetTarget / fitterSetMeshes */
    public static native int fitterSetCorrespondenceDirection(long fitter, int reversed);
    public static native int fitterGetReversedCorrespondence(long fitter, int[] modelVertexId, double[] w);
    public static native int fitterUpdateIcpSurface(long fitter, double initialSigma, double endSigma, int maxIterations, int nIterations);
    public static native int fitterUpdateIcpSurfaceSample(long fitter, double initialSigma, double endSigma, int maxIterations, double[] z);
    public static native int fitterPosteriorLogpdfIcpSurface(long fitter, double initialSigma, double endSigma, int maxIterations,
                                                             double[] meshXyz, double[] out1);
    /** cpXyz [3 M], w [M] in {0, 1} of the last surface correspondence */
    public static native int fitterGetSurfaceCorrespondence(long fitter, double[] cpXyz, double[] w);
    // gingr_fitter_surface_distance_stats / gingr_mesh_distance_stats: out4 = {sum d, max d, count, sum log N(d; 0, sdev)}
    public static native int fitterSurfaceDistanceStats(long fitter, int direction, long nPoints, double[] pointsXyzOrNull,
                                                        int boundaryAware, double sdev, double[] out4);
    // gingr_classic_cpd_*: kind 0 rigid, 1 affine, 2 non-rigid; the handle owns (TY, sigma2) on the device
    public static native long classicCpdCreate(long ctx, int kind, double[] templateXyz, double[] targetXyz, double lambda, double beta,
                                               double w);                                                            // 0 on failure
    public static native void classicCpdDestroy(long h);
    public static native int classicCpdIterate(long h, int nIterations);
    public static native int classicCpdGet(long h, double[] tyXyzOrNull, double[] sigma2OrNull1, double[] transform13OrNull,
                                           double[] wXyzOrNull);
    public static native int classicCpdSet(long h, double[] tyXyzOrNull, double sigma2);
    public static native int meshDistanceStats(long ctx, double[] pointsXyz, double[] verticesXyz, int[] triangles, int boundaryAware,
                                               double sdev, double[] out4);

    // ---- GPMM construction in HBM (GPMMTriangleMesh3D.Gaussian / GaussianMixture / AutomaticGaussian, automaticGPMMfromTemplate)
    /** returns the gingr_model handle (0 on failure; see lastError); maxRank <= 0 = model limit; rowEnd <= 0 = all rows */
    public static native long gpmmBuildGaussian(long ctx, long mTotal, double[] refXyz, double[] sigmas, double[] scalings,
                                                double relativeTolerance, int maxRank, long rowBegin, long rowEnd);
    /** gingr_gpmm_build_diagonal: one scalar kernel per coordinate, each kernel k = { kind (0 Gaussian mixture, 1 dot, 2 lookup),
     *  sigmas, scalings, mirror, scaling, lookup (M x M row-major or null) } spread over parallel arrays of length 3 (x, y, z);
     *  coordinates whose entries are equal share one factorisation.  GaussianSymmetry = kind 0 with mirror {-1, +1, +1}. */
    public static native long gpmmBuildDiagonal(long ctx, long mTotal, double[] refXyz, int[] kind3, double[][] sigmas3,
                                                double[][] scalings3, double[] mirror3, double[] scaling3, double[][] lookup3,
                                                double relativeTolerance, int maxRank, long rowBegin, long rowEnd);
    /** closestPointOnSurface for every point: any output may be null (cpXyz[3n], d2[n], triId[n], bary[3n]) */
    public static native int meshClosestPoints(long ctx, double[] pointsXyz, double[] verticesXyz, int[] triangles, double[] cpXyz,
                                               double[] d2, int[] triId, double[] bary);
    /** model.newReference(newReference, interpolator): new point i = sum_k weights[3i+k] * source point vertexIds[3i+k];
     *  returns the model handle or 0 */
    public static native long modelNewReference(long ctx, long sourceModel, double[] newRefXyz, int[] vertexIds, double[] weights,
                                                long rowBegin, long rowEnd);
    // ---- classic rigid / similarity ICP (other/algorithms/icp/RigidICP.scala; kind 0 = RigidRegistrator3D, 1 = AffineRegistrator3D)
    public static native long rigidIcpCreate(long ctx, int kind, double[] templateXyz, double[] targetXyz);
    public static native void rigidIcpDestroy(long handle);
    /** distances[k] = mean closest-point distance iteration k measured before moving the template (may be null) */
    public static native int rigidIcpIterate(long handle, int nIterations, double[] distances);
    public static native int rigidIcpGet(long handle, double[] pointsXyz, double[] transform13);
    public static native int rigidIcpSet(long handle, double[] pointsXyz);

    // ---- optimal-step non-rigid ICP (gingr/other/algorithms/icp/NonRigidOptimalStepICP.scala): the correspondence of an explicit
    // template through the surface-ICP query (fitterSetFitPoints + fitterIcpSurfacePhase(…, 0) + fitterGetSurfaceCorrespondence),
    // and the least-squares step; kind 0 = N-ICP-T, 1 = N-ICP-A; edges = unique (p1 < p2) vertex pairs; outLmXyz may be null
    public static native int fitterSetFitPoints(long fitter, double[] fitXyz);
    /** one phase of the surface-ICP update (gingr_fitter_icp_surface_phase_async); phase 0 = the correspondence query alone */
    public static native int fitterIcpSurfacePhase(long fitter, double initialSigma, double endSigma, int maxIterations, int phase);
    public static native int nicpSolve(long ctx, int kind, double[] templateXyz, int[] edges, double[] w, double[] cpXyz, int[] lmIds,
                                       double[] lmTargetXyz, double alpha, double beta, double gamma, double[] outXyz, double[] outLmXyz);
    /** out2 = { maximumPointDistance, minimumPointDistance } (PointSetHelper) */
    public static native int pointsetDistanceExtrema(long ctx, double[] xyz, double[] out2);
    /** any array may be null; basisColMajor is 3 M_local x rank, unit columns */
    public static native int modelDownload(long ctx, long model, double[] ref, double[] mean, double[] basisColMajor, double[] variance);
    public static native int modelRank(long model);

    // ---- retry counter of the probabilistic proposal (GingrAlgorithm.scala:69-70): lives next to the device state
    /** setTo >= 0 writes, setTo < 0 only reads; out1[0] receives the value */
    public static native int fitterRetryCounter(long fitter, int setTo, int[] out1);

    // ---- device group: row shards on several GPUs of one node from this ONE process (gingr_group_*; no torch, no MPI)
    public static native long groupCreate(int[] devices);                  // 0 on failure
    public static native void groupDestroy(long group);
    public static native String groupLastError(long group);
    public static native int groupModelUpload(long group, long mTotal, int rank, double[] refXyz, double[] meanXyz,
                                              double[] basisColMajor, double[] variance);
    public static native int groupGpmmBuildGaussian(long group, long mTotal, double[] refXyz, double[] sigmas, double[] scalings,
                                                    double relativeTolerance, int maxRank);
    public static native int groupSetTarget(long group, double[] targetXyz);
    public static native int groupSetLandmarks(long group, int[] pid, double[] xyz, double[] cov9);
    public static native int groupSetOptions(long group, int globalTransform, double stepLength);
    public static native int groupSetState(long group, double[] alpha, double[] poseScalars11, int iteration, int status);
    /** fitXyz [3 M_total] gathered from all shards (may be null) */
    public static native int groupGetState(long group, double[] alpha, double[] poseScalars11, int[] iterStatus2, double[] fitXyz);
    public static native int groupUpdateCpd(long group, double w, double lambda, int nIterations);
    public static native int groupUpdateIcp(long group, double initialSigma, double endSigma, int maxIterations, int nIterations);
    public static native int groupSynchronize(long group);
    /** out2 = {physical devices behind the shards, 1 if the peer-read send buffers are fine-grained device memory} */
    public static native int groupExchangeInfo(long group, int[] out2);

    // ---- native RCCL exchange: one JVM process per GPU, the library enqueues ncclAllReduce on the context's stream itself
    // (gingr_ctx_rccl_*, include/gingr_hip.h).  id32 = the 128 bytes of the ncclUniqueId as int[32]: rank 0 creates it, the host
    // application hands it to the other ranks (socket, file, MPI ...), every rank calls ctxRcclInit (collective).
    public static native int rcclUniqueId(long ctx, int[] id32);
    public static native int ctxRcclInit(long ctx, int[] id32, int world, int rank);
    /** one-off: sum the basis moments of the row shards (between modelUpload of a shard and modelFinalize) */
    public static native int ctxRcclAllreduceModelMoments(long ctx, long model);
    public static native int fitterUpdateCpdRccl(long fitter, double w, double lambda, int nIterations);
    public static native int fitterUpdateIcpRccl(long fitter, double initialSigma, double endSigma, int maxIterations, int nIterations);
    /** gingr_ctx_option: 0 cull, 1 fine cull, 2 closest-point grid, 3 triangle grid -- code paths with identical results (tests, timing comparisons);
     *  4 split column-sum exchange of the native RCCL path (round 5, default off: the first half's all-reduce behind the second half of pass 1);
     *  5 surface-ICP Gram matrix as the model's moment minus the rejected rows (-1 default: from 16 384 rows on, 0 never, 1 always) */
    public static native int ctxSetOption(long ctx, int option, int value);

    // ---- round 4: every flavour of the update on row shards (0 CPD, 1 ICP point cloud, 2 ICP surface), the sampled proposal
    // (z = r standard normals, the SAME on every shard; null = mean update) and the log transition density
    // (GeneratorWrapperStochastic.scala:42-63).  Triangles index the vertices of the FULL template / target.
    public static native int groupSetMeshes(long group, int[] modelTriangles, int[] targetTriangles);
    public static native int groupSetSurfaceMethod(long group, int method);
    /** IcpConfiguration.reverseCorrespondenceDirection for the group's ICP flavours (set the meshes first when there is more than one shard) */
    public static native int groupSetCorrespondenceDirection(long group, boolean reversed);
    public static native int groupUpdate(long group, int flavour, double w, double lambda, double initialSigma, double endSigma,
                                         int maxIterations, int nIterations, double[] z);
    public static native int groupPosteriorLogpdf(long group, int flavour, double w, double lambda, double initialSigma, double endSigma,
                                                  int maxIterations, double[] meshXyzFull, double[] out1);
    public static native int fitterUpdateRccl(long fitter, int flavour, double w, double lambda, double initialSigma, double endSigma,
                                              int maxIterations, int nIterations, double[] z);
    public static native int fitterPosteriorLogpdfRccl(long fitter, int flavour, double w, double lambda, double initialSigma,
                                                       double endSigma, int maxIterations, double[] meshXyzFull, double[] out1);

    // ---- one Metropolis-Hastings step per call (gingr_fitter_mh_step, include/gingr_hip.h): proposal (kind 0: update(x, probabilistic =
    // true) with the draws z; kind 1: the parameters of a random-walk proposal), model-to-target likelihood of the proposal and the
    // transition densities of the informed proposal, one synchronisation.  intOut = {iteration, status, forward status, backward status},
    // dblOut = {log value, distance sum, distance max, count, log q(x'|x) (NaN unless needForward), log q(x|x')}.
    // fitterMhRestore: the proposal was rejected, the state the step started from is the device state again.
    public static native int fitterMhStep(long fitter, int flavour, int kind, double w, double lambda, double initialSigma, double endSigma,
                                          int maxIterations, double[] z, double[] alpha, double[] poseScalars11, int iteration, int status,
                                          double evalSdev, long evalPoints, boolean needForward, double[] alphaOut, double[] fitOut,
                                          double[] poseOut11, int[] intOut4, double[] dblOut6);
    public static native int fitterMhRestore(long fitter);
}

/*
 * gingr_hip.h -- C ABI of libgingr_hip.so, the MI355X (gfx950) implementation of GiNGR's
 * per-iteration update (reference: unibas-gravis/GiNGR, gingr/api/GingrAlgorithm.update).
 *
 * This header is the drop-in boundary: plain C, pointers and sizes only, no C++/torch/JVM types.
 * A JVM host binds it through the JNI shim in jvm/ (see INTEGRATION.md); the Python host layer
 * gingr_amd/ binds it with ctypes.  G/ = src/main/scala/gingr/ of the reference.
 *
 * Conventions
 *  - every function returns a gingr_status (0 = ok); gingr_last_error(ctx) has the text.
 *  - host point arrays are interleaved x,y,z float64 ("x1x,x1y,x1z,x2x,..."), the layout of
 *    G/api/registration/utils/PointSequenceConverter.scala:54-59.
 *  - the caller owns every host buffer; the library never keeps a host pointer after the call returns.
 *  - a gingr_ctx binds ONE device and ONE stream and is not thread-safe; distinct contexts are independent
 *    (one per MH chain / per GPU), mirroring the reference's "one algorithm instance per chain"
 *    (G/api/GingrAlgorithm.scala:69-70 retryCounter is an unsynchronised var).
 *  - functions whose name ends in _async only enqueue work on the context's stream.
 */
#ifndef GINGR_HIP_H
#define GINGR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gingr_ctx gingr_ctx;
typedef struct gingr_model gingr_model;
typedef struct gingr_fitter gingr_fitter;

typedef enum gingr_status {
    GINGR_OK = 0,
    GINGR_ERR_BAD_ARGUMENT = 1,
    GINGR_ERR_HIP = 2,          /* a HIP runtime call failed */
    GINGR_ERR_NONFINITE = 3,    /* a result is NaN/Inf (reference: exception inside Try => ModelFlexibilityError) */
    GINGR_ERR_NOT_SPD = 4,      /* Cholesky pivot <= 0 in the posterior solve (same mapping) */
    GINGR_ERR_NO_DEVICE = 5,
    GINGR_ERR_STATE = 6         /* call order violated (e.g. update before target upload) */
} gingr_status;

/* G/api/GlobalTranformationType.scala:20-24 */
typedef enum gingr_global_transform {
    GINGR_NO_TRANSFORMS = 0,
    GINGR_RIGID_TRANSFORMS = 1,
    GINGR_SIMILARITY_TRANSFORMS = 2
} gingr_global_transform;

/* G/api/FittingStatuses.scala:22 */
typedef enum gingr_fitting_status {
    GINGR_FIT_NONE = 0,
    GINGR_FIT_CONVERGED = 1,
    GINGR_FIT_MAX_ITERATION = 2,
    GINGR_FIT_MODEL_FLEXIBILITY_ERROR = 3
} gingr_fitting_status;

/* ------------------------------------------------------------------ context */

int gingr_device_count(void);
int gingr_ctx_create(int device, gingr_ctx **out);
void gingr_ctx_destroy(gingr_ctx *ctx);
const char *gingr_last_error(const gingr_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own. NULL restores it. */
int gingr_ctx_set_stream(gingr_ctx *ctx, void *hip_stream);
void *gingr_ctx_get_stream(gingr_ctx *ctx);
int gingr_ctx_synchronize(gingr_ctx *ctx);
const char *gingr_build_info(void);
/* Run-time options of a context.  All of them select between code paths with IDENTICAL results (the tests compare them bit for bit);
 * they exist for those tests and for same-box timing comparisons, not for tuning -- nothing in the library reads the environment.
 *   GINGR_OPT_CULL       1 (default) / 0: exact-zero tile culling of the CPD passes and exact pruning of the closest-point scans
 *   GINGR_OPT_FINE_CULL  -1 (default: by the regime the device reports) / 0 / 1: quarter-tile culling variant of the CPD passes
 *   GINGR_OPT_NN_GRID    1 (default) / 0: the ICP closest point searches the target's uniform grid first / tile scan only;
 *                        2: as 1, and the stateless gingr_nn searches a grid too (built per call; from a cold start its
 *                        default forms -- all pairs up to 2^26 of them, the box-pruned scan above -- are faster)
 *   GINGR_OPT_TRI_GRID   0 / 1 (default) / 2: the surface ICP's closest surface point searches a grid of the (fixed) target triangles
 *                        first, warm-started from the previous iteration, and the tile scan only answers what the grid cannot certify:
 *                        never / from 16 384 target triangles on / always.  2 also bins the MOVING template's triangles on the device
 *                        in front of every self-intersection test (round 5; same decisions, measured slower than the tile scan at 41k x 82k)
 *   GINGR_OPT_SPLIT_EXCHANGE  0 (default) / 1: row-sharded CPD through the library's own RCCL exchange (gingr_fitter_update_*_rccl_async):
 *                        the column-sum pass runs in two halves of the target tiles and the all-reduce of the first half is enqueued on a
 *                        second stream of the context (event-ordered, same communicator) while the second half computes, so that only
 *                        the second half's all-reduce is exposed.  Costs two short launches instead of one (the emulated per-rank time
 *                        rises by a few microseconds); whether it pays depends on the all-reduce latency of the node.
 *                        Same sums up to the order of the chunk partials (<= 1e-12 on the state).
 *   GINGR_OPT_GRAM_DOWNDATE  -1 (default: from 16 384 local rows on) / 1 / 0: ICP with a surface correspondence gives every accepted pair
 *                        the weight 1 / sigma2 and every rejected one 0 (ICP.scala:50,90-92), so its weighted Gram matrix is the model's
 *                        one-off moment Q^T Q minus the rows of the rejected vertices, scaled: one pass over the rejected rows (0.2 % of
 *                        them at 41k) instead of one over the basis / 0: the pass over the basis.  Same matrix up to the rounding of the
 *                        subtraction (<= 1e-12).  -1 also decides again on the DEVICE in every iteration: with more than one zero-weight
 *                        vertex in eight (open targets, partial overlap) the pass over the basis runs, exactly as with 0 (round 6; a
 *                        fixed rule -- the choice never depends on timing); 1 forces the downdate whatever the rejected fraction is.
 *                        Results are therefore NOT bit-stable across the 16 384-row threshold, across the one-in-eight rule or across
 *                        this option: the two forms round differently, and the exact `intersection point != vertex` comparison of the
 *                        self-intersection test (ClosestPointRegistrator.scala:67) can turn a last-bit difference into a different
 *                        accept / reject decision in the next iteration.
 * No reference counterpart (the reference has one code path per operation). */
typedef enum gingr_ctx_option {
    GINGR_OPT_CULL = 0,
    GINGR_OPT_FINE_CULL = 1,
    GINGR_OPT_NN_GRID = 2,
    GINGR_OPT_TRI_GRID = 3,
    GINGR_OPT_SPLIT_EXCHANGE = 4,
    GINGR_OPT_GRAM_DOWNDATE = 5
} gingr_ctx_option;
int gingr_ctx_set_option(gingr_ctx *ctx, int32_t option, int32_t value);
int gingr_ctx_get_option(gingr_ctx *ctx, int32_t option, int32_t *value);

/* --------------------------------------------- stateless all-pairs operators
 * Each call uploads its inputs, runs on the device, downloads and synchronises.
 */

/*
 * CPD soft-assignment statistics of ONE affinity evaluation, P never materialised.
 * Replaces CpdRegistrationState.P (G/api/registration/config/CPD.scala:54-75), the row/column sums of
 * CPDCorrespondence.estimate (:36) and updateSigma2 (:133-147).
 *   fit[3M], target[3N], sigma2, w  ->  den[N] (= colsum K + c), P1[M], PX[3M], Pt1[N],
 *   scalars[6] = { Np, xPx, trPXY, yPy, sigma2_next, c }.   Any output pointer may be NULL.
 */
int gingr_cpd_stats(gingr_ctx *ctx, int64_t M, const double *fit, int64_t N, const double *target, double sigma2,
                    double w, double *den, double *P1, double *PX, double *Pt1, double *scalars);

/* sigma2_0 = sum_ij ||x_j - y_i||^2 / (3 M N)      (CPD.scala:81-90) */
int gingr_cpd_initial_sigma2(gingr_ctx *ctx, int64_t M, const double *ref, int64_t N, const double *target,
                             double *sigma2_out);

/*
 * Exact brute-force nearest neighbour, lowest index wins ties.
 * Replaces ClosestPointUnstructuredPointsDomain3D.closestPointCorrespondence
 * (G/api/registration/utils/ClosestPointRegistrator.scala:133-148) / scalismo findClosestPoint.
 *   query[3M], target[3N] -> idx[M] (int32), d2[M] (squared distance), *mean_distance (mean of sqrt(d2)).
 */
int gingr_nn(gingr_ctx *ctx, int64_t M, const double *query, int64_t N, const double *target, int32_t *idx,
             double *d2, double *mean_distance);

/*
 * Gaussian-kernel covariance block out[i*nb + j] = scaling * exp(-||a_i - b_j||^2 / sigma^2)
 * (G/api/gpmm/GPMMHelper.scala:99-102: GaussianKernel(sigma) * scaling; the (x) I3 of DiagonalKernel is implicit;
 *  with sigma = sqrt(2) beta, scaling = 1 it is CPDFactory.initializeKernelMatrixG,
 *  G/other/algorithms/cpd/CPDFactory.scala:54-66).
 */
int gingr_gauss_block(gingr_ctx *ctx, int64_t na, const double *A, int64_t nb, const double *B, double sigma,
                      double scaling, double *out);

/* ------------------------------------------------------------------- model
 * A point distribution model resident on the device: scalismo PointDistributionModel (reference, mean, basisMatrix,
 * variance) as held by GeneralRegistrationState.model (G/api/GeneralRegistrationState.scala:29).
 *   ref[3M], mean[3M] (mean DISPLACEMENT), basis[3M*r] COLUMN-major (3M rows) exactly as Breeze stores
 *   basisMatrix, variance[r].
 * Row sharding (multi-GPU): the model holds points [row_begin, row_end) of M_total; pass 0, M for one GPU.
 * Host arrays are always the FULL model; only the shard is uploaded.
 */
int gingr_model_upload(gingr_ctx *ctx, int64_t M_total, int32_t rank, const double *ref, const double *mean,
                       const double *basis_colmajor, const double *variance, int64_t row_begin, int64_t row_end,
                       gingr_model **out);
void gingr_model_destroy(gingr_model *model);
/* Row-sharded models only: after upload every shard holds its partial one-off moments of the basis (Q^T Q and the
 * per-coordinate cross moments; float64, count elements at dev_ptr); the host all-reduces (sum) them across shards once,
 * then calls gingr_model_finalize, which factors Q^T Q / 1e-5 + I for the coefficient projections.
 * Single-shard uploads are finalized by gingr_model_upload. */
int gingr_model_gram_exchange(gingr_model *model, void **dev_ptr, int64_t *count);
int gingr_model_finalize(gingr_ctx *ctx, gingr_model *model);
/* ---- GPMM construction on the device ----------------------------------------------------------
 * Replaces GPMMTriangleMesh3D(reference, relativeTolerance).Gaussian / .GaussianMixture / .AutomaticGaussian
 * (G/api/gpmm/GPMMHelper.scala:96-130) and automaticGPMMfromTemplate (G/api/registration/utils/GPMMHelper.scala:39-69):
 * zero-mean GP with DiagonalKernel(sum_i scaling_i * GaussianKernel(sigma_i), 3), low-rank approximation by scalismo's
 * pivoted Cholesky (stopped at relative_tolerance * trace, or at max_rank columns; max_rank <= 0 means the model limit
 * 512) and the eigen-decomposition of its factor (LowRankGaussianProcess.approximateGPCholesky, GPMM.construct :39-55).
 * The basis is produced in HBM; the result is a finished (single shard) or to-be-finalized (row shard) gingr_model exactly
 * as from gingr_model_upload.  ref = interleaved xyz of the FULL reference; row_end <= 0 means M_total. */
int gingr_gpmm_build_gaussian(gingr_ctx *ctx, int64_t M_total, const double *ref, int32_t n_kernels, const double *sigmas,
                              const double *scalings, double relative_tolerance, int32_t max_rank, int64_t row_begin,
                              int64_t row_end, gingr_model **out);
/* The other kernels of GPMMTriangleMesh3D (G/api/gpmm/GPMMHelper.scala:96-142) and gingr.simple.SimpleTriangleModels3D.create
 * (G/simple/SimpleModels.scala:52-75): a DiagonalKernel with one scalar kernel per coordinate,
 *   GINGR_KERNEL_GAUSSIAN_MIXTURE  sum_i scalings[i] exp(-|x-y|^2 / sigmas[i]^2), plus mirror * the same kernel evaluated at
 *                                  (diag(-1,1,1) x, y) when mirror = +-1 -- KernelHelper.symmetrizeKernel (KernelHelper.scala:25-38):
 *                                  GaussianSymmetry = (mirror -1 for x, +1 for y and z)
 *   GINGR_KERNEL_DOT               scaling * (x . y) -- DotProductKernel.k returns x.dot(y) whatever kernel / gamma it wraps
 *                                  (KernelHelper.scala:40-51): GaussianDot(sigma, scaling), InverseLaplacianDot(scaling, gamma)
 *   GINGR_KERNEL_LOOKUP            scaling * lookup[i * M_total + j] -- LookupKernel(reference, m) on the reference points
 *                                  (KernelHelper.scala:76-84), m = pinv(graph Laplacian) for InverseLaplacian (host array)
 * Three equal kernels run one scalar pivoted Cholesky (the coordinates swap in triplets); different kernels run scalismo's
 * generic factorisation over the 3M (point, coordinate) entries.  Exact ties follow scalismo's rule (first maximum in the
 * permuted index order).  kx / ky / kz may be the same pointer.
 * gingr_gpmm_build_gaussian(..) == gingr_gpmm_build_diagonal with the same mixture (mirror 0) three times. */
#define GINGR_KERNEL_GAUSSIAN_MIXTURE 0
#define GINGR_KERNEL_DOT 1
#define GINGR_KERNEL_LOOKUP 2
typedef struct gingr_scalar_kernel {
    int32_t kind;
    int32_t n_kernels;       /* mixture */
    const double *sigmas;    /* mixture, host */
    const double *scalings;  /* mixture, host */
    double mirror;           /* mixture: 0, +1 or -1 */
    double scaling;          /* dot / lookup */
    const double *lookup;    /* lookup: host, M_total x M_total row-major */
} gingr_scalar_kernel;
int gingr_gpmm_build_diagonal(gingr_ctx *ctx, int64_t M_total, const double *ref, const gingr_scalar_kernel *kx,
                              const gingr_scalar_kernel *ky, const gingr_scalar_kernel *kz, double relative_tolerance,
                              int32_t max_rank, int64_t row_begin, int64_t row_end, gingr_model **out);
/* PointSetHelper.maximumPointDistance / minimumPointDistance (GPMMHelper.scala:75-87; the O(n^2) scans behind
 * AutomaticGaussian and automaticGPMMfromTemplate): largest pairwise distance, smallest distance to the nearest OTHER point. */
int gingr_pointset_distance_extrema(gingr_ctx *ctx, const double *xyz, int64_t n, double *max_distance, double *min_distance);
/* Copy the local rows of a model back to the host in gingr_model_upload's layout (any pointer may be NULL):
 * ref / mean [3 M_local], basis column-major [3 M_local x rank] (unit columns: Q0 / sqrt(variance)), variance [rank]. */
int gingr_model_download(gingr_ctx *ctx, const gingr_model *model, double *ref, double *mean, double *basis_colmajor,
                         double *variance);
int64_t gingr_model_num_points(const gingr_model *model); /* local rows */
int32_t gingr_model_rank(const gingr_model *model);

/* PointDistributionModel.instance(alpha) posed and scaled:  s * (R (ref + mean + U sqrt(lam) alpha - c) + c + t)
 * (G/api/ModelFittingParameters.scala:130-143).  out_xyz[3*M_local]. */
int gingr_model_instance(gingr_ctx *ctx, const gingr_model *model, const double *alpha, const double euler[3],
                         const double center[3], const double translation[3], double scale, double *out_xyz);

/* PointDistributionModel.transform(rigid).coefficients(mesh): GP regression at all points with noise 1e-5 I3
 * (G/api/GingrAlgorithm.scala:212-216,234-237).  mesh_xyz[3*M_local] -> alpha[r].  Single shard only. */
int gingr_model_coefficients(gingr_ctx *ctx, const gingr_model *model, const double euler[3], const double center[3],
                             const double translation[3], const double *mesh_xyz, double *alpha);

/* PointDistributionModel.transform(rigid).posterior(obs).mean  (G/api/GingrAlgorithm.scala:297-301).
 * Observations: every local point i with weight[i] > 0 is observed at obs_xyz[3i..] with covariance I3/weight[i];
 * n_lm landmark observations (point id, target point, full 3x3 covariance, row-major) are appended and their point ids
 * must carry weight 0.  Outputs: mean_xyz[3*M_local] (posterior mean mesh), coeffs[r] (regression coefficients). */
int gingr_model_posterior_mean(gingr_ctx *ctx, const gingr_model *model, const double euler[3], const double center[3],
                               const double translation[3], const double *obs_xyz, const double *weight,
                               int32_t n_lm, const int32_t *lm_pid, const double *lm_xyz, const double *lm_cov,
                               double *mean_xyz, double *coeffs);

/* ------------------------------------------------------------------ fitter
 * Device-resident registration state = the numeric content of GeneralRegistrationState
 * (alpha, Euler angles, centre, translation, scale, sigma2, fit, iteration, status) plus target and
 * configuration; one update = GingrAlgorithm.update followed by GingrGeneratorWrapper.propose's fit refresh
 * (G/api/GingrAlgorithm.scala:192-254; G/api/sampling/generators/GingrGeneratorWrapper.scala:28-39).
 */
typedef struct gingr_state_scalars {
    double euler[3];        /* phi, theta, psi  (EulerAngles, G/api/ModelFittingParameters.scala:35-37) */
    double center[3];       /* rotation centre */
    double translation[3];
    double scale;
    double sigma2;
    int32_t iteration;
    int32_t status;         /* gingr_fitting_status */
} gingr_state_scalars;

typedef struct gingr_cpd_params {   /* CpdConfiguration, CPD.scala:105-115 */
    double w;
    double lambda;
} gingr_cpd_params;

typedef struct gingr_icp_params {   /* IcpConfiguration, ICP.scala:54-66 (PointcloudClosestPoint flavour) */
    double initial_sigma;
    double end_sigma;
    int32_t max_iterations;
} gingr_icp_params;

int gingr_fitter_create(gingr_ctx *ctx, const gingr_model *model, gingr_fitter **out);
void gingr_fitter_destroy(gingr_fitter *f);
/* target cloud, replicated on every shard */
int gingr_fitter_set_target(gingr_fitter *f, int64_t N, const double *target_xyz);
/* landmark observations (GeneralRegistrationState.landmarkCorrespondences, GeneralRegistrationState.scala:43-62);
 * pids are GLOBAL point ids; cov row-major 3x3 per landmark.  n_lm = 0 clears. */
int gingr_fitter_set_landmarks(gingr_fitter *f, int32_t n_lm, const int32_t *lm_pid, const double *lm_xyz,
                               const double *lm_cov);
int gingr_fitter_set_options(gingr_fitter *f, int32_t global_transform, double step_length);
/* The stopping rule of a run, applied on the device (GingrAlgorithm.run's dropWhile, G/api/GingrAlgorithm.scala:142-153, with the
 * CPD rule |sigma2 - last sigma2| < threshold, G/api/registration/config/CPD.scala:108-110): with threshold >= 0 an update that
 * moves sigma2 by less than threshold marks the device state as stopped, and every later update enqueued on it leaves it as it is
 * -- exactly like a failed fit -- so a host can enqueue all updates of a run in ONE call and read the state it stopped at.
 * threshold < 0 (the default): no rule.  The call also clears the mark; so does gingr_fitter_set_state.
 * gingr_fitter_stop_rule_hit: the mark as of the last gingr_fitter_get_state (no transfer of its own). */
int gingr_fitter_set_stop_threshold(gingr_fitter *f, double threshold);
int gingr_fitter_stop_rule_hit(gingr_fitter *f, int32_t *hit);
/* state in: alpha[r] + scalars; the fit is recomputed on the device (modelInstanceShapePoseScale). */
int gingr_fitter_set_state(gingr_fitter *f, const double *alpha, const gingr_state_scalars *s);
/* state out (synchronises): alpha[r], scalars, fit_xyz[3*M_local]; any pointer may be NULL. */
int gingr_fitter_get_state(gingr_fitter *f, double *alpha, gingr_state_scalars *s, double *fit_xyz);
/* diagnostics of the LAST update (synchronises): P1[M_local], PX[3*M_local] (CPD) or nn idx (ICP), coeffs. */
int gingr_fitter_get_cpd_stats(gingr_fitter *f, double *P1, double *PX, double *den, double *scalars6);
int gingr_fitter_get_icp_idx(gingr_fitter *f, int32_t *idx, double *d2);

/* n_iterations updates back to back on the stream, no host synchronisation in between (single shard). */
int gingr_fitter_update_cpd_async(gingr_fitter *f, const gingr_cpd_params *p, int32_t n_iterations);
int gingr_fitter_update_icp_async(gingr_fitter *f, const gingr_icp_params *p, int32_t n_iterations);

/* ---- ICP with the surface correspondence -----------------------------------------------------------
 * The reference's DEFAULT ICP method: ICPCorrespondence.estimate with TriangularClosestPoint (ICP.scala:36-52,63) ->
 * ClosestPointTriangleMesh3D.closestPointCorrespondence (ClosestPointRegistrator.scala:75-100): closest point on the target
 * SURFACE, rejected (weight 0) when the nearest target vertex is a boundary vertex, when the vertex normals are opposite, or
 * when the line through the template vertex along the closest-point vector meets the template itself first.
 * gingr_fitter_set_meshes: triangle lists (vertex ids of the model reference resp. of the target set by
 * gingr_fitter_set_target, 3 ids per triangle); call it after gingr_fitter_set_target.  On a row shard the model triangles are those
 * of the WHOLE template (ids of the full model) and the update runs through the sharded entry points further down.
 * The self-intersection test restates scalismo's getIntersectionPoints as a Moeller-Trumbore line/triangle test
 * (semantics unpinned without the scalismo source). */
int gingr_fitter_set_meshes(gingr_fitter *f, int64_t n_model_triangles, const int32_t *model_triangles,
                            int64_t n_target_triangles, const int32_t *target_triangles);
/* 0 = TriangularClosestPoint (default), 1 = AlongNormalClosestPoint (ClosestPointRegistrator.scala:102-131: the target point
 * hit first by the line through the template vertex along its vertex normal; no hit = rejected) */
int gingr_fitter_set_surface_method(gingr_fitter *f, int32_t method);
/* reverseCorrespondenceDirection (ICP.scala:46-48, ClosestPointRegistrator.scala:34-49): the correspondence is computed from the
 * target to the template and inverted; holds for all three ICP flavours until reset (call after gingr_fitter_set_target and, for
 * the surface flavours, gingr_fitter_set_meshes).  gingr_fitter_get_reversed_correspondence: per TARGET vertex the template vertex
 * it was assigned to and its weight in {0, 1} (last phase 0). */
int gingr_fitter_set_correspondence_direction(gingr_fitter *f, int32_t reversed);
int gingr_fitter_get_reversed_correspondence(gingr_fitter *f, int32_t *model_vertex_id, double *w);
int gingr_fitter_update_icp_surface_async(gingr_fitter *f, const gingr_icp_params *params, int32_t n_iterations);
int gingr_fitter_icp_surface_phase_async(gingr_fitter *f, const gingr_icp_params *params, int32_t phase);
/* probabilistic proposal / log transition density with the surface correspondence (see the _sample / _logpdf entry points below) */
int gingr_fitter_update_icp_surface_sample_async(gingr_fitter *f, const gingr_icp_params *params, const double *z);
int gingr_fitter_posterior_logpdf_icp_surface(gingr_fitter *f, const gingr_icp_params *params, const double *mesh_xyz,
                                              double *logpdf);
/* correspondences of the last surface phase 0: closest surface point [3 M] and weight in {0, 1} [M] per model vertex */
int gingr_fitter_get_surface_correspondence(gingr_fitter *f, double *cp_xyz, double *w);

/* ---- surface distance statistics ("the same query inside IndependentPointDistanceEvaluator") ----
 * d_i = |p_i - closestPointOnSurface(p_i)| reduced on the device: out = {sum d, max d, number of points counted,
 * sum of log N(d; 0, sdev)} (the last is 0 when sdev == 0; breeze Gaussian(0, sdev).logPdf).
 * gingr_fitter_surface_distance_stats works on the fitter's CURRENT state (meshes from gingr_fitter_set_meshes):
 *   direction 0: the first n_points vertices of the current fit (0 = all; `points` must be NULL) against the target surface
 *                = IndependentPointDistanceEvaluator.distModelToTarget
 *                  (G/api/sampling/evaluators/IndependentPointDistanceEvaluator.scala:54-60; with numberOfPointsForComparison
 *                  the reference walks the decimated instance's point ids 0..n'-1 over the full sample, :49-50);
 *   direction 1: `points` [3 n_points] (NULL = every target vertex) against the surface of the current fit
 *                = distTargetToModel (:62-67).
 * boundary_aware != 0 skips the points whose surface point lies nearest to a boundary vertex of the mesh
 * (G/api/helper/RegistrationComparison.scala:63-73, avgDistanceBoundaryAware).
 * gingr_mesh_distance_stats is the same reduction for any point list against any triangle mesh (host arrays):
 * MeshMetrics.avgDistance = out[0] / out[2], RegistrationComparison.maxDistance (:24-35) = out[1], Hausdorff = the larger of
 * the two directed maxima.  Both calls synchronise. */
int gingr_fitter_surface_distance_stats(gingr_fitter *f, int32_t direction, int64_t n_points, const double *points,
                                        int32_t boundary_aware, double sdev, double out[4]);
int gingr_mesh_distance_stats(gingr_ctx *ctx, int64_t n_points, const double *points, int64_t n_vertices, const double *vertices,
                              int64_t n_triangles, const int32_t *triangles, int32_t boundary_aware, double sdev, double out[4]);


/* Closest point of a triangle mesh to every query point (scalismo mesh.operations.closestPointOnSurface, the query behind
 * ClosestPointTriangleMesh3D, ClosestPointRegistrator.scala:75-100, and TriangleMeshInterpolator3D): cp_xyz[3n], squared
 * distance d2[n], the triangle tri_id[n] it lies in (lowest triangle number on exact ties) and the barycentric weights bary[3n] of
 * that triangle's three corners at the closest point (vertex -> (1,0,0), edge -> (1-q, q, 0)).  Any output may be NULL. */
int gingr_mesh_closest_points(gingr_ctx *ctx, int64_t n_points, const double *points, int64_t n_vertices, const double *vertices,
                              int64_t n_triangles, const int32_t *triangles, double *cp_xyz, double *d2, int32_t *tri_id,
                              double *bary);
/* model.newReference(newReference, interpolator) (scalismo PointDistributionModel; used as
 * SimpleRegistrator.scala:89-90 with NearestNeighborInterpolator and as examples/DemoHelper/DemoDatasetLoader.scala:58-62 with
 * TriangleMeshInterpolator3D): mean and every basis function of the new point i are the fixed combination
 * sum_k weights[3i+k] * (value at source point vertex_ids[3i+k]); eigenvalues and rank are unchanged, nothing is
 * re-orthonormalised.  Nearest neighbour: ids (nn, nn, nn), weights (1, 0, 0); triangle mesh: the corners of the triangle from
 * gingr_mesh_closest_points and its barycentric weights.  The basis is gathered in HBM from the source model (a complete model
 * of the same context); the result is a model like any other (row shard [row_begin, row_end) of M_new; row_end <= 0 = M_new). */
int gingr_model_new_reference(gingr_ctx *ctx, const gingr_model *src, int64_t M_new, const double *new_ref,
                              const int32_t *vertex_ids, const double *weights, int64_t row_begin, int64_t row_end,
                              gingr_model **out);

/* ---- probabilistic proposal (single shard) ------------------------------------------------
 * update(current, probabilistic = true): the shape proposal is posterior.sample() instead of posterior.mean
 * (G/api/GingrAlgorithm.scala:211).  z[r] are standard-normal draws from the HOST's random generator (the JVM's
 * scalismo.utils.Random there, numpy here); the sample is a + L^-T z with L L^T = I + Q^T L Q, which has exactly the
 * distribution of scalismo's SVD-parameterised posterior sample.  One iteration per call. */
int gingr_fitter_update_cpd_sample_async(gingr_fitter *f, const gingr_cpd_params *p, const double *z);
int gingr_fitter_update_icp_sample_async(gingr_fitter *f, const gingr_icp_params *p, const double *z);
/* posterior(of the fitter's CURRENT state).gp.logpdf(posterior.coefficients(mesh)): the quantity
 * GeneratorWrapperStochastic.logTransitionProbability evaluates (G/api/sampling/generators/GeneratorWrapperStochastic.scala:42-63).
 * mesh_xyz[3*M]; the state is not modified.  rank <= 512.  A failed posterior gives GINGR_ERR_NOT_SPD / _NONFINITE
 * (the reference returns -infinity in that case).  Like the reference's Memoize(computePosterior, 10)
 * (G/api/GingrAlgorithm.scala:68) the fitter remembers which state the correspondences / Gram / right-hand side in its buffers
 * belong to: when gingr_fitter_set_state writes exactly that state again (same coefficients, pose, sigma2, flavour and
 * parameters), the update and these queries reuse them instead of recomputing (single shard; invisible in the results). */
int gingr_fitter_posterior_logpdf_cpd(gingr_fitter *f, const gingr_cpd_params *p, const double *mesh_xyz, double *logpdf);
int gingr_fitter_posterior_logpdf_icp(gingr_fitter *f, const gingr_icp_params *p, const double *mesh_xyz, double *logpdf);

/* The retry counter of the probabilistic proposal (G/api/GingrAlgorithm.scala:69-70,196-202,210: `retryCounter`, a private var
 * of the algorithm INSTANCE): a sampled proposal whose posterior cannot be computed returns the state unchanged up to 10 times in
 * a row before the state is marked ModelFlexibilityError; every successful posterior gives one retry back (at most 10).  The
 * fitter is the device-side stand-in of one algorithm instance, so the counter lives with it (it survives gingr_fitter_set_state).
 * set_to >= 0 writes the counter, set_to < 0 only reads; value_out (nullable) receives the value.  Synchronises. */
int gingr_fitter_retry_counter(gingr_fitter *f, int32_t set_to, int32_t *value_out);

/* ---- row-sharded update, host-driven exchange (multi-GPU) -----------------------------------------------------
 * One iteration = phases 0..GINGR_NUM_PHASES-1.  After phase p < GINGR_NUM_SEGMENTS the host all-reduces (sum, float64)
 * exchange segment p across shards (RCCL over xGMI via torch.distributed, or nothing for one shard), then runs phase p+1.
 * The exchange buffer is device memory owned by the library; gingr_fitter_exchange gives its address and the
 * [offset, count) of every segment in float64 elements.
 * Stream contract: phase kernels are enqueued on the context's stream (gingr_ctx_set_stream / gingr_ctx_get_stream).  A host that
 * runs its collective on another stream must order the two itself: wait for the context's stream before the all-reduce and for
 * the collective before the next phase (gingr_amd/sharded.py does exactly that when the streams differ).  The in-library
 * alternative that needs no host collective at all is the device group below (gingr_group_*).
 *   phase 0 -> segment 0: den partial column sums [N]           (the CPD column-sum exchange; unused for ICP)
 *   phase 1 -> segment 1: weighted Gram [rp*rp] + rhs [rp] + sigma2 sums [8]
 *   phase 2: replicated r x r algebra (posterior solve, projections and Umeyama from the model's one-off moments),
 *            state commit, new fit of the local rows -- nothing to exchange.
 */
#define GINGR_NUM_PHASES 3
#define GINGR_NUM_SEGMENTS 2
/* Row-sharded SURFACE ICP only (round 4): the rejection tests against the template itself need the whole posed template, so an
 * iteration starts with one more step -- phase GINGR_PHASE_GATHER writes this shard's rows of the fit into the full-fit buffer
 * ([3][M_total] planes, original point order, zeros elsewhere; gingr_fitter_fullfit_exchange gives its address), the host
 * all-reduces (sum) it across the shards = an all-gather, then phases 0, 1, 2 as above.  In the one-call protocols below it is
 * exchange segment GINGR_SEGMENT_FULLFIT of the callback. */
#define GINGR_PHASE_GATHER 3
#define GINGR_SEGMENT_FULLFIT 2
/* Reversed correspondence direction on row shards (round 5; G/api/registration/utils/ClosestPointRegistrator.scala:34-49): the queries
 * of that direction are the vertices of the (replicated) target, so they partition by index range -- every shard scans its range
 * against the gathered template and leaves, per vertex of the WHOLE template, the sum of its accepted target points and their
 * number ([4][M_total] = sum x, sum y, sum z, count; gingr_fitter_reversal_exchange gives the address).  Between phases 0 and 1 the
 * host all-reduces (sum) that buffer = exchange segment GINGR_SEGMENT_REVSUM of the callback; phase 1 then takes the shard's rows. */
#define GINGR_SEGMENT_REVSUM 3
int gingr_fitter_exchange(gingr_fitter *f, void **dev_ptr, int64_t offsets[GINGR_NUM_SEGMENTS],
                          int64_t counts[GINGR_NUM_SEGMENTS]);
int gingr_fitter_cpd_phase_async(gingr_fitter *f, const gingr_cpd_params *p, int32_t phase);
int gingr_fitter_icp_phase_async(gingr_fitter *f, const gingr_icp_params *p, int32_t phase);
/* The same protocol as ONE call per n iterations: the library runs the phases and calls `reduce(user, segment, device_ptr, count)`
 * wherever a segment has to be summed across the shards; the callback enqueues an in-place float64 sum all-reduce of those
 * `count` elements, ordered after everything already on the context's stream and before whatever is enqueued on it afterwards
 * (RCCL on that stream, or torch.distributed with that stream current), and returns 0 -- any other value aborts the update and
 * is returned as GINGR_ERR_STATE.  Nothing waits for the GPU: with an asynchronous collective all n iterations are enqueued
 * back to back.  ICP point-cloud correspondences need no segment-0 exchange (the rows are independent) and the callback is not
 * called for it.  Replaces the host loop of three phase calls + two collectives per iteration (gingr_amd/sharded.py);
 * reference: the per-iteration work of G/api/GingrAlgorithm.scala:192-254 on a row shard. */
typedef int (*gingr_allreduce_fn)(void *user, int32_t segment, void *device_ptr, int64_t count);
int gingr_fitter_update_cpd_sharded_async(gingr_fitter *f, const gingr_cpd_params *p, int32_t n_iterations, gingr_allreduce_fn reduce,
                                          void *user);
int gingr_fitter_update_icp_sharded_async(gingr_fitter *f, const gingr_icp_params *p, int32_t n_iterations, gingr_allreduce_fn reduce,
                                          void *user);

/* Any flavour, deterministic or sampled, as ONE call (round 4: the surface correspondence -- the reference's default ICP method,
 * G/api/registration/config/ICP.scala:63 -- the probabilistic proposal and the transition density are row-sharded too).
 *   flavour 0 CPD (cp), 1 ICP point cloud (ip), 2 ICP surface (ip; gingr_fitter_set_meshes on every shard with the triangles of the
 *   WHOLE template -- vertex ids of the full model -- and of the target; the shard's queries are its own rows, the tests against the
 *   template itself see all of it through the gathered fit, exchange segment GINGR_SEGMENT_FULLFIT).
 *   z (nullable): rank standard-normal draws = update(current, probabilistic = true), n_iterations must be 1; the same z on every
 *   shard (the sample a + L^-T z is replicated r x r algebra).
 * gingr_fitter_posterior_logpdf_sharded: posterior(state).gp.logpdf(posterior.coefficients(mesh))
 * (G/api/sampling/generators/GeneratorWrapperStochastic.scala:42-63) with mesh_xyz_full = the FULL mesh [3 M_total] on every shard
 * (each takes its rows); Q0^T e travels in the tail of segment 1, the log-density kernel is replicated; synchronises.
 * Reversed correspondence direction (ICP.scala:46-48; gingr_fitter_set_correspondence_direction after gingr_fitter_set_meshes): every
 * shard scans its index range of the target queries against the gathered template; the per-template-vertex sums are all-reduced
 * (GINGR_SEGMENT_REVSUM, between phases 0 and 1) and every shard keeps the observations of its own rows.
 * gingr_fitter_fullfit_exchange / gingr_fitter_reversal_exchange: address / element count of the full-fit buffer / the reversal sums
 * for a host that drives the phases itself. */
int gingr_fitter_update_sharded_async(gingr_fitter *f, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                      int32_t n_iterations, const double *z, gingr_allreduce_fn reduce, void *user);
int gingr_fitter_posterior_logpdf_sharded(gingr_fitter *f, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                          const double *mesh_xyz_full, gingr_allreduce_fn reduce, void *user, double *logpdf);
int gingr_fitter_fullfit_exchange(gingr_fitter *f, void **dev_ptr, int64_t *count);
int gingr_fitter_reversal_exchange(gingr_fitter *f, void **dev_ptr, int64_t *count);
/* The gather of the fit through a real ALL-GATHER instead of the zero-padded all-reduce (half the wire bytes, no reduction) for hosts
 * that have one (the library's own RCCL path uses ncclAllGather this way).  The shards must be the balanced contiguous partition of
 * the rows over `world` shards (the first M_total % world shards hold one row more -- what gingr_group_* and gingr_amd.sharded use).
 *   gingr_fitter_gather_stage   enqueues the copy of this shard's rows (original order) into its slot of a staging buffer
 *                               [world][3][chunk], chunk = ceil(M_total / world), and returns the slot (send), the buffer (recv) and
 *                               the element count per rank (3 chunk): the host runs an in-place all-gather of exactly that shape on
 *                               the context's stream (ncclAllGather(send, recv, count, ncclDouble, ...));
 *   gingr_fitter_gather_finish  enqueues the kernel that spreads the slots over the planes of the full-fit buffer.
 * Together they replace phase GINGR_PHASE_GATHER + the all-reduce of segment GINGR_SEGMENT_FULLFIT; then phases 0, 1, 2 as above. */
int gingr_fitter_gather_stage(gingr_fitter *f, int32_t world, int32_t rank, void **send_ptr, void **recv_ptr, int64_t *count_per_rank);
int gingr_fitter_gather_finish(gingr_fitter *f, int32_t world);

/* ---- one Metropolis-Hastings step of GingrAlgorithm.run's chain in ONE call (G/api/GingrAlgorithm.scala:115-190; scalismo
 * MetropolisHastings.next = propose, evaluate both, transition ratio, accept / reject) ---------------------------------------------
 * The device state x (gingr_fitter_set_state, or what the previous step left) is the current sample.  The call enqueues
 *   x' = kind 0: update(x, probabilistic = true) with the r standard normals z (GeneratorWrapperStochastic.propose,
 *        G/api/sampling/generators/GeneratorWrapperStochastic.scala:28-40), or
 *        kind 1: the parameters a random-walk proposal chose (alpha, scalars incl. iteration + 1; the fit is re-instantiated on the
 *        device: GingrGeneratorWrapper.scala:28-39)
 *   log_q_forward  = logTransitionProbability(x, x') = posterior(x).gp.logpdf(posterior.coefficients(x.fit)): with step length 1 the
 *                    reference projects FROM.fit (GeneratorWrapperStochastic.scala:42-63), so this is a function of x alone -- the
 *                    log_q_backward of the step that produced x -- and is computed only with need_forward != 0 (else NaN, status -1)
 *   log_q_backward = logTransitionProbability(x', x) = posterior(x').gp.logpdf(posterior.coefficients(x'.fit))
 *   log_value      = sum over the first eval_points fit vertices of x' (0 = all) of log N(|v - closestPointOnSurface(v)|; 0, eval_sdev)
 *                    against the target surface (IndependentPointDistanceEvaluator.scala:54-70, ModelToTargetEvaluation)
 * and synchronises ONCE.  x' is the device state on return; alpha_out[r] / fit_out[3 M] (nullable) / res->scalars describe it.  A
 * density is -inf with its status GINGR_ERR_NOT_SPD / GINGR_ERR_NONFINITE when the posterior it needs fails (what the reference's Try
 * turns into Double.NegativeInfinity).  The host combines these with the densities of its random-walk components and decides; after a
 * rejection gingr_fitter_mh_restore makes x the device state again (asynchronous).  Every state's correspondences and Gram are
 * computed once (two-slot posterior memo).  Single shard, meshes set, flavour as in gingr_fitter_update_sharded_async. */
typedef struct gingr_mh_request {
    int32_t flavour, kind;
    const gingr_cpd_params *cpd; /* flavour 0 */
    const gingr_icp_params *icp; /* flavours 1, 2 */
    const double *z;             /* kind 0 */
    const double *alpha;         /* kind 1 */
    const gingr_state_scalars *scalars;
    double eval_sdev;
    int64_t eval_points;
    int32_t need_forward;
} gingr_mh_request;
typedef struct gingr_mh_result {
    gingr_state_scalars scalars; /* of x' */
    double log_value, dist_sum, dist_max;
    int64_t count;
    double log_q_forward, log_q_backward;
    int32_t forward_status, backward_status;
} gingr_mh_result;
int gingr_fitter_mh_step(gingr_fitter *f, const gingr_mh_request *request, double *alpha_out, double *fit_out, gingr_mh_result *result);
int gingr_fitter_mh_restore(gingr_fitter *f);

/* ---- native RCCL exchange: the row-sharded update with one process per GPU and the collective enqueued by the LIBRARY ------------
 * BASELINE.json north_star: "the N x M affinity/distance matrix shards by reference-point rows across the 8 GPUs of one node with an
 * RCCL all-reduce over xGMI for CPD column sums".  Every rank owns one context (one device, one stream), one row shard of the model
 * and one fitter, as with gingr_fitter_update_*_sharded_async; here the two per-iteration all-reduces are ncclAllReduce calls the
 * library itself places on the context's stream between the phase kernels -- n iterations are enqueued with no callback, no stream
 * hop and no host synchronisation.  librccl is bound at run time (dlopen; a copy already loaded in the process -- e.g. torch's -- is
 * shared), so a single-GPU host does not need it.
 *   bootstrap: rank 0 calls gingr_rccl_unique_id (128 bytes), the HOST distributes them to all ranks (a JVM host: its own channel;
 *   bench.py: the torch.distributed store), every rank calls gingr_ctx_rccl_init(ctx, id, world, rank) -- collective, returns when the
 *   communicator of the `world` ranks exists.  The communicator belongs to the context and dies with it.
 *   one-off: gingr_ctx_rccl_allreduce_async sums the basis moments (gingr_model_gram_exchange) before gingr_model_finalize.
 *   per n iterations: gingr_fitter_update_{cpd,icp}_rccl_async.
 * gingr_rccl_load (optional, before anything else): bind a specific librccl by path instead of the default search
 * (already-loaded librccl.so.1, then the loader's path).  gingr_ctx_rccl_info: what was bound (diagnostics; any pointer may be NULL).
 * Reference: none (the reference is single-process / single-device); the protocol is that of G/api/GingrAlgorithm.scala:192-254 on
 * a row shard as in the host-driven section above. */
#define GINGR_RCCL_UNIQUE_ID_BYTES 128
int gingr_rccl_load(gingr_ctx *ctx, const char *librccl_path);
int gingr_rccl_unique_id(gingr_ctx *ctx, void *id_bytes);
int gingr_ctx_rccl_init(gingr_ctx *ctx, const void *id_bytes, int32_t world, int32_t rank);
int gingr_ctx_rccl_info(gingr_ctx *ctx, int32_t *world, int32_t *rank, int32_t *version, char *library_path, int32_t path_capacity);
int gingr_ctx_rccl_allreduce_async(gingr_ctx *ctx, void *device_ptr, int64_t count);
int gingr_fitter_update_cpd_rccl_async(gingr_fitter *f, const gingr_cpd_params *p, int32_t n_iterations);
int gingr_fitter_update_icp_rccl_async(gingr_fitter *f, const gingr_icp_params *p, int32_t n_iterations);
/* any flavour / sampled proposal / transition density over the context's communicator: gingr_fitter_update_sharded_async and
 * gingr_fitter_posterior_logpdf_sharded with the library's own ncclAllReduce as the exchange */
int gingr_fitter_update_rccl_async(gingr_fitter *f, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                   int32_t n_iterations, const double *z);
int gingr_fitter_posterior_logpdf_rccl(gingr_fitter *f, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                       const double *mesh_xyz_full, double *logpdf);

/* ---- device group: the row-sharded update across the GPUs of ONE node from ONE host process (multi-GPU for a C / JVM host) ----
 * SURVEY.md section 8b "gingr_group_create(ndev, devs[], ...) wrapping the same calls with row-sharding".  The group owns one
 * context, one model shard (contiguous rows, the first M mod n shards hold one extra row) and one fitter per entry of `devices`;
 * the target cloud and the r-sized state are replicated.  Per iteration the two exchange segments above are summed across the
 * shards inside the library: every shard writes its partial segment to a send buffer, the shards exchange HIP events, and each
 * shard sums all send buffers in rank order (its own and, through the xGMI peer mapping, the remote ones) with one kernel --
 * a one-shot all-reduce; both messages are small (N doubles; rp*rp + rp + 8), so it is latency bound.  Every shard computes
 * bit-identical sums, the r x r solve is replicated, results do not depend on timing.  One host thread per device enqueues
 * that device's kernels; gingr_group_update_*_async returns when n iterations are ENQUEUED on every device
 * (gingr_group_synchronize / gingr_group_get_state wait for them).  Not thread-safe: one caller thread per group.
 * The same device may be listed several times (logical shards on one GPU: how the protocol is tested on a one-GPU box).
 * Reference: the reference is single-process / single-device; the calls mirror gingr_model_upload / gingr_gpmm_build_gaussian /
 * gingr_fitter_* one to one (same argument meaning; host arrays are always the FULL model / cloud / fit).
 * Errors: the gingr_status of the first failing shard; gingr_group_last_error has the text. */
typedef struct gingr_group gingr_group;
int gingr_group_create(int32_t ndev, const int32_t *devices, gingr_group **out);
void gingr_group_destroy(gingr_group *g);
int32_t gingr_group_size(const gingr_group *g);
const char *gingr_group_last_error(const gingr_group *g);
gingr_ctx *gingr_group_ctx(gingr_group *g, int32_t shard);      /* e.g. for the timing hooks of one shard */
int gingr_group_shard_rows(const gingr_group *g, int32_t shard, int64_t *row_begin, int64_t *row_end);
int gingr_group_model_upload(gingr_group *g, int64_t M_total, int32_t rank, const double *ref, const double *mean,
                             const double *basis_colmajor, const double *variance);
int gingr_group_gpmm_build_gaussian(gingr_group *g, int64_t M_total, const double *ref, int32_t n_kernels, const double *sigmas,
                                    const double *scalings, double relative_tolerance, int32_t max_rank);
int32_t gingr_group_model_rank(const gingr_group *g);
int gingr_group_set_target(gingr_group *g, int64_t N, const double *target_xyz);
int gingr_group_set_landmarks(gingr_group *g, int32_t n_lm, const int32_t *lm_pid, const double *lm_xyz, const double *lm_cov);
int gingr_group_set_options(gingr_group *g, int32_t global_transform, double step_length);
int gingr_group_set_state(gingr_group *g, const double *alpha, const gingr_state_scalars *s);
/* alpha[r] and scalars from shard 0 (replicated state), fit_xyz[3*M_total] gathered from all shards; synchronises */
int gingr_group_get_state(gingr_group *g, double *alpha, gingr_state_scalars *s, double *fit_xyz);
int gingr_group_update_cpd_async(gingr_group *g, const gingr_cpd_params *p, int32_t n_iterations);
int gingr_group_update_icp_async(gingr_group *g, const gingr_icp_params *p, int32_t n_iterations);
/* Round 4: the surface correspondence, the sampled proposal and the transition density through the group (the mirror of
 * gingr_fitter_set_meshes / _set_surface_method / gingr_fitter_update_sharded_async / gingr_fitter_posterior_logpdf_sharded; same
 * argument meaning: flavour 0 CPD, 1 ICP point cloud, 2 ICP surface; z nullable, then one iteration; triangles in the vertex ids of
 * the FULL model / target).  gingr_group_posterior_logpdf synchronises. */
int gingr_group_set_meshes(gingr_group *g, int64_t n_model_triangles, const int32_t *model_triangles, int64_t n_target_triangles,
                           const int32_t *target_triangles);
int gingr_group_set_surface_method(gingr_group *g, int32_t method);
/* gingr_fitter_set_correspondence_direction for every shard (ICP.scala:46-48).  With more than one shard the correspondence runs
 * against the GATHERED template (set the meshes first, also for the vertex-to-vertex flavour 1) and is sharded by QUERY range: every
 * shard scans its slice of the replicated target's vertices, the per-template-vertex sums of the slices are totalled by one more
 * exchange (GINGR_SEGMENT_REVSUM), and each shard then forms the observations of its own rows; Gram, right-hand side and everything
 * behind them stay sharded by rows.  gingr_group_set_meshes drops the direction again (as gingr_fitter_set_meshes does). */
int gingr_group_set_correspondence_direction(gingr_group *g, int32_t reversed);
int gingr_group_update_async(gingr_group *g, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int32_t n_iterations,
                             const double *z);
int gingr_group_posterior_logpdf(gingr_group *g, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                 const double *mesh_xyz_full, double *logpdf);
int gingr_group_synchronize(gingr_group *g);
/* How the group exchanges (diagnostics for a first run on real multi-GPU hardware): *distinct_devices = physical devices behind
 * the shards; *fine_grained = 1 when the peer-read send buffers are fine-grained device allocations (always the case when
 * distinct_devices > 1: gingr_group_set_target fails with GINGR_ERR_HIP instead of falling back to coarse-grained memory, which
 * a peer could not read coherently).  Either pointer may be NULL.  No reference counterpart (the reference is single-device). */
int gingr_group_exchange_info(const gingr_group *g, int32_t *distinct_devices, int32_t *fine_grained);

/* -------------------------------------------------------------- timing hooks
 * HIP-event timing of the dominant kernels on the context's stream (bench.py's live roofline measurement).
 * which: 0 = cpd_colsum, 1 = cpd_rowstats, 2 = gram, 3 = whole update, 4 = basis sweep (one streaming pass over Q0),
 * 5 = posterior solve (unfused tail only), 6 / 7 = the device group's exchange of segment 0 / 1 on this shard (from the record of
 * the shard's own event to the end of its sum kernel: includes the wait for the slowest peer; the host-driven sharded update records its two collectives there too), 8 = the
 * nearest-neighbour scan kernel alone.  Returns accumulated ms and launches since
 * the last reset.  Enabling adds two event records per launch. */
int gingr_ctx_timing_enable(gingr_ctx *ctx, int32_t enable);
int gingr_ctx_timing_read(gingr_ctx *ctx, int32_t which, double *total_ms, int64_t *launches);
int gingr_ctx_timing_reset(gingr_ctx *ctx);
/* Diagnostics of the nearest-neighbour scan (ClosestPointRegistrator.scala:139-145 is an unpruned scan of every pair; this one prunes
 * exactly): with counting enabled every gingr_nn / ICP launch on this context adds the distance tests it really executed to a device
 * counter (a second instantiation of the kernel; the default one carries no counter).  gingr_ctx_nn_counting(ctx, 1) switches it on
 * and clears the counter, (ctx, 0) switches it off; gingr_ctx_nn_tests waits for the stream and returns the count. */
int gingr_ctx_nn_counting(gingr_ctx *ctx, int32_t enable);
int gingr_ctx_nn_tests(gingr_ctx *ctx, int64_t *tests);

/* ---- classic Coherent Point Drift (the reference's `other/` CPD family as a second consumer of the
 * affinity statistics and of the Gaussian kernel block) ----------------------------------------------------------------------
 * One handle = CPDFactory(templatePoints, lambda, beta, w).registerRigidly / registerAffine / registerNonRigidly(targetPoints)
 * (G/other/algorithms/cpd/CPDFactory.scala:28-80); kind 0 rigid (similarity: s, R, t; RigidCPD.scala:107-137), 1 affine
 * (AffineCPD.scala:35-60), 2 non-rigid (G + lambda sigma2 diag(1/P1)) W = diag(1/P1) P X - Y, TY = Y + G W
 * (NonRigidCPD.scala:45-88).  The state (TY, sigma2) of RigidCPD.Registration (:59-83) lives on the device; it starts at
 * (template, sum |y_m - x_n|^2 / (3 M N)).  gingr_classic_cpd_iterate runs n Iteration()s (:85-88) back to back and synchronises;
 * the caller owns the convergence test |sigma2' - sigma2| < tolerance (:70-78).  gingr_classic_cpd_get: any of ty_xyz [3 M],
 * sigma2, transform13 = {s, R or B row-major [9], t [3]} (rigid / affine, last Maximization), w_xyz [3 M] (non-rigid) may be NULL.
 * moving_xyz = the templatePoints [3 M], target_xyz = the targetPoints [3 N]. */
typedef struct gingr_classic_cpd gingr_classic_cpd;
int gingr_classic_cpd_create(gingr_ctx *ctx, int32_t kind, int64_t M, const double *moving_xyz, int64_t N, const double *target_xyz,
                             double lambda, double beta, double w, gingr_classic_cpd **out);
void gingr_classic_cpd_destroy(gingr_classic_cpd *h);
int gingr_classic_cpd_iterate(gingr_classic_cpd *h, int32_t n_iterations);
int gingr_classic_cpd_get(gingr_classic_cpd *h, double *ty_xyz, double *sigma2, double *transform13, double *w_xyz);
int gingr_classic_cpd_set(gingr_classic_cpd *h, const double *ty_xyz, double sigma2);

/* ---- classic rigid / similarity ICP (the reference's other/ baseline) ------------------------------------------------
 * G/other/algorithms/icp/RigidICP.scala:24-84 (Iteration / Registration), ICPFactory.scala:28-38, RigidICPRegistration.scala:24-46
 * with the two registrators of G/other/utils/PoseRegistrator.scala:27-43: kind 0 = RigidRegistrator3D
 * (LandmarkRegistration.rigid3DLandmarkRegistration about the origin), kind 1 = AffineRegistrator3D
 * (similarity3DLandmarkRegistration).  One iteration = closest target point of every template point (exact, lowest index on
 * ties), least-squares transform of the pairs, template <- transform(template); the template lives in HBM.
 * gingr_rigid_icp_iterate: distances[k] (nullable) = mean closest-point distance measured by iteration k BEFORE it moves the
 * template (RigidICP.scala:60-71,79-82) -- what Registration's convergence test compares.  _get: current template [3M] and the
 * last transform {s, R[9] row-major, t[3]} (either may be NULL).  _set: replace the template points. */
typedef struct gingr_rigid_icp gingr_rigid_icp;
int gingr_rigid_icp_create(gingr_ctx *ctx, int32_t kind, int64_t M, const double *moving_xyz, int64_t N, const double *target_xyz,
                           gingr_rigid_icp **out);
void gingr_rigid_icp_destroy(gingr_rigid_icp *h);
int gingr_rigid_icp_iterate(gingr_rigid_icp *h, int32_t n_iterations, double *distances);
int gingr_rigid_icp_get(gingr_rigid_icp *h, double *points_xyz, double *transform13);
int gingr_rigid_icp_set(gingr_rigid_icp *h, const double *points_xyz);


/* ---- optimal-step non-rigid ICP baselines (G/other/algorithms/icp/NonRigidOptimalStepICP.scala:31-284) ------------------------
 * The reference's loop is: closest-point correspondence of the current template mesh (ClosestPointTriangleMesh3D, the same
 * query as the GiNGR ICP path), then one sparse least-squares solve `A \ B`.  Both halves run on the device:
 *
 * gingr_fitter_set_fit_points: replace the shape the fitter's NEXT correspondence query looks at (normally the instance of the
 * current state) by explicit points [3M] -- the N-ICP template is no instance of a model.  Then
 * gingr_fitter_icp_surface_phase_async(f, p, 0) + gingr_fitter_get_surface_correspondence give (closest points, weights).
 *
 * gingr_nicp_solve: the least-squares step as its normal equations (dense on the device, blocked MFMA Cholesky), kind 0 = N-ICP-T
 * (:151-190, unknown n x 3 displacements), kind 1 = N-ICP-A (:241-283, one 4 x 3 affine map per vertex).  edges [2E]: the unique
 * vertex pairs p1 < p2 of the template triangles (trianglesToEdges :67-76); w [n], cp_xyz [3n]: the correspondence; lm_ids [L]
 * template vertices of the landmarks, lm_target_xyz [3L] their targets (UL); alpha stiffness, beta landmark weight, gamma the
 * fourth diagonal entry of G (N-ICP-A).  out_xyz [3n]: the moved template; out_lm_xyz (nullable, [3L]): the moved landmark
 * vertices (DL X of N-ICP-A).  Quirks of the reference kept: N-ICP-T's landmark rows put their ones into the FIRST L columns and
 * are not scaled by beta (only their right-hand side is); N-ICP-A zeroes the weights of the landmark vertices. */
int gingr_fitter_set_fit_points(gingr_fitter *f, const double *fit_xyz);
int gingr_nicp_solve(gingr_ctx *ctx, int32_t kind, int64_t n, const double *moving_xyz, int64_t n_edges, const int32_t *edges,
                     const double *w, const double *cp_xyz, int32_t n_lm, const int32_t *lm_ids, const double *lm_target_xyz, double alpha,
                     double beta, double gamma, double *out_xyz, double *out_lm_xyz);

#ifdef __cplusplus
}
#endif
#endif /* GINGR_HIP_H */

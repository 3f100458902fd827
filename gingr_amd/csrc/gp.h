// Low-rank GP kernels (gp.hip): device data structures and launchers.  Internal to libgingr_hip.so.
#pragma once

#include "common.h"
#include "svd3.h"

#include <functional>

// Regularisation of PointDistributionModel.coefficients (scalismo DiscreteLowRankGaussianProcess.coefficients:
// "val sigma2 = 1e-5"); reached from G/api/GingrAlgorithm.scala:215,236.
#define GINGR_COEFF_NOISE 1e-5

// Device-resident pose/state scalars.  R is always euler_to_rot(euler) (the reference rebuilds Rotation(phi,theta,psi)
// from the stored Euler angles every iteration, G/api/ModelFittingParameters.scala:39-41,60-64).
struct DevState {
    double R[9];  // row-major
    double euler[3];
    double center[3];
    double t[3];
    double scale;
    double sigma2;
    int32_t iteration;
    int32_t status;  // gingr_fitting_status
    int32_t err;     // sticky numerical-failure flag of the CURRENT update (0 ok, GINGR_ERR_NONFINITE, GINGR_ERR_NOT_SPD)
    int32_t pad;     // error code of the last update (0: none), for the host
    int32_t stopped; // the run's stopping rule fired on this state (PostSolveArgs::stop_threshold): later updates leave it as it is
    int32_t pad2;
};

// DevState from the scalars the host pushed (one thread; state_init_kernel, and the tail of the launch that brings a Metropolis-Hastings
// step's random-walk parameters: fitter.hip, mh_begin_kernel)
__device__ inline void state_init_body(DevState *st, const gingr_state_scalars *h, double *zero_slot) {
    if (zero_slot) *zero_slot = 0.0;  // see SweepArgs::absmax_slot
    for (int q = 0; q < 3; ++q) {
        st->euler[q] = h->euler[q];
        st->center[q] = h->center[q];
        st->t[q] = h->translation[q];
    }
    euler_to_rot(st->euler, st->R);
    st->scale = h->scale;
    st->sigma2 = h->sigma2;
    st->iteration = h->iteration;
    st->status = h->status;
    st->err = 0;
    st->pad = 0;
    st->stopped = 0;
    st->pad2 = 0;
}

// Candidate global alignment produced by the Umeyama step (R2 went through the Euler parameterisation).
struct DevPose {
    double R[9];
    double euler[3];
    double t[3];
    double center[3];
    double scale;
};

// One-off moments of the basis (computed at upload, summed over ALL shards before finalize).  With them every step of the
// update after the posterior solve -- both coefficient projections and the Umeyama sums -- is O(r^2) replicated algebra
// with no pass over Q0 and no collective ("moment form"):
//   p~_i = ref_i + mean_i - c0
//   S_tot      = sum_i Q0_i^T Q0_i                        [rp*rp]   (= sum_d S[d][d])
//   S[d][e]    = sum_i Q0[3i+d]^T Q0[3i+e]                 [9][rp*rp]
//   V[d][e]    = sum_i Q0[3i+d]^T p~_i[e]                  [9][rp]
//   W[d]       = sum_i Q0[3i+d]^T                          [3][rp]
struct MomentLayout {
    int32_t rp;
    __host__ __device__ int64_t stot() const { return 0; }
    __host__ __device__ int64_t S(int d, int e) const { return (int64_t)rp * rp * (1 + d * 3 + e); }
    __host__ __device__ int64_t V(int d, int e) const { return (int64_t)rp * rp * 10 + (int64_t)rp * (d * 3 + e); }
    __host__ __device__ int64_t W(int d) const { return (int64_t)rp * rp * 10 + (int64_t)rp * (9 + d); }
    __host__ __device__ int64_t total() const { return (int64_t)rp * rp * 10 + (int64_t)rp * 12; }
};

// What the post-solve kernel reads besides the state (gp.hip: post_solve_kernel).  Binv = (S_tot / eps + I)^-1 is applied to every
// term of the second coefficient projection up front -- to the constant moment vectors once per model (pvec), to S[d][e] alpha and
// S[d][e] alpha_1 by the mat-vec launch (zbuf) -- so that the kernel itself touches no r x r matrix.
struct PostVec {
    int32_t rp;
    // vectors of gingr_model::pvec (entry-major: entry i of vector v at pvec[i * kRows + v]; then 16 scalars)
    static constexpr int kV = 0;    // [9]  V[d][e]
    static constexpr int kW = 9;    // [3]  W[d]
    static constexpr int kBV = 12;  // [9]  Binv V[d][e]
    static constexpr int kBW = 21;  // [3]  Binv W[d]
    static constexpr int kRows = 24;
    __host__ __device__ int64_t consts() const { return (int64_t)kRows * rp; }  // Pp[9] = sum p~ p~^T, Ps[3] = sum p~, c0[3], n = M_total
    __host__ __device__ int64_t total() const { return consts() + 16; }
    // vectors of the fitter's zbuf (entry-major [rp][28], written by launch_post_matvecs)
    static constexpr int kZa = 0;    // [9]  S[d][e] alpha
    static constexpr int kBSa = 9;   // [9]  (Binv S[d][e]) alpha
    static constexpr int kBTa = 18;  // [9]  (Binv S[d][e] C) a
    static constexpr int kA1 = 27;   //      alpha_1 = C a
    static constexpr int kZRows = 28;
};

// Zero rows behind the last basis row of every model (model_create_impl allocates and clears them): the wide Gram pass
// (gp_wide.hip) fetches whole 16-row steps, two of them ahead, without clamping its addresses -- 48 rows; three times that for the
// moment blocks, which take the basis as M rows of width 3 rp (launch_moment_grams).
constexpr int kBasisRowSlack = 144;

struct gingr_model {
    gingr_ctx *ctx = nullptr;
    int64_t M_total = 0, row_begin = 0, row_end = 0, M = 0;  // M = local points
    int32_t r = 0, rp = 0;                                    // rank and rank padded to a multiple of 16
    double *Q0 = nullptr;     // [3M + kBasisRowSlack][rp] row-major, Q0[row][k] = U[row][k] * sqrt(lambda_k), zero padded
    double *ref = nullptr;    // SoA [3][M]
    double *mean = nullptr;   // SoA [3][M]
    double *mom = nullptr;    // MomentLayout: local sums until finalize (the exchange buffer of the one-off all-reduce)
    // Rows live on the device in Morton (Z-curve) order of the local reference points so that a workgroup's points are
    // spatially coherent (exact-zero tile culling in the CPD passes).  perm[s] = original local index of device row s.
    int32_t *perm = nullptr;
    std::vector<int32_t> hperm, hiperm;  // host copies: device position -> original, original -> device position
    int32_t *iperm = nullptr;            // device copy of hiperm (local original index -> device position): the fit gather of a shard
    std::vector<double> h_full_pts;      // host: ref + mean of ALL M_total points, interleaved xyz (spatial order of a shard's triangles)
    double *Binv = nullptr;   // [rp*rp] (S_tot/eps + I)^-1, valid after finalize
    double *eigV = nullptr;   // [r*r] eigenvectors of S_tot = Q^T Q (column k, row stride r) and
    double *eigL = nullptr;   // [r] its eigenvalues (descending): uniform-weight posterior (launch_posterior_solve_eig); rank <= 192
    bool eig_ready = false;  // decided once by model_finalize_impl (fitter.hip), identically on every shard
    double *cmat = nullptr;   // [19][rp*rp], valid after finalize: [0] C = Binv S_tot / eps (alpha_1 = C a),
                              // [1 + 3d + e] Binv S[d][e], [10 + 3d + e] Binv S[d][e] C  (Binv S[d][e] alpha_1 = (Binv S[d][e] C) a)
    double *pvec = nullptr;   // PostVec: the r-vectors and scalars the post-solve kernel reads, valid after finalize
    double c0[3] = {0, 0, 0};  // centroid of the FULL reference: fixed centring point of the Umeyama sums
    double Pp[9] = {0};        // sum_i p~_i p~_i^T over the FULL model (host, identical on every shard)
    double Ps[3] = {0};        // sum_i p~_i
    std::vector<double> variance;  // host copy of lambda (gingr_model_download)
    bool finalized = false;
};

// gp.hip: basis rows of a model on a new reference as fixed convex combinations of three source rows (gingr_model_new_reference)
void launch_interp_pack(gingr_ctx *ctx, const double *Qs, int32_t rp, const int32_t *inv_src, const int32_t *ids, const double *w,
                        const int32_t *perm_new, int64_t row_begin, int64_t M, double *Q0);

// fitter.hip: common part of model construction; fill_basis writes m->Q0 on ctx->stream (see gingr_model_upload)
int model_create_impl(gingr_ctx *ctx, int64_t M_total, int32_t rank, const double *ref, const double *mean,
                      const double *variance, int64_t row_begin, int64_t row_end,
                      const std::function<int(gingr_model *)> &fill_basis, gingr_model **out);

// fitter.hip hooks for the device group (group.hip): where phases 0 / 1 write this shard's partial exchange segments, and where
// the gather step of a sharded surface update writes the shard's contribution to the full fit (nullptr: in place)
void fitter_set_partial_output(gingr_fitter *f, double *base);
void fitter_set_partial_fullfit(gingr_fitter *f, double *base);
// the sharded update (fitter.hip) for the other translation units: flavour 0 CPD, 1 ICP point cloud, 2 ICP surface; z nullable (sampled proposal)
// gather (nullable): the host's own all-gather of the fit -- 0 done, > 0 not possible here (fall back to the zero-padded all-reduce), < 0 failed
typedef int (*fitter_gather_fn)(void *user, gingr_fitter *f);
// split_native: the caller's `reduce` is the library's own RCCL all-reduce (user = the context) and honours ctx->exchange_stream:
// GINGR_OPT_SPLIT_EXCHANGE may then run the CPD column-sum exchange in two halves
int fitter_sharded_update(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int32_t n_iterations,
                          const double *z, gingr_allreduce_fn reduce, void *user, fitter_gather_fn gather = nullptr, bool split_native = false);
int fitter_sharded_logpdf(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, const double *mesh_xyz_full,
                          gingr_allreduce_fn reduce, void *user, double *logpdf, fitter_gather_fn gather = nullptr);
int fitter_run_phase(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int phase);
int fitter_logpdf_prepare(gingr_fitter *f, const double *mesh_xyz_full);  // before the exchange of segment 1
int fitter_logpdf_finish(gingr_fitter *f, double *logpdf);                // behind it
int fitter_set_zrand(gingr_fitter *f, const double *z);                    // nullable: the next phase 2 draws a sample (device group)
double *fitter_fullfit(gingr_fitter *f);                                   // [3][M_total] or nullptr
void fitter_set_partial_revsum(gingr_fitter *f, double *base);             // where phase 0 of the reversed direction leaves its sums
double *fitter_revsum(gingr_fitter *f);                                    // [4][M_total] or nullptr
gingr_ctx *fitter_ctx(gingr_fitter *f);
// the all-gather of the fit (rccl_exchange.hip): whether THIS shard could take part, and what all ranks agreed on (-1: not yet)
bool fitter_gather_possible(gingr_fitter *f, int32_t world, int32_t rank);
int fitter_gather_agreed(gingr_fitter *f, int32_t world);
void fitter_set_gather_agreed(gingr_fitter *f, int32_t world, int agreed);
const gingr_model *fitter_model(gingr_fitter *f);

// ---- the device-side choice between the Gram downdate and the pass over the basis (surface ICP) -----------
// counts[b] = zero-weight vertices of block b of obs_points_kernel (256 vertices each; every launch rewrites all of them: nothing to
// clear).  A kernel that carries a gate sums the counts in its prologue; "many" = more than one vertex in eight has weight 0 -- a
// fixed rule, independent of timing: both forms of the Gram matrix round differently (fitter.hip: kGramDowndateMinRows).
struct ZeroGate {          // (plain data: ZeroGate{} and memset-zeroed argument blocks mean "no gate")
    const int32_t *counts;  // nullptr: no gate (the kernel always runs)
    int32_t nblocks;
    int32_t run_if_many;    // the kernel runs when "many" == (run_if_many != 0)
    int64_t M;
};

// ZeroGate (gp.h): does this launch run?  Called by ALL threads of the workgroup (one barrier pair); the same answer in every workgroup
// of every launch that carries the same gate.
__device__ __forceinline__ bool gate_open(const ZeroGate &g) {
    if (!g.counts) return true;
    __shared__ int gate_sum;
    if (threadIdx.x == 0) gate_sum = 0;
    __syncthreads();
    int s = 0;
    for (int b = threadIdx.x; b < g.nblocks; b += blockDim.x) s += g.counts[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0 && s != 0) atomicAdd(&gate_sum, s);  // (integers: the order does not matter)
    __syncthreads();
    const bool many = (int64_t)gate_sum * 8 > g.M;
    return many == (g.run_if_many != 0);
}

// ---- basis sweeps ------------------------------------------------------------------------------------------
enum SweepMode {
    SWEEP_RHS = 0,      // T only: out[k] = sum_i Q0_i^T e_i, e from evec planes
    SWEEP_PROJ1 = 1,    // F(a) then T with e = Q0_i a
    SWEEP_SHAPES = 2,   // F(alpha_c, alpha): newshape planes + Umeyama partial sums
    SWEEP_PROJ2 = 3,    // T: e = R2^T (newshape - t2) - ref - mean
    SWEEP_FIT = 4,      // F(alpha): xyz = s (R (ref + mean + v - c) + c + t) from DevState
    SWEEP_POSED = 5,    // F(a): xyz = R (ref + mean + v - c) + c + t  (posterior mean mesh; no scale)
    SWEEP_RHS_ICP = 6   // T with e formed in the pass: the ICP observation of row i is target[icp_idx[i]], weight 1 / sigma2 (obs_points_kernel)
};

struct SweepArgs {
    const double *Q0;
    const double *ref;
    const double *mean;
    int64_t M;
    int32_t rp;
    const double *coef0;   // forward coefficient vector(s), [rp] each
    const double *coef1;
    const double *evec;    // SoA [3][M] (SWEEP_RHS)
    const double *shape_in;  // SoA [3][M] (SWEEP_PROJ2)
    double *shape_out;     // SoA [3][M] (SWEEP_SHAPES: newshape; SWEEP_FIT / SWEEP_POSED: result)
    const DevState *state;
    const DevPose *pose;
    double c0[3];
    double *partial;       // [nblocks][rp] transposed partials or [nblocks][24] Umeyama partials
    double *out;           // reduced result: [rp] or [24]
    double *zero_slot;     // nullable: one double the pass clears (consumed by a LATER launch on the stream)
    // SWEEP_FIT, nullable (round 4): qboxes[q] = {lo[3], hi[3]} of the 64-point quarter q of the shape written (the quarter boxes of
    // launch_tile_bbox, i.e. boxes + 6 ntiles), box_centre[3] on the device, and *absmax_slot = max(*absmax_slot, largest
    // |coordinate - centre|) -- the slot must have been cleared by an EARLIER launch on the stream (post_solve_kernel, state_init_kernel)
    double *qboxes;
    const double *box_centre;
    double *absmax_slot;
    int32_t no_reduce;     // != 0: leave the [nblocks][rp] partials in `partial` (the phase-1 finalize kernel adds them up)
    // SWEEP_RHS_ICP: matched target positions, the target planes (n_targets points), the landmark mask (nullable); weight / e are
    // also written out.  A position outside [0, n_targets) -- the searches leave -1 for a query whose distances are all NaN -- gives
    // a NaN observation (the posterior then fails through the normal status path) instead of a read outside the target allocation.
    const int32_t *icp_idx;
    const double *tx, *ty, *tz;
    int64_t n_targets;
    const int32_t *lm_mask;
    double *weight_out, *evec_out;
    const DevState *frame;  // SWEEP_PROJ2 (nullable): project in the rigid frame of this state instead of `pose`
    ZeroGate gate;          // SWEEP_RHS: the pass leaves at once when the gate says so (memset-zeroed args: no gate)
};

int sweep_num_blocks(int64_t M);
int64_t sweep_ws_doubles(int64_t M, int32_t rp);
void launch_sweep(gingr_ctx *ctx, SweepMode mode, const SweepArgs &a);

// ---- weighted Gram (MFMA f64) -----------------------------------------------------------------------------
int64_t gram_ws_doubles(int64_t M, int32_t rp);
// G[rp*rp] (full symmetric) = sum_i w_i Q0_i^T Q0_i over local points; weight == nullptr means w = 1
// returns the number of slab partials in ws; G == nullptr leaves them unreduced (launch_phase1_finalize adds them up)
int launch_gram(gingr_ctx *ctx, const double *Q0, int64_t M, int32_t rp, const double *weight, double *ws, double *G,
                const double *evec = nullptr, double *rhs_partial = nullptr, bool *rhs_done = nullptr, const ZeroGate *gate = nullptr);
// evec ([3][M] planes) + rhs_partial ([slabs][rp]) + rhs_done: when the triangle kernel runs (rp <= 112) it also leaves the slab
// partials of Q0^T evec and sets *rhs_done -- the caller then skips its SWEEP_RHS pass and reduces over the returned slab count.

// gp_wide.hip: the same partials for rp >= 128 (eight waves share the triangle, several workgroups per slab past 136 tiles); the
// right-hand-side partials always ride along when rhs_partial is given.  ws: gram_wide_ws_doubles(M, rp) doubles (slab partials, then
// the row-indexed {w, e} array of row_expand_kernel).  Returns the slab count.
int64_t gram_wide_ws_doubles(int64_t M, int32_t rp);
int launch_gram_wide(gingr_ctx *ctx, const double *Q0, int64_t M, int32_t rp, const double *weight, double *ws, const double *evec,
                     double *rhs_partial, const ZeroGate *gate = nullptr);

// the same pass over `rows` plain rows of width rp (unweighted, 48 zero rows behind them); ws: gram_rows_ws_doubles(rows, rp)
int64_t gram_rows_ws_doubles(int64_t rows, int32_t rp);
int launch_gram_rows(gingr_ctx *ctx, const double *Z, int64_t rows, int32_t rp, double *ws);

// one launch for the reductions at the end of phase 1 (gp.hip: phase1_finalize_kernel)
struct Phase1FinalizeArgs {
    int32_t rp;
    const double *gram_partial;   // [nslabs][rp*rp] (upper patches); nslabs == 0: G is not touched, unless scaled_src is given
    int32_t nslabs;
    // nslabs == 0 and scaled_src != nullptr: G = scaled_contribute ? scaled_src / sigma2 : 0 (uniform-weight ICP: the model's Q^T Q)
    // nslabs  > 0 and scaled_src != nullptr: G = ((scaled_contribute ? scaled_src : 0) - sum of the partials) / sigma2 (0 / 1 weights of
    //                                        the surface ICP: the partials hold Q^T Q of the zero-weight rows, launch_gram_downdate)
    const double *scaled_src;
    const double *sigma2;
    int32_t scaled_contribute;
    double *G;
    const double *sweep_partial;  // [sweep_blocks][rp]
    int32_t sweep_blocks;
    double *rhs;
    int32_t scalar_mode;          // 0: clear the 8 scalars (ICP); 1: the four sums of the CPD passes from `part`
    const double *part;           // GINGR_SCALAR_PART block partials (affinity.hip)
    double *scalars_local;        // nullable: the shard's own copy {Np, xPx, trPXY, yPy}
    double *sc8;                  // the 8 scalars of the exchange segment
    int32_t contribute_xpx;
    // gate.counts != nullptr: when the gate says "many" the launch in front was the weighted pass over the basis instead of the downdate:
    // alt_nslabs slab partials hold the weighted Gram itself (no scaled_src), the right-hand-side partials are alt_nslabs rows
    ZeroGate gate;
    int32_t alt_nslabs;
};
void launch_phase1_finalize(gingr_ctx *ctx, const Phase1FinalizeArgs &a);
// partial[b] = sum over the vertices i of slab b with weight[i] == 0 of Q0_i^T Q0_i (full rp x rp, zeros when the slab has none);
// returns the slab count (<= 256: the workspace of launch_gram is large enough).  Any rp (112-column patches above 112).
int launch_gram_downdate(gingr_ctx *ctx, const double *Q0, int64_t M, int32_t rp, const double *weight, double *ws,
                         const ZeroGate *gate = nullptr);

// ---- observations ------------------------------------------------------------------------------------------
// CPD: yhat = y + (PX/P1 - y), weight = 1/(sigma2*lambda/P1)  (CPD.scala:37-46,126); e = w (R^T(yhat - c - t) - (ref - c) - mean)
void launch_obs_cpd(gingr_ctx *ctx, const gingr_model *m, const DevState *st, Cloud fit, const double *P1,
                    const double *PX, double lambda, const int32_t *lm_mask, double *weight, double *evec);
// ICP: obs = target[idx], weight = 1/sigma2  (ICP.scala:90-92)
void launch_obs_icp(gingr_ctx *ctx, const gingr_model *m, const DevState *st, Cloud target, const int32_t *idx,
                    const int32_t *lm_mask, double *weight, double *evec);
// generic: obs points given as SoA planes with per-point weights
// zero_counts (nullable, [ceil(M / 256)]): the number of zero-weight vertices per 256-vertex block (ZeroGate::counts)
void launch_obs_points(gingr_ctx *ctx, const gingr_model *m, const DevState *st, const double *obs_soa,
                       const double *weight_in, double *weight, double *evec, const int32_t *lm_mask = nullptr,
                       int32_t *zero_counts = nullptr);
// landmarks with full 3x3 covariance added into G and rhs of the (reduced) exchange segment; local pids, local rows only
void launch_landmarks(gingr_ctx *ctx, const gingr_model *m, const DevState *st, int32_t n_lm, const int32_t *lm_pid_local,
                      const double *lm_xyz, const double *lm_cov, double *G, double *rhs);

// ---- small dense kernels ------------------------------------------------------------------------------------
// a = (I + G)^-1 rhs  by Cholesky (work: posterior_work_doubles(rp)); sets st->err on failure
// zrand (nullable, [r] on the device): standard-normal draws; the result is then a posterior SAMPLE a + L^-T z
void launch_posterior_solve(gingr_ctx *ctx, int32_t r, int32_t rp, const double *G, const double *rhs, const double *zrand,
                            double *work, double *a, DevState *st);
// out2[0] = posterior.gp.logpdf(posterior.coefficients(mesh)), out2[1] != 0: the posterior failed; qte = Q0^T e (model-frame
// residual).  fx ([rp*rp + 2*rp], nullable): receives the state-only part of the computation (cached == false) or provides it
// (cached == true: only the mesh-dependent part runs).  sync (2 zero-initialised words on the device) + a launch number `epoch`
// that never repeats: r <= 128 then runs the two factorisations on two workgroups; keep_factor == false lets that form skip the
// write-back of the state-only part (fx then only carries the posterior coefficients between the two workgroups).
int launch_posterior_logpdf(gingr_ctx *ctx, int32_t r, int32_t rp, const double *G, const double *rhs, const double *Stot,
                            const double *qte, double *fx, bool cached, double *work, double *out2, unsigned *sync = nullptr,
                            unsigned epoch = 0, bool keep_factor = true, double *nfac = nullptr);
// a + L^-T z from what the two-workgroup log-density kernel left behind (nfac: [rp*rp] factor of I + G, [16*rp] W rows; a_mean:
// the posterior coefficients): the sampled proposal of a state whose posterior has been factored already
void launch_posterior_sample_cached(gingr_ctx *ctx, int32_t r, int32_t rp, const double *nfac, const double *a_mean, const double *zrand,
                                    double *a, DevState *st);
// a = (I + S_tot / sigma2)^-1 rhs through the one-off eigen-decomposition S_tot = V diag(lam) V^T: a = V ((V^T rhs) / (1 + lam / sigma2)).
// The posterior of point-cloud ICP without landmarks -- every row weighs 1 / sigma2 (ICP.scala:90-92) -- needs no factorisation.
void launch_posterior_solve_eig(gingr_ctx *ctx, int32_t r, int32_t rp, const double *eigV, const double *eigL, const double *sigma2,
                                const double *rhs, double *a, DevState *st);
// eigen-decomposition of the leading n x n block of the symmetric G (row stride ldg): evals [n] descending, Vs [n*n] (Vs[i*n + k] =
// component i of eigenvector k); the register kernel of eig.hip up to 192 columns, the two-sided cyclic Jacobi of gpmm.hip above that
// and for numerically singular matrices (gpmm.hip sym_eig); synchronises the stream
int launch_jacobi_eig(gingr_ctx *ctx, const double *G, int32_t ldg, int32_t n, double *evals, double *Vs);
// eig.hip: one-sided register Jacobi on the Cholesky factor, n <= kSymEigColsMaxN, up to three problems per launch (see sym_eig, gpmm.hip)
constexpr int kSymEigColsMaxN = 192;
int64_t sym_eig_cols_work_doubles(int32_t n);
void launch_sym_eig_cols(gingr_ctx *ctx, int count, const double *const *G, const int32_t *ldg, const int32_t *n, double *const *work,
                         double *const *evals, double *const *Vs, int32_t *const *info);
// doubles of the `work` buffer launch_posterior_solve / launch_posterior_logpdf need (used when r > 128)
int64_t posterior_work_doubles(int32_t rp);
// Binv = (S/eps + I)^-1  (work: [rp*rp]); *err_flag != 0 on failure
// (work: binv_work_doubles(rp) doubles)
int64_t binv_work_doubles(int32_t rp);
void launch_binv(gingr_ctx *ctx, int32_t r, int32_t rp, const double *S, double *work, double *Binv, int32_t *err_flag);
// out = Binv (p/eps)
void launch_coeff_solve(gingr_ctx *ctx, int32_t r, int32_t rp, const double *Binv, const double *p, double *out);
void launch_state_init(gingr_ctx *ctx, DevState *st, const gingr_state_scalars *host_scalars_dev, double *zero_slot = nullptr);

// the nine moment blocks S[d][e] = sum_i q_{3i+d} q_{3i+e}^T into mom (MomentLayout); ws: moment_grams_ws_doubles(M, rp)
int64_t moment_grams_ws_doubles(int64_t M, int32_t rp);
void launch_moment_grams(gingr_ctx *ctx, const double *Q0, int64_t M, int32_t rp, double *ws, double *mom);
// p~ planes = ref + mean - c0
void launch_centered_mean(gingr_ctx *ctx, const gingr_model *m, double *ptil);

// Everything after the posterior solve in ONE workgroup: alpha_1, step blend, Umeyama from moments, second projection,
// alpha', state commit / failure status, sigma2 update (GingrAlgorithm.scala:212-246).
struct PostSolveArgs {
    int32_t r, rp;
    const double *pvec;     // gingr_model::pvec (PostVec)
    const double *zbuf;     // [rp][28] from launch_post_matvecs (PostVec::kZa ...)
    double *alpha;          // in/out: shape coefficients of the state
    const double *scalars;  // reduced {Np, xPx, trPXY, yPy, ...} (CPD) or nullptr
    int32_t is_icp;
    double icp_step, icp_end;
    double step;
    int32_t global_transform;
    DevState *state;
    double *zero_slot;      // nullable: cleared by thread 0 (SweepArgs::absmax_slot of the fit pass that follows)
    int32_t *retry;         // retryCounter of the algorithm instance (GingrAlgorithm.scala:69-70), device word; nullable
    int32_t probabilistic;  // update(current, probabilistic = true)
    double stop_threshold;  // >= 0: mark the state as stopped when this update moved sigma2 by less (CPD.scala:108-110); < 0: no rule
};
#define GINGR_RETRY_INIT 10 /* retryCounterInitialize, GingrAlgorithm.scala:69 */
void launch_post_solve(gingr_ctx *ctx, const PostSolveArgs &a);
// the 28 independent r x r mat-vecs the post-solve kernel consumes (they only depend on alpha and a); zbuf: [rp][PostVec::kZRows]
void launch_post_matvecs(gingr_ctx *ctx, const gingr_model *m, const double *alpha, const double *a, double *zbuf);
// out = scale * A B (r x r, leading dimension rp); one-off products at model finalisation
void launch_small_gemm(gingr_ctx *ctx, int32_t r, int32_t rp, const double *A, const double *B, double scale, double *out);
// the vector part of the model's PostVec block from the 12 moment vectors V[d][e], W[d] ([12][rp]); one-off at model finalisation
void launch_postvec(gingr_ctx *ctx, int32_t r, int32_t rp, const double *Binv, const double *moment_vectors, double *pvec);

// basis packing: stage is column-major [r][3M] (local rows, ORIGINAL order), out Q0 [3M][rp] in device (perm) order
void launch_pack_basis(gingr_ctx *ctx, const double *stage_colmajor, const double *variance_dev, int64_t M, int32_t r,
                       int32_t rp, const int32_t *perm, double *Q0);

// Model upload and the device-resident fitter: the host-side orchestration of one GiNGR update as a fixed sequence
// of kernels with no host synchronisation (C ABI in include/gingr_hip.h).
//
// One update = GingrAlgorithm.update (G/api/GingrAlgorithm.scala:192-254) followed by GingrGeneratorWrapper.propose's
// fit refresh and iteration++ (G/api/sampling/generators/GingrGeneratorWrapper.scala:28-39), split into three phases
// whose boundaries are exactly the points where a row-sharded run exchanges partial sums:
//   0  CPD column sums of K over the local rows (ICP: nearest neighbour, nothing to exchange)        -> segment 0
//   1  den, row statistics, observations, weighted Gram + right-hand side (+ landmarks), sigma2 sums -> segment 1
//   2  replicated O(r^2) algebra: posterior solve, then (moment form, gp.h) alpha_1, step blend, Umeyama, second
//      projection, alpha', state commit or failure status; finally the new fit of the local rows (one pass over Q0)
#ifdef GINGR_MH_TRACE
#include <chrono>
#include <cstdio>
#endif
#include "gp.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <functional>

struct gingr_fitter {
    gingr_ctx *ctx = nullptr;
    const gingr_model *m = nullptr;
    int64_t N = 0;
    double *target = nullptr;  // SoA [3][N]
    double *inv_den = nullptr, *Pt1 = nullptr;
    double *fit = nullptr;  // SoA [3][M]
    double *P1 = nullptr, *PX = nullptr;
    int32_t *nn_idx = nullptr;
    double *nn_d2 = nullptr;
    double *weight = nullptr, *evec = nullptr, *newshape = nullptr;
    double *alpha = nullptr, *acoef = nullptr, *alpha_c = nullptr, *zbuf = nullptr;
    double *zrand = nullptr;     // [rp] standard-normal draws of a probabilistic update (posterior.sample())
    bool zrand_active = false;
    // upload of the draws by the asynchronous entry points: a pinned buffer of its own (the synchronous entry points rewrite `pin`),
    // guarded by an event -- the buffer is rewritten only after the previous upload has left it, and nothing depends on when a copy
    // from pageable memory happens to consume its source
    double *zpin = nullptr;
    hipEvent_t zpin_done = nullptr;
    DevState *st = nullptr;
    DevPose *pose = nullptr;
    gingr_state_scalars *hs_dev = nullptr;
    // alpha [rp], hs_dev and st live in ONE allocation (state_block), in this order: set_state pushes [alpha | scalars] and get_state
    // pulls [alpha | scalars | DevState] in a single transfer each, through the pinned host buffer `pin` (no pageable staging)
    double *state_block = nullptr;
    double *pin = nullptr;
    size_t pin_doubles = 0;
    // the Metropolis-Hastings step's results straight into the pinned buffer (round 6): the device's view of `pin`, a counter of the
    // read-back kernel's finished workgroups, and the launch number the last of them stores into pin[pin_doubles - 1]
    double *pin_dev = nullptr;
    int32_t *mh_done = nullptr;
    uint64_t mh_epoch = 0;
    double *scalars = nullptr;  // local {Np, xPx, trPXY, yPy, -, c, -, -}
    double *part = nullptr;     // block partials of the scalar sums
    double *absmax = nullptr;   // [0] target, [1] fit: largest |coordinate| (exponent-argument range check)
    int32_t *tperm = nullptr;   // target cloud is kept in Morton order: tperm[s] = original target index of device position s
    std::vector<int32_t> h_tperm;
    double *tboxes = nullptr, *fboxes = nullptr;  // bounding boxes of the 256-point tiles of target / fit
    int32_t *tile_bad = nullptr;                  // target tiles holding a non-finite 1/den (never culled)
    double *xch = nullptr;
    int64_t off[GINGR_NUM_SEGMENTS] = {0, 0}, cnt[GINGR_NUM_SEGMENTS] = {0, 0};
    double *ws = nullptr;
    int64_t ws_doubles = 0;
    int colsum_chunks = 0;  // > 0: the column sums of phase 0 are still chunk partials in ws (single shard; added up by den_finalize)
    double *work = nullptr;
    void *aos = nullptr;  // staging for interleaved transfers, max(3M, 3N) doubles
    int32_t n_lm = 0;
    int32_t *lm_pid = nullptr;
    double *lm_xyz = nullptr, *lm_cov = nullptr;
    int32_t *lm_mask = nullptr;
    int32_t global_transform = GINGR_RIGID_TRANSFORMS;
    double step_length = 1.0;
    double stop_threshold = -1.0;  // gingr_fitter_set_stop_threshold: the run's stopping rule, applied by post_solve_kernel (< 0: none)
    int32_t stop_hit = 0;          // DevState::stopped as of the last gingr_fitter_get_state
    bool has_state = false;
    // ---- ICP surface correspondence (surface.hip): triangles in device vertex positions and a spatial triangle order
    bool icp_surface = false;                      // correspondence flavour of the ICP phases
    int32_t surface_method = 0;                    // 0 TriangularClosestPoint, 1 AlongNormalClosestPoint (ICP.scala:32-34)
    // reversed correspondence direction (ICP.scala:46-48): per TARGET vertex buffers, then one observation per model vertex
    bool reversed = false;
    // ... on a row shard the correspondence itself is replicated work (its queries are the replicated target) against the GATHERED
    // template, so the per-template-vertex arrays cover the whole template in ORIGINAL vertex order (set_meshes builds them); the
    // observations of this shard's rows are picked out afterwards (reversal_local_kernel)
    int32_t *radj_ptr = nullptr, *radj_tri = nullptr, *rmbnd = nullptr;
    double *rmvn = nullptr, *rfboxes = nullptr;
    void *rws = nullptr;
    // Round 5: the SCAN of the reversed direction is sharded too.  Its queries are the replicated target, so they partition by index
    // range -- this shard takes the device positions [rq0, rq0 + rqn) of the target, the fraction of the cloud that its rows are of
    // the template -- and every shard accumulates, per TEMPLATE vertex of the whole template (original ids), the sum of its accepted
    // target points and their number: revsum [4][M_total] = {sum x, sum y, sum z, count}.  One all-reduce (sum) of that buffer
    // (exchange segment GINGR_SEGMENT_REVSUM, between phases 0 and 1) gives every shard the totals; it keeps its own rows
    // (reversal_local_kernel).  partial_revsum: where the contribution goes when the sum lands elsewhere (device group).
    int64_t rq0 = 0, rqn = 0;
    double *rtvn_loc = nullptr;   // [3][rqn] vertex normals of the target's query range (the target is fixed: built once)
    double *revsum = nullptr, *partial_revsum = nullptr;
    // The nearest-template-VERTEX search of that direction runs against a spatially ordered copy of the gathered template: the
    // gathered fit is in original vertex order (the triangles index it), whose 256-vertex tiles are not compact, so the box-pruned
    // scan degenerated to all pairs (100 us for 5 121 queries x 40 962 vertices).  gperm (k-d leaf order of the first gathered fit,
    // fixed afterwards: a deforming template stays coherent) / gsorted [3][M_total]; matches are mapped back to original ids.
    int32_t *gperm = nullptr;
    double *gsorted = nullptr;
    int32_t *rnn_pos = nullptr;  // last search's matches as POSITIONS in gsorted: the warm start of the next one (queries and order are fixed)
    bool rnn_warm = false;
    // ... and the closest-point scan of that direction (target vertices against the MOVING template's triangles) starts every query
    // from the triangle that was closest to it last time (round 5; the forward direction has done so since round 2): positions in mtri
    int32_t *rtri_pos = nullptr;
    bool rtri_warm = false;
    int32_t *mtri_orig = nullptr, *mboundary = nullptr;
    double *rcp = nullptr, *rd2 = nullptr, *rnnd2 = nullptr, *rw01 = nullptr, *robs = nullptr, *rwin = nullptr;
    int32_t *rnn = nullptr, *rpre = nullptr, *rhit = nullptr, *rkeys = nullptr, *rvals = nullptr, *rskeys = nullptr, *rsvals = nullptr;
    void *rsort = nullptr;
    size_t rsort_bytes = 0;
    int64_t Tm = 0, Tt = 0;                        // model / target triangle counts
    int32_t *mtri = nullptr, *ttri = nullptr, *ttri_orig = nullptr;
    int32_t *madj_ptr = nullptr, *madj_tri = nullptr, *tadj_ptr = nullptr, *tadj_tri = nullptr;  // vertex -> triangles
    double *mcn = nullptr, *tcn = nullptr, *mvn = nullptr, *tvn = nullptr;  // cell / vertex normals (SoA)
    double *mtboxes = nullptr, *ttboxes = nullptr;  // triangle tile boxes [nt][6], quarter boxes [4 nt][6], group boxes [nt / 16 + 1][6] (line_nearest)
    double *mtribox = nullptr, *ttribox = nullptr;  // per-triangle boxes [T][6] (tri_tile_bbox_kernel), staged by the scan kernels
    int32_t *tboundary = nullptr;                   // target boundary vertices (device target positions)
    double *surf_cp = nullptr, *surf_d2 = nullptr, *surf_w01 = nullptr, *surf_win = nullptr, *surf_nnd2 = nullptr;
    int32_t *surf_nn = nullptr, *surf_pre = nullptr, *surf_hit = nullptr;
    // ---- memo of the posterior inputs (the reference keeps Memoize(computePosterior, 10), GingrAlgorithm.scala:68): phases 0 and
    // 1 (correspondences, Gram, right-hand side) depend only on (shape, pose, sigma2) of the state and on the flavour / its
    // parameters.  state_key describes the state last written by gingr_fitter_set_state while the device still holds it; post_key
    // the state whose phase-0/1 results sit in the exchange buffer.  A Metropolis-Hastings step asks for the posterior of the same
    // state up to three times (proposal, both transition densities); single shard only (a sharded run all-reduces the buffer).
    struct Key {
        std::vector<double> v;  // alpha[r], euler, center, translation, scale, sigma2
        int flavour = -1;       // 0 CPD, 1 ICP point cloud, 2 ICP surface; + method / direction bits
        double p0 = 0, p1 = 0;  // CPD: w, lambda
        bool same(const Key &o) const { return flavour == o.flavour && p0 == o.p0 && p1 == o.p1 && v == o.v; }
    };
    Key state_key, post_key;
    bool state_key_valid = false;
    int post_stage = 0;       // 0 nothing, 1 phase 0 done, 2 phases 0 and 1 done for post_key
    bool skip_phase1 = false;
    double *small = nullptr;  // 8 doubles of device scratch for scalar results
    void *stat_scratch = nullptr;  // StatScratch of gingr_fitter_surface_distance_stats (kept across calls)
    // Second memo slot and the factor cache of the transition-density query.  A Metropolis-Hastings step works on two states, the
    // current x and the candidate x' (proposal from x, q(x'|x), q(x|x')), and the next step starts from one of the two: `alt_seg`
    // keeps the [G, rhs, scalars] segment of the state the live memo held before (alt_key), and is swapped back in instead of
    // recomputing phases 0 and 1.  Only the probabilistic entry points use it (allow_alt): after a swap the correspondence arrays
    // on the device belong to the other state (corr_stale), which the getters of the deterministic path must never see.
    // fxbuf[live] / fxbuf[live ^ 1] go with the live / alt slot: [rp*rp] factor of S_tot + eps (I + G), [rp] posterior
    // coefficients, [rp] reciprocal diagonal -- what posterior_logpdf_lds_kernel leaves for posterior_logpdf_cached_kernel.
    double *alt_seg = nullptr;
    // The two memo slots exchange ROLES, not contents: seg_swapped = the live [G, rhs, scalars] segment is alt_seg and the parked one
    // sits in the exchange buffer.  Entry points that work on the exchange buffer itself (deterministic, sharded) move it back first.
    bool seg_swapped = false;
    double *seg1_live() const { return seg_swapped ? alt_seg : xch + off[1]; }
    double *seg1_parked() const { return seg_swapped ? xch + off[1] : alt_seg; }
    Key alt_key;
    int alt_stage = 0;
    bool allow_alt = false, corr_stale = false;
    double *fxbuf[2] = {nullptr, nullptr};
    unsigned *lp_sync = nullptr;  // hand-over words of posterior_logpdf_split_kernel
    double *lp_scratch = nullptr;  // [rp*rp + 2 rp], ranks >= 128 on a row shard: where the two-workgroup transition density leaves its
                                   // state-only part when no memo slot wants it (fitter_logpdf_finish; allocated on first use)
    unsigned lp_epoch = 0;
    bool fx_valid[2] = {false, false};
    // nfac[slot]: [rp*rp] Cholesky factor of I + G, [16*rp] the transposed inverses of its diagonal blocks -- left by the two-workgroup log-density kernel for the
    // sampled proposal that may start from this state (a + L^-T z without factoring again); a sits in fxbuf[slot] + rp*rp
    double *nfac[2] = {nullptr, nullptr};
    bool nf_valid[2] = {false, false};
    int live = 0;
    int32_t *surf_tri_pos = nullptr;  // per model vertex: position (in ttri) of its closest target triangle of the last scan
    bool surf_tri_warm = false;
    bool nn_warm = false, surf_nn_warm = false;  // nn_idx / surf_nn hold last time's matches against the CURRENT target
    NNGrid tgrid;  // uniform grid over the target cloud (set_target): the point-cloud ICP's closest-point search (nn_grid.hip)
    TriGrid ttgrid;  // uniform grid over the target TRIANGLES (set_meshes): the surface ICP's closest surface point (surface.hip)
    MovGrid mgrid;   // the same over the TEMPLATE's triangles, rebuilt on the device every iteration: the self-intersection test (round 5)
    void forget_posteriors() {
        post_stage = 0;
        alt_stage = 0;
        fx_valid[0] = fx_valid[1] = false;
        nf_valid[0] = nf_valid[1] = false;
    }
    // Where phases 0 / 1 put THIS shard's partial sums (same segment layout as xch).  nullptr: into xch itself (single shard, or a
    // host that all-reduces xch in place -- torch.distributed).  The device group (group.hip) points it at the shard's send
    // buffer: peers read that while the summed result lands in xch, so nobody overwrites what a peer may still be reading.
    double *partial_out = nullptr;
    // Row-sharded surface ICP: the tests against the template itself (vertex normals, self-intersection) need the WHOLE posed template,
    // so every iteration starts with a gather -- each shard contributes its rows of the fit to a [3][M_total] buffer in ORIGINAL
    // point order (zeros elsewhere), the buffers are summed across the shards (exchange segment 2: an all-gather spelled as the
    // all-reduce the other segments already use) -- and the template triangles index that buffer.  partial_fullfit: where the
    // contribution goes when the sum lands elsewhere (device group); nullptr = in place.
    double *fullfit = nullptr, *partial_fullfit = nullptr;
    // ... or, where the host has a real all-gather (RCCL: rccl_exchange.hip; gingr_fitter_gather_stage / _finish): the shard's rows go
    // into ITS slot of gstage [world][3][chunk] (original row order, chunk = ceil(M_total / world)), the slots are all-gathered in
    // place and one kernel spreads them over the planes of `fullfit` -- half the wire bytes of the zero-padded all-reduce, no sum
    int32_t *zero_counts = nullptr;  // [ceil(M / 256)] zero-weight vertices per block of the surface observations (gp.h: ZeroGate)
    double *gstage = nullptr;
    int gstage_world = 0;
    // what the ranks agreed on for this (model, meshes, world): -1 not asked yet, 0 the zero-padded all-reduce, 1 the all-gather
    // (rccl_exchange.hip: the choice between two different collectives must be the same on every rank)
    int gather_agreed = -1, gather_agreed_world = 0;
    bool sharded() const { return m->M != m->M_total; }
    int32_t *retry = nullptr;  // device word: retryCounter of the algorithm instance this fitter stands for (GingrAlgorithm.scala:69-70)
    // ---- one Metropolis-Hastings step per call (gingr_fitter_mh_step): the state x the step started from stays on the device --
    // [alpha | scalars | DevState] in mh_save, its fit in fit_alt (the proposal's fit is written to the OTHER buffer and the two
    // pointers are exchanged, no copy) -- so that a rejected proposal costs one small copy and one pass over the basis.  The states
    // a step produces on the device are unknown to the host until it reads them: the posterior memo keys them by a serial number.
    double *fit_alt = nullptr, *mh_save = nullptr;
    double *mh_rb = nullptr;  // [head + 8 + 3M]: what one step sends back, gathered for one copy
    Key mh_key;
    bool mh_saved = false;
    uint64_t mh_serial = 0;
    // The quarter boxes / |coordinate - centre| maximum of the fit are read by the two CPD pair loops only: the pass that writes the
    // fit produces them once a CPD phase has asked for them (cpd_seen), an ICP-only fitter runs the plain, shorter pass.
    bool cpd_seen = false, fit_boxes_valid = false;
    // GINGR_OPT_SPLIT_EXCHANGE (fitter_sharded_update): 0 the whole column-sum pass; 1 / 2 only the first / second half of the target
    // tiles (phase 0 is then run twice, with the all-reduce of the first half in between on the context's second stream)
    int split_half = 0;
};

// GINGR_OPT_GRAM_DOWNDATE by default: from this many local rows on.  The downdate pays while fewer than ~20 % of the rows have weight 0
// (41k: 15 + 17.5 us of downdate + right-hand-side sweep against 44 us of the weighted Gram pass at 0.2 % rejected; it costs one memory
// round trip per four zero-weight vertices of a slab); small meshes -- the femur chain rejects a fifth of its 1 622 vertices -- keep
// the pass over the basis, which is cheap there (13.6 us).  A fixed rule, not a measured one: both forms round differently.
constexpr int64_t kGramDowndateMinRows = 16384;
// The triangle grid pays from a few ten thousand target triangles on (41k x 82k: 94 -> 27 + 10 us per closest-point search); on a
// small mesh the tile scan with its sixteen query copies per workgroup is faster (femur, 3 240 triangles: 16 us against 15 + 7).
constexpr int64_t kTriGridMinTriangles = 16384;
constexpr size_t kScalarsDoubles = (sizeof(gingr_state_scalars) + 7) / 8, kDevStateDoubles = (sizeof(DevState) + 7) / 8;

namespace {

// a <-> b (exchange != 0) or a <- b: the [G, rhs, scalars] segments of the two posterior memo slots
__global__ __launch_bounds__(256) void swap_segments_kernel(double *__restrict__ a, double *__restrict__ b, int64_t n, int exchange) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double va = a[i], vb = b[i];
    a[i] = vb;
    if (exchange) b[i] = va;
}

// ---- one Metropolis-Hastings step: small transfers as kernels (round 5).  A blit copy on this stack costs 4-8 us on the device
// timeline with the barrier packets around it; the step had 4.7 of them.  (a) the proposal's draws / parameters travel in the
// kernel's ARGUMENT (<= 160 doubles) and the same launch parks the current state; (b) the state block, the eight results and the
// fit (original order, interleaved) are gathered into ONE buffer for ONE device-to-host copy.
constexpr int kMhPayload = 160;
struct MhPayload {
    double v[kMhPayload];
};
// save[0..m) = src[0..m), THEN dst[0..n) = payload (dst may be src: the random-walk parameters overwrite the state block that was just parked)
// st != nullptr: the payload was [alpha | scalars] of a random-walk proposal -- the device state is initialised from it in the same launch
// (state_init_kernel's work: one launch less per such step)
__global__ __launch_bounds__(256) void mh_begin_kernel(MhPayload payload, int n, double *dst, const double *src, int m, double *__restrict__ save,
                                                       DevState *st, const gingr_state_scalars *hs, double *zero_slot) {
    const int t = threadIdx.x;
    double keep = 0.0;
    if (t < m) keep = src[t];
    __syncthreads();
    if (t < m) save[t] = keep;
    if (t < n) dst[t] = payload.v[t];
    if (st) {
        __syncthreads();  // (the scalars just written by this workgroup are read back by its thread 0)
        if (t == 0) state_init_body(st, hs, zero_slot);
    }
}
// out[0..nblock) = block; out[nblock + 3 perm[i] + d] = fit[d][i].  With `flag`: out is HOST memory (the fitter's pinned buffer through
// its device address); the last workgroup to finish stores `epoch` into *flag behind a system-scope fence, and the host, spinning on
// that word, has the results without a copy launch and without the wake-up of a stream synchronisation (~12 us of a 235 us step).
__global__ __launch_bounds__(256) void mh_readback_kernel(const double *__restrict__ block, int nblock, const double *__restrict__ fit, int64_t M,
                                                         const int32_t *__restrict__ perm, double *__restrict__ out, double *flag,
                                                         int32_t *__restrict__ done, double epoch) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nblock) out[i] = block[i];
    if (fit && i < M) {
        const int64_t o = perm ? perm[i] : i;
        double *dst = out + nblock + 3 * o;
        dst[0] = fit[i];
        dst[1] = fit[M + i];
        dst[2] = fit[2 * M + i];
    }
    if (!flag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
            *done = 0;  // (the next launch is behind this one in the stream)
            __threadfence_system();
            *reinterpret_cast<volatile double *>(flag) = epoch;
        }
    }
}

// full[d][g] = this shard's fit of original point g (device position iperm[g - row_begin]) or 0 for the points of other shards
__global__ __launch_bounds__(256) void fit_contribution_kernel(const double *__restrict__ fit, const int32_t *__restrict__ iperm, int64_t M,
                                                               int64_t row_begin, int64_t M_total, double *__restrict__ full) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= M_total) return;
    const int64_t l = g - row_begin;
    const bool mine = l >= 0 && l < M;
    const int64_t pos = mine ? iperm[l] : 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) full[d * M_total + g] = mine ? fit[d * M + pos] : 0.0;
}

// stage[d][l] = this shard's fit of its l-th row in ORIGINAL order (device position iperm[l]); l < M
__global__ __launch_bounds__(256) void fit_to_stage_kernel(const double *__restrict__ fit, const int32_t *__restrict__ iperm, int64_t M, int64_t chunk,
                                                           double *__restrict__ stage) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= M) return;
    const int64_t pos = iperm[l];
#pragma unroll
    for (int d = 0; d < 3; ++d) stage[d * chunk + l] = fit[d * M + pos];
}
// full[d][g] = stage[q][d][g - begin(q)], q = the shard that owns row g under the balanced contiguous partition of M_total rows over
// `world` shards (the first M_total % world shards hold one row more)
__global__ __launch_bounds__(256) void stage_to_fullfit_kernel(const double *__restrict__ stage, int world, int64_t chunk, int64_t M_total,
                                                               double *__restrict__ full) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= M_total) return;
    const int64_t base = M_total / world, extra = M_total % world;
    const int64_t cut = extra * (base + 1);  // rows below `cut` sit in shards of base + 1 rows
    const int64_t q = g < cut ? g / (base + 1) : extra + (g - cut) / (base > 0 ? base : 1);
    const int64_t b = q * base + (q < extra ? q : extra);
    const double *p = stage + q * 3 * chunk + (g - b);
#pragma unroll
    for (int d = 0; d < 3; ++d) full[d * M_total + g] = p[d * chunk];
}

// the observations of this shard's rows out of the per-template-vertex arrays of the whole template (original vertex order): device
// position p of the shard holds original vertex row_begin + perm[p]
// (sums: [4][M_total] = {sum x, sum y, sum z, count} of the accepted target points per template vertex, summed over all shards'
// query ranges: the observation of a vertex is their mean, its weight count / sigma2 -- k isotropic observations of one point)
__global__ __launch_bounds__(256) void reversal_local_kernel(int64_t M, int64_t row_begin, const int32_t *__restrict__ perm, int64_t M_total,
                                                             const double *__restrict__ sums, const double *__restrict__ sigma2,
                                                             double *__restrict__ obs, double *__restrict__ win) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= M) return;
    const int64_t g = row_begin + perm[p];
    const double k = sums[3 * M_total + g];
    const double kk = k > 0.0 ? k : 1.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) obs[d * M + p] = sums[d * M_total + g] / kk;
    win[p] = k / sigma2[0];
}

// dst[d][p] = src[d][perm[p]] (SoA planes of n points)
__global__ __launch_bounds__(256) void soa_permute_kernel(const double *__restrict__ src, const int32_t *__restrict__ perm, int64_t n,
                                                          double *__restrict__ dst) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int64_t g = perm[p];
#pragma unroll
    for (int d = 0; d < 3; ++d) dst[d * n + p] = src[d * n + g];
}
// idx[j] = map[pos[j]] (positions in the spatially ordered template -> original vertex ids); negative entries stay
__global__ __launch_bounds__(256) void index_map_kernel(const int32_t *__restrict__ pos, int64_t n, const int32_t *__restrict__ map,
                                                        int32_t *__restrict__ idx) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int32_t v = pos[j];
    idx[j] = v >= 0 ? map[v] : v;
}

// dst[d][i] = src[d][q0 + i]: a compact copy of the planes of an SoA array for the index range [q0, q0 + n)
__global__ __launch_bounds__(256) void soa_range_kernel(const double *__restrict__ src, int64_t stride, int64_t q0, int64_t n, double *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) dst[d * n + i] = src[d * stride + q0 + i];
}

template <typename T>
int dev_alloc(gingr_ctx *ctx, T **p, size_t count) {
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T)));
    // diagnostic (GINGR_DEBUG_POISON=1): new buffers start as NaN / 0xFFFFFFFF instead of whatever the allocator hands out, so that a
    // read of something never written shows instead of depending on what ran before
    static const bool poison = getenv("GINGR_DEBUG_POISON") != nullptr;
    if (poison) {  // (memsets of device memory are asynchronous to the host and not ordered with the context's non-blocking stream)
        HIP_TRY(ctx, hipMemset(*p, 0xFF, (count ? count : 1) * sizeof(T)));
        HIP_TRY(ctx, hipDeviceSynchronize());
    }
    return GINGR_OK;
}

void dev_free(void *p) {
    if (p) (void)hipFree(p);
}

int check_launch(gingr_ctx *ctx) {
    HIP_TRY(ctx, hipGetLastError());
    return GINGR_OK;
}

// pose <- rigid part of the state (scale 1): the frame a mesh is projected in by the transition-density query
__global__ void pose_of_state_kernel(const DevState *__restrict__ st, DevPose *__restrict__ pose) {
    const int t = threadIdx.x;
    if (t < 9) pose->R[t] = st->R[t];
    if (t < 3) {
        pose->euler[t] = st->euler[t];
        pose->t[t] = st->t[t];
        pose->center[t] = st->center[t];
    }
    if (t == 0) pose->scale = 1.0;
}

Cloud cloud_of(const double *soa, int64_t n) { return Cloud{soa, soa + n, soa + 2 * n, n}; }

// where GINGR_OPT_SPLIT_EXCHANGE cuts the target cloud: half of its 256-point tiles
int64_t split_cut(int64_t N) { return (N / 512) * 256; }

SweepArgs base_args(const gingr_fitter *f) {
    SweepArgs a;
    memset(&a, 0, sizeof(a));
    a.Q0 = f->m->Q0;
    a.ref = f->m->ref;
    a.mean = f->m->mean;
    a.M = f->m->M;
    a.rp = f->m->rp;
    a.state = f->st;
    a.pose = f->pose;
    a.c0[0] = f->m->c0[0];
    a.c0[1] = f->m->c0[1];
    a.c0[2] = f->m->c0[2];
    a.partial = f->ws;
    return a;
}

// boxes + |coordinate - centre| maximum of a fit that was NOT written by refresh_fit (explicit fit points; a target set after the state)
void fit_boxes_now(gingr_fitter *f) {
    if (!f->fboxes) return;
    (void)hipMemsetAsync(f->absmax + 1, 0, sizeof(double), f->ctx->stream);
    launch_tile_bbox(f->ctx, cloud_of(f->fit, f->m->M), f->fboxes, f->absmax + 2, f->absmax + 1);
    f->fit_boxes_valid = true;
}

// fit = modelInstanceShapePoseScale(model, state)
void refresh_fit(gingr_fitter *f) {
    SweepArgs a = base_args(f);
    a.coef0 = f->alpha;
    a.shape_out = f->fit;
    if (f->fboxes && f->m->rp <= 512 && f->cpd_seen) {  // the pass also leaves the quarter boxes and the |coordinate - centre| maximum
        a.qboxes = f->fboxes + 6 * ceil_div(f->m->M, 256);
        a.box_centre = f->absmax + 2;
        a.absmax_slot = f->absmax + 1;
        launch_sweep(f->ctx, SWEEP_FIT, a);
        f->fit_boxes_valid = true;
    } else {
        launch_sweep(f->ctx, SWEEP_FIT, a);
        f->fit_boxes_valid = false;
        if (f->cpd_seen) fit_boxes_now(f);  // rank > 512: the generic pass, the boxes by a launch of their own
    }
}

void free_meshes(gingr_fitter *f) {
    void *rptrs[] = {f->mtri_orig, f->mboundary, f->rcp, f->rd2, f->rnnd2, f->rw01, f->robs, f->rwin, f->rnn, f->rpre, f->rhit,
                     f->rkeys, f->rvals, f->rskeys, f->rsvals, f->rsort, f->radj_ptr, f->radj_tri, f->rmbnd, f->rmvn, f->rfboxes,
                     f->rtvn_loc, f->revsum, f->rws, f->gperm, f->gsorted, f->rnn_pos, f->rtri_pos};
    for (void *p : rptrs) dev_free(p);
    f->radj_ptr = f->radj_tri = f->rmbnd = nullptr;
    f->rmvn = f->rfboxes = f->rtvn_loc = f->revsum = f->gsorted = nullptr;
    f->gperm = f->rnn_pos = f->rtri_pos = nullptr;
    f->rnn_warm = f->rtri_warm = false;
    f->gather_agreed = -1;  // (new meshes: the ranks agree again)
    f->rq0 = f->rqn = 0;
    f->rws = nullptr;
    f->mtri_orig = f->mboundary = f->rnn = f->rpre = f->rhit = f->rkeys = f->rvals = f->rskeys = f->rsvals = nullptr;
    f->rcp = f->rd2 = f->rnnd2 = f->rw01 = f->robs = f->rwin = nullptr;
    f->rsort = nullptr;
    f->rsort_bytes = 0;
    f->reversed = false;
    void *ptrs[] = {f->mtri, f->ttri, f->ttri_orig, f->madj_ptr, f->madj_tri, f->tadj_ptr, f->tadj_tri, f->mcn, f->tcn, f->mvn,
                    f->tvn, f->mtboxes, f->ttboxes, f->tboundary, f->surf_cp, f->surf_d2, f->surf_w01, f->surf_win, f->surf_nnd2,
                    f->surf_nn, f->surf_pre, f->surf_hit, f->surf_tri_pos, f->mtribox, f->ttribox};
    for (void *p : ptrs) dev_free(p);
    f->mtri = f->ttri = f->ttri_orig = f->madj_ptr = f->madj_tri = f->tadj_ptr = f->tadj_tri = f->tboundary = nullptr;
    f->mcn = f->tcn = f->mvn = f->tvn = f->mtboxes = f->ttboxes = f->mtribox = f->ttribox = nullptr;
    f->surf_cp = f->surf_d2 = f->surf_w01 = f->surf_win = f->surf_nnd2 = nullptr;
    f->surf_nn = f->surf_pre = f->surf_hit = f->surf_tri_pos = nullptr;
    f->surf_tri_warm = f->surf_nn_warm = false;
    tri_grid_free(&f->ttgrid);
    mov_grid_free(&f->mgrid);
    f->Tm = f->Tt = 0;
}

int model_finalize_impl(gingr_ctx *ctx, gingr_model *m) {
    DevBuf work, flag;
    HIP_TRY(ctx, work.alloc((size_t)binv_work_doubles(m->rp) * sizeof(double)));
    HIP_TRY(ctx, flag.alloc(sizeof(int32_t)));
    launch_binv(ctx, m->r, m->rp, m->mom, work.as<double>(), m->Binv, flag.as<int32_t>());
    GINGR_TRY(check_launch(ctx));
    int32_t err = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&err, flag.p, sizeof(err), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (err) return gingr_set_error(ctx, GINGR_ERR_NOT_SPD, "model finalize: Q^T Q / 1e-5 + I is not positive definite");
    // constant products of the moment form (gp.h: cmat, PostVec): C = Binv S_tot / eps, Binv S[d][e], Binv S[d][e] C (work = S[d][e] C)
    const MomentLayout ml{m->rp};
    const int64_t rr = (int64_t)m->rp * m->rp;
    launch_small_gemm(ctx, m->r, m->rp, m->Binv, m->mom + ml.stot(), 1.0 / GINGR_COEFF_NOISE, m->cmat);
    for (int d = 0; d < 3; ++d)
        for (int e = 0; e < 3; ++e) {
            launch_small_gemm(ctx, m->r, m->rp, m->Binv, m->mom + ml.S(d, e), 1.0, m->cmat + (1 + d * 3 + e) * rr);
            launch_small_gemm(ctx, m->r, m->rp, m->mom + ml.S(d, e), m->cmat, 1.0, work.as<double>());
            launch_small_gemm(ctx, m->r, m->rp, m->Binv, work.as<double>(), 1.0, m->cmat + (10 + d * 3 + e) * rr);
        }
    // the moment vectors V[d][e], W[d] (contiguous in mom from V(0, 0) on) and Binv times them; then the scalars of the full model
    const PostVec pvl{m->rp};
    launch_postvec(ctx, m->r, m->rp, m->Binv, m->mom + ml.V(0, 0), m->pvec);
    {
        double cst[16];
        for (int q = 0; q < 9; ++q) cst[q] = m->Pp[q];
        for (int q = 0; q < 3; ++q) {
            cst[9 + q] = m->Ps[q];
            cst[12 + q] = m->c0[q];
        }
        cst[15] = (double)m->M_total;
        HIP_TRY(ctx, hipMemcpyAsync(m->pvec + pvl.consts(), cst, sizeof(cst), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // cst leaves scope
    }
    GINGR_TRY(check_launch(ctx));
    // The eigen-decomposition S_tot = V diag(lam) V^T for the uniform-weight posterior (point-cloud ICP without landmarks: the
    // posterior (I + S_tot / sigma2)^-1 rhs is two mat-vecs then).  Decided HERE, once, outside every asynchronous update: S_tot is
    // the all-reduced moment, bit-identical on every shard, and the decomposition is deterministic, so all shards of a sharded model
    // take the same path.  Up to the 192 columns of the register kernel (eig.hip: 0.24 ms at rank 100, 3.5 ms at 192); above that the
    // two-sided kernel would take tens of ms of every model's set-up, more than the Cholesky path costs an ICP run (0.1 ms an iteration).
    m->eig_ready = false;
    if (m->r <= kSymEigColsMaxN) {
        if (launch_jacobi_eig(ctx, m->mom + ml.stot(), m->rp, m->r, m->eigL, m->eigV) == GINGR_OK)
            m->eig_ready = true;
        else
            (void)hipGetLastError();
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    m->finalized = true;
    return GINGR_OK;
}

}  // namespace

// ===================================================================================================== model

// Shared by gingr_model_upload (basis from the host) and the on-device GPMM builder (gpmm.hip): everything of a model
// except how Q0 = U sqrt(lambda) gets filled.  fill_basis runs after the row permutation exists and must write all of
// m->Q0 ([3M][rp], device row order, zero padded) on ctx->stream.
int model_create_impl(gingr_ctx *ctx, int64_t M_total, int32_t rank, const double *ref, const double *mean,
                      const double *variance, int64_t row_begin, int64_t row_end,
                      const std::function<int(gingr_model *)> &fill_basis, gingr_model **out) {
    if (!ctx || !out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (M_total < 1 || rank < 1 || rank > 512 || !ref || !mean || !variance)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "model_upload: need M >= 1 and 1 <= rank <= 512");
    if (row_begin < 0 || row_end > M_total || row_begin >= row_end)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "model_upload: bad row shard [%lld,%lld)", (long long)row_begin,
                               (long long)row_end);
    for (int32_t k = 0; k < rank; ++k)
        if (!(variance[k] >= 0.0)) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "model_upload: variance[%d] < 0", k);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gingr_model *m = new gingr_model();
    m->ctx = ctx;
    m->M_total = M_total;
    m->row_begin = row_begin;
    m->row_end = row_end;
    m->M = row_end - row_begin;
    m->r = rank;
    m->rp = (int32_t)round_up(rank, 16);
    m->variance.assign(variance, variance + rank);
    const int64_t M = m->M;
    // centroid of the full reference (identical on every shard)
    double c[3] = {0, 0, 0};
    for (int64_t i = 0; i < M_total; ++i) {
        c[0] += ref[3 * i];
        c[1] += ref[3 * i + 1];
        c[2] += ref[3 * i + 2];
    }
    for (int d = 0; d < 3; ++d) m->c0[d] = c[d] / (double)M_total;

    int rc = GINGR_OK;
    DevBuf aos;
    auto fail = [&](int code) {
        gingr_model_destroy(m);
        return code;
    };
    if ((rc = dev_alloc(ctx, &m->Q0, (size_t)(3 * M + kBasisRowSlack) * m->rp)) || (rc = dev_alloc(ctx, &m->ref, (size_t)3 * M)) ||
        (rc = dev_alloc(ctx, &m->mean, (size_t)3 * M)) || (rc = dev_alloc(ctx, &m->mom, (size_t)MomentLayout{m->rp}.total())) ||
        (rc = dev_alloc(ctx, &m->Binv, (size_t)m->rp * m->rp)) || (rc = dev_alloc(ctx, &m->eigV, (size_t)m->r * m->r)) ||
        (rc = dev_alloc(ctx, &m->eigL, (size_t)m->r)) ||
        (rc = dev_alloc(ctx, &m->cmat, (size_t)19 * m->rp * m->rp)) ||
        (rc = dev_alloc(ctx, &m->pvec, (size_t)PostVec{m->rp}.total())))
        return fail(rc);
    if (aos.alloc((size_t)3 * M * sizeof(double)) != hipSuccess)
        return fail(gingr_set_error(ctx, GINGR_ERR_HIP, "model_upload: out of device memory"));
    // device row order = Morton order of the local mean shape
    {
        std::vector<double> pts((size_t)3 * M);
        for (int64_t i = 0; i < 3 * M; ++i) pts[(size_t)i] = ref[3 * row_begin + i] + mean[3 * row_begin + i];
        morton_order(pts.data(), M, m->hperm);
        m->hiperm.resize((size_t)M);
        for (int64_t sidx = 0; sidx < M; ++sidx) m->hiperm[(size_t)m->hperm[(size_t)sidx]] = (int32_t)sidx;
        if ((rc = dev_alloc(ctx, &m->perm, (size_t)M)) || (rc = dev_alloc(ctx, &m->iperm, (size_t)M))) return fail(rc);
        if (hipMemcpy(m->perm, m->hperm.data(), (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(m->iperm, m->hiperm.data(), (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
            return fail(gingr_set_error(ctx, GINGR_ERR_HIP, "model_upload: permutation copy failed"));
    }
    (void)hipMemsetAsync(m->Q0 + (size_t)3 * M * m->rp, 0, (size_t)kBasisRowSlack * m->rp * sizeof(double), ctx->stream);
    if ((rc = fill_basis(m))) return fail(rc);
    (void)hipMemcpyAsync(aos.p, ref + 3 * row_begin, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    launch_aos_to_soa(ctx, aos.as<double>(), M, m->ref, m->perm);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpyAsync(aos.p, mean + 3 * row_begin, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    launch_aos_to_soa(ctx, aos.as<double>(), M, m->mean, m->perm);
    // one-off moments of the local rows (MomentLayout): S_tot, S[d][e], V[d][e], W[d]
    {
        const MomentLayout ml{m->rp};
        DevBuf gws, sws, ptil, ev;
        if (gws.alloc((size_t)std::max(gram_ws_doubles(M, m->rp), moment_grams_ws_doubles(M, m->rp)) * sizeof(double)) != hipSuccess ||
            sws.alloc((size_t)sweep_ws_doubles(M, m->rp) * sizeof(double)) != hipSuccess ||
            ptil.alloc((size_t)3 * M * sizeof(double)) != hipSuccess || ev.alloc((size_t)3 * M * sizeof(double)) != hipSuccess)
            return fail(gingr_set_error(ctx, GINGR_ERR_HIP, "model_upload: out of device memory"));
        launch_gram(ctx, m->Q0, M, m->rp, nullptr, gws.as<double>(), m->mom + ml.stot());
        launch_moment_grams(ctx, m->Q0, M, m->rp, gws.as<double>(), m->mom);
        launch_centered_mean(ctx, m, ptil.as<double>());
        std::vector<double> ones((size_t)M, 1.0);
        DevBuf dones;
        if (dones.alloc((size_t)M * sizeof(double)) != hipSuccess)
            return fail(gingr_set_error(ctx, GINGR_ERR_HIP, "model_upload: out of device memory"));
        (void)hipMemcpyAsync(dones.p, ones.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        SweepArgs a;
        memset(&a, 0, sizeof(a));
        a.Q0 = m->Q0;
        a.ref = m->ref;
        a.mean = m->mean;
        a.M = M;
        a.rp = m->rp;
        a.evec = ev.as<double>();
        a.partial = sws.as<double>();
        for (int d = 0; d < 3; ++d)
            for (int e = 0; e <= 3; ++e) {  // e == 3: the all-ones plane gives W[d]
                (void)hipMemsetAsync(ev.p, 0, (size_t)3 * M * sizeof(double), ctx->stream);
                const double *src = e < 3 ? ptil.as<double>() + (size_t)e * M : dones.as<double>();
                (void)hipMemcpyAsync(ev.as<double>() + (size_t)d * M, src, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice,
                                     ctx->stream);
                a.out = m->mom + (e < 3 ? ml.V(d, e) : ml.W(d));
                launch_sweep(ctx, SWEEP_RHS, a);
            }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            return fail(gingr_set_error(ctx, GINGR_ERR_HIP, "model_upload: kernel launch failed"));
    }
    // host moments of p~ over the FULL model (identical on every shard)
    for (int q = 0; q < 9; ++q) m->Pp[q] = 0.0;
    for (int q = 0; q < 3; ++q) m->Ps[q] = 0.0;
    if (M != M_total) {  // a shard keeps the mean shape of the whole model on the host (gingr_fitter_set_meshes: triangle order)
        m->h_full_pts.resize((size_t)3 * M_total);
        for (int64_t i = 0; i < 3 * M_total; ++i) m->h_full_pts[(size_t)i] = ref[i] + mean[i];
    }
    for (int64_t i = 0; i < M_total; ++i) {
        double pt[3];
        for (int d = 0; d < 3; ++d) pt[d] = ref[3 * i + d] + mean[3 * i + d] - m->c0[d];
        for (int d = 0; d < 3; ++d) {
            m->Ps[d] += pt[d];
            for (int e = 0; e < 3; ++e) m->Pp[d * 3 + e] += pt[d] * pt[e];
        }
    }
    if (row_begin == 0 && row_end == M_total) {
        rc = model_finalize_impl(ctx, m);
        if (rc) return fail(rc);
    }
    *out = m;
    return GINGR_OK;
}

static void free_stat_scratch(void *p);  // defined next to StatScratch (surface distance statistics)

extern "C" {

int gingr_model_upload(gingr_ctx *ctx, int64_t M_total, int32_t rank, const double *ref, const double *mean,
                       const double *basis_colmajor, const double *variance, int64_t row_begin, int64_t row_end,
                       gingr_model **out) {
    if (!ctx || !out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (!basis_colmajor) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "model_upload: basis is null");
    DevBuf stage, var;
    auto fill = [&](gingr_model *m) -> int {
        const int64_t M = m->M;
        if (stage.alloc((size_t)3 * M * rank * sizeof(double)) != hipSuccess || var.alloc(rank * sizeof(double)) != hipSuccess)
            return gingr_set_error(ctx, GINGR_ERR_HIP, "model_upload: out of device memory");
        // basis: column k of the shard = rows [3*row_begin, 3*row_end) of host column k
        if (hipMemcpy2DAsync(stage.p, (size_t)3 * M * sizeof(double), basis_colmajor + 3 * row_begin,
                             (size_t)3 * M_total * sizeof(double), (size_t)3 * M * sizeof(double), (size_t)rank,
                             hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return gingr_set_error(ctx, GINGR_ERR_HIP, "model_upload: basis copy failed");
        (void)hipMemcpyAsync(var.p, variance, rank * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        launch_pack_basis(ctx, stage.as<double>(), var.as<double>(), M, rank, m->rp, m->perm, m->Q0);
        return GINGR_OK;
    };
    return model_create_impl(ctx, M_total, rank, ref, mean, variance, row_begin, row_end, fill, out);
}

void gingr_model_destroy(gingr_model *m) {
    if (!m) return;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    dev_free(m->Q0);
    dev_free(m->ref);
    dev_free(m->mean);
    dev_free(m->mom);
    dev_free(m->Binv);
    dev_free(m->eigV);
    dev_free(m->eigL);
    dev_free(m->cmat);
    dev_free(m->pvec);
    dev_free(m->perm);
    dev_free(m->iperm);
    delete m;
}

int64_t gingr_model_num_points(const gingr_model *m) { return m ? m->M : 0; }
int32_t gingr_model_rank(const gingr_model *m) { return m ? m->r : 0; }

int gingr_model_gram_exchange(gingr_model *m, void **dev_ptr, int64_t *count) {
    if (!m || !dev_ptr || !count) return GINGR_ERR_BAD_ARGUMENT;
    *dev_ptr = m->mom;
    *count = MomentLayout{m->rp}.total();
    return GINGR_OK;
}

int gingr_model_finalize(gingr_ctx *ctx, gingr_model *m) {
    if (!ctx || !m) return GINGR_ERR_BAD_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return model_finalize_impl(ctx, m);
}

// ===================================================================================================== fitter
int gingr_fitter_create(gingr_ctx *ctx, const gingr_model *model, gingr_fitter **out) {
    if (!ctx || !model || !out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (!model->finalized)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "fitter_create: sharded model not finalized (gingr_model_finalize)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gingr_fitter *f = new gingr_fitter();
    f->ctx = ctx;
    f->m = model;
    const int64_t M = model->M;
    const int32_t rp = model->rp;
    int rc;
    if ((rc = dev_alloc(ctx, &f->fit, (size_t)3 * M)) || (rc = dev_alloc(ctx, &f->P1, (size_t)M)) ||
        (rc = dev_alloc(ctx, &f->PX, (size_t)3 * M)) || (rc = dev_alloc(ctx, &f->nn_idx, (size_t)M)) ||
        (rc = dev_alloc(ctx, &f->nn_d2, (size_t)M)) || (rc = dev_alloc(ctx, &f->weight, (size_t)M)) ||
        (rc = dev_alloc(ctx, &f->zero_counts, (size_t)ceil_div(M, 256))) ||
        (rc = dev_alloc(ctx, &f->evec, (size_t)3 * M)) || (rc = dev_alloc(ctx, &f->newshape, (size_t)3 * M)) ||
        (rc = dev_alloc(ctx, &f->state_block, (size_t)rp + kScalarsDoubles + kDevStateDoubles + 8)) || (rc = dev_alloc(ctx, &f->acoef, (size_t)rp)) ||
        (rc = dev_alloc(ctx, &f->fxbuf[0], (size_t)rp * rp + 2 * rp)) || (rc = dev_alloc(ctx, &f->fxbuf[1], (size_t)rp * rp + 2 * rp)) ||
        (rc = dev_alloc(ctx, &f->nfac[0], (size_t)(rp + 16) * rp)) || (rc = dev_alloc(ctx, &f->nfac[1], (size_t)(rp + 16) * rp)) ||
        (rc = dev_alloc(ctx, &f->alt_seg, (size_t)rp * rp + 2 * rp + 8)) || (rc = dev_alloc(ctx, &f->lp_sync, (size_t)2)) ||
        (rc = dev_alloc(ctx, &f->alpha_c, (size_t)rp)) || (rc = dev_alloc(ctx, &f->zbuf, (size_t)PostVec::kZRows * rp)) || (rc = dev_alloc(ctx, &f->zrand, (size_t)rp)) ||
        (rc = dev_alloc(ctx, &f->pose, 1)) || (rc = dev_alloc(ctx, &f->fit_alt, (size_t)3 * M)) ||
        (rc = dev_alloc(ctx, &f->mh_save, (size_t)rp + kScalarsDoubles + kDevStateDoubles)) ||
        (rc = dev_alloc(ctx, &f->scalars, 8)) || (rc = dev_alloc(ctx, &f->part, GINGR_SCALAR_PART)) || (rc = dev_alloc(ctx, &f->absmax, GINGR_AUX)) || (rc = dev_alloc(ctx, &f->work, (size_t)std::max<int64_t>((int64_t)rp * rp, posterior_work_doubles(rp)))) ||
        (rc = dev_alloc(ctx, &f->lm_mask, (size_t)M))) {
        gingr_fitter_destroy(f);
        return rc;
    }
    f->alpha = f->state_block;
    f->hs_dev = reinterpret_cast<gingr_state_scalars *>(f->state_block + rp);
    f->st = reinterpret_cast<DevState *>(f->state_block + rp + kScalarsDoubles);
    f->small = f->state_block + rp + kScalarsDoubles + kDevStateDoubles;  // behind the state: one transfer brings both back (mh_step)
    f->pin_doubles = (size_t)3 * M + rp + kScalarsDoubles + kDevStateDoubles + 16 + 2;  // (+ the eight results of gingr_fitter_mh_step, + its flag)
    if (hipHostMalloc(reinterpret_cast<void **>(&f->pin), f->pin_doubles * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        f->pin = nullptr;
        gingr_fitter_destroy(f);
        return gingr_set_error(ctx, GINGR_ERR_HIP, "fitter_create: pinned host buffer");
    }
    if ((rc = dev_alloc(ctx, &f->retry, 1)) || (rc = dev_alloc(ctx, &f->mh_done, 1))) {
        gingr_fitter_destroy(f);
        return rc;
    }
    f->pin[f->pin_doubles - 1] = 0.0;
    if (hipMemset(f->mh_done, 0, sizeof(int32_t)) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&f->pin_dev), f->pin, 0) != hipSuccess) {
        (void)hipGetLastError();
        f->pin_dev = nullptr;  // (the step then copies its results back as before)
    }
    {
        const int32_t init = GINGR_RETRY_INIT;
        if (hipMemcpy(f->retry, &init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            gingr_fitter_destroy(f);
            return gingr_set_error(ctx, GINGR_ERR_HIP, "fitter_create: retry counter upload");
        }
    }
    (void)hipMemsetAsync(f->lm_mask, 0, (size_t)M * sizeof(int32_t), ctx->stream);
    (void)hipMemsetAsync(f->alpha, 0, (size_t)rp * sizeof(double), ctx->stream);
    (void)hipMemsetAsync(f->scalars, 0, 8 * sizeof(double), ctx->stream);
    *out = f;
    return GINGR_OK;
}

void gingr_fitter_destroy(gingr_fitter *f) {
    if (!f) return;
    if (f->ctx) {
        (void)hipSetDevice(f->ctx->device);
        (void)hipStreamSynchronize(f->ctx->stream);
    }
    free_stat_scratch(f->stat_scratch);
    dev_free(f->target);
    dev_free(f->inv_den);
    dev_free(f->Pt1);
    dev_free(f->fit);
    dev_free(f->fit_alt);
    dev_free(f->mh_save);
    dev_free(f->mh_rb);
    dev_free(f->mh_done);
    dev_free(f->P1);
    dev_free(f->PX);
    dev_free(f->nn_idx);
    dev_free(f->nn_d2);
    dev_free(f->weight);
    dev_free(f->evec);
    dev_free(f->newshape);
    dev_free(f->state_block);
    if (f->pin) (void)hipHostFree(f->pin);
    if (f->zpin) (void)hipHostFree(f->zpin);
    if (f->zpin_done) (void)hipEventDestroy(f->zpin_done);
    dev_free(f->acoef);
    dev_free(f->alpha_c);
    dev_free(f->zbuf);
    dev_free(f->zrand);
    dev_free(f->pose);
    dev_free(f->scalars);
    dev_free(f->fxbuf[0]);
    dev_free(f->fxbuf[1]);
    dev_free(f->alt_seg);
    dev_free(f->nfac[0]);
    dev_free(f->nfac[1]);
    dev_free(f->lp_sync);
    dev_free(f->lp_scratch);
    dev_free(f->retry);
    dev_free(f->part);
    dev_free(f->absmax);
    nn_grid_free(&f->tgrid);
    tri_grid_free(&f->ttgrid);
    dev_free(f->tperm);
    dev_free(f->tboxes);
    dev_free(f->fboxes);
    dev_free(f->tile_bad);
    dev_free(f->xch);
    dev_free(f->fullfit);
    dev_free(f->gstage);
    dev_free(f->zero_counts);
    dev_free(f->ws);
    dev_free(f->work);
    dev_free(f->aos);
    dev_free(f->lm_pid);
    dev_free(f->lm_xyz);
    dev_free(f->lm_cov);
    dev_free(f->lm_mask);
    free_meshes(f);
    delete f;
}

int gingr_fitter_set_target(gingr_fitter *f, int64_t N, const double *target_xyz) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    f->nn_warm = f->surf_nn_warm = f->surf_tri_warm = false;  // positions in another target (any in-range position would still be a valid start)
    f->forget_posteriors();  // the posterior memos describe other inputs
    gingr_ctx *ctx = f->ctx;
    if (N < 1 || N > INT32_MAX || !target_xyz) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "set_target: bad N");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t M = f->m->M;
    const int32_t rp = f->m->rp;
    dev_free(f->target);
    dev_free(f->inv_den);
    dev_free(f->Pt1);
    dev_free(f->xch);
    dev_free(f->ws);
    dev_free(f->aos);
    dev_free(f->tperm);
    dev_free(f->tboxes);
    dev_free(f->fboxes);
    dev_free(f->tile_bad);
    free_meshes(f);  // the target triangles refer to the previous target
    f->target = f->inv_den = f->Pt1 = f->xch = f->ws = f->tboxes = f->fboxes = nullptr;
    f->aos = nullptr;
    f->tperm = f->tile_bad = nullptr;
    f->N = N;
    GINGR_TRY(dev_alloc(ctx, &f->target, (size_t)3 * N));
    GINGR_TRY(dev_alloc(ctx, &f->inv_den, (size_t)N));
    GINGR_TRY(dev_alloc(ctx, &f->Pt1, (size_t)N));
    // exchange segments (float64 elements)
    f->cnt[0] = N;
    f->cnt[1] = (int64_t)rp * rp + rp + 8 + rp;  // G, rhs, the scalar sums, and Q0^T e of a sharded transition-density query
    int64_t o = 0;
    for (int s = 0; s < GINGR_NUM_SEGMENTS; ++s) {
        f->off[s] = o;
        o += round_up(f->cnt[s], 32);  // 256-byte aligned segments
    }
    GINGR_TRY(dev_alloc(ctx, &f->xch, (size_t)o));
    f->seg_swapped = false;  // (the posterior memos were forgotten above: nothing lives in either slot)
    HIP_TRY(ctx, hipMemsetAsync(f->xch, 0, (size_t)o * sizeof(double), ctx->stream));
    int64_t w = cpd_colsum_ws_doubles(M, N);
    auto mx = [&](int64_t v) {
        if (v > w) w = v;
    };
    mx(cpd_rowstats_ws_doubles(M, N));
    mx(ceil_div(nn_ws_bytes(M, N), 8));
    mx(ceil_div(nn_ws_bytes(N, M), 8));  // reversed correspondence direction: the targets query the model vertices
    mx(gram_ws_doubles(M, rp) + sweep_ws_doubles(M, rp));  // phase 1 keeps both sets of partials until its finalize kernel
    f->ws_doubles = w;
    GINGR_TRY(dev_alloc(ctx, &f->ws, (size_t)w));
    const int64_t big = M > N ? M : N;
    double *aos = nullptr;
    GINGR_TRY(dev_alloc(ctx, &aos, (size_t)3 * big));
    f->aos = aos;
    HIP_TRY(ctx, hipMemcpyAsync(aos, target_xyz, (size_t)3 * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    morton_order(target_xyz, N, f->h_tperm);
    GINGR_TRY(nn_grid_build(ctx, target_xyz, N, f->h_tperm.data(), M, &f->tgrid));
    GINGR_TRY(dev_alloc(ctx, &f->tperm, (size_t)N));
    HIP_TRY(ctx, hipMemcpyAsync(f->tperm, f->h_tperm.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    GINGR_TRY(dev_alloc(ctx, &f->tboxes, (size_t)ceil_div(N, 256) * 30));  // tile boxes + four quarter boxes per tile
    GINGR_TRY(dev_alloc(ctx, &f->fboxes, (size_t)ceil_div(M, 256) * 30));
    f->fit_boxes_valid = false;
    GINGR_TRY(dev_alloc(ctx, &f->tile_bad, (size_t)ceil_div(N, 256)));
    launch_aos_to_soa(ctx, aos, N, f->target, f->tperm);
    launch_tile_bbox(ctx, cloud_of(f->target, N), f->tboxes);
    launch_cloud_centroid(ctx, cloud_of(f->target, N), f->absmax + 2);
    launch_cloud_absmax(ctx, cloud_of(f->target, N), f->absmax + 2, f->absmax);
    if (f->has_state) fit_boxes_now(f);  // the fit on the device predates this target (its centre)
    GINGR_TRY(check_launch(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GINGR_OK;
}

int gingr_fitter_set_landmarks(gingr_fitter *f, int32_t n_lm, const int32_t *lm_pid, const double *lm_xyz,
                               const double *lm_cov) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    f->forget_posteriors();  // the posterior memos describe other inputs
    gingr_ctx *ctx = f->ctx;
    if (n_lm < 0 || (n_lm > 0 && (!lm_pid || !lm_xyz || !lm_cov)))
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "set_landmarks: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(f->lm_pid);
    dev_free(f->lm_xyz);
    dev_free(f->lm_cov);
    f->lm_pid = nullptr;
    f->lm_xyz = f->lm_cov = nullptr;
    f->n_lm = n_lm;
    const int64_t M = f->m->M;
    std::vector<int32_t> mask((size_t)M, 0), local((size_t)(n_lm > 0 ? n_lm : 1), -1);
    for (int32_t l = 0; l < n_lm; ++l) {
        if (lm_pid[l] < 0 || lm_pid[l] >= f->m->M_total)
            return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "set_landmarks: point id %d out of range", lm_pid[l]);
        const int64_t lp = (int64_t)lm_pid[l] - f->m->row_begin;
        if (lp >= 0 && lp < M) {
            const int32_t pos = f->m->hiperm[(size_t)lp];  // device (Morton) position of the original point
            local[(size_t)l] = pos;
            mask[(size_t)pos] = 1;
        }
    }
    HIP_TRY(ctx, hipMemcpy(f->lm_mask, mask.data(), (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice));
    if (n_lm > 0) {
        GINGR_TRY(dev_alloc(ctx, &f->lm_pid, (size_t)n_lm));
        GINGR_TRY(dev_alloc(ctx, &f->lm_xyz, (size_t)3 * n_lm));
        GINGR_TRY(dev_alloc(ctx, &f->lm_cov, (size_t)9 * n_lm));
        HIP_TRY(ctx, hipMemcpy(f->lm_pid, local.data(), (size_t)n_lm * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(f->lm_xyz, lm_xyz, (size_t)3 * n_lm * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(f->lm_cov, lm_cov, (size_t)9 * n_lm * sizeof(double), hipMemcpyHostToDevice));
    }
    return GINGR_OK;
}

int gingr_fitter_set_options(gingr_fitter *f, int32_t global_transform, double step_length) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    if (global_transform < 0 || global_transform > 2)
        return gingr_set_error(f->ctx, GINGR_ERR_BAD_ARGUMENT, "set_options: unknown global transformation %d", global_transform);
    f->global_transform = global_transform;
    f->step_length = step_length;
    return GINGR_OK;
}

int gingr_fitter_set_stop_threshold(gingr_fitter *f, double threshold) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (threshold != threshold) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "set_stop_threshold: NaN");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    f->stop_threshold = threshold < 0.0 ? -1.0 : threshold;
    f->stop_hit = 0;
    // a state the rule stopped at earlier takes updates again
    if (f->has_state) HIP_TRY(ctx, hipMemsetAsync(&f->st->stopped, 0, sizeof(int32_t), ctx->stream));
    return GINGR_OK;
}

int gingr_fitter_stop_rule_hit(gingr_fitter *f, int32_t *hit) {
    if (!f || !hit) return GINGR_ERR_BAD_ARGUMENT;
    *hit = f->stop_hit;
    return GINGR_OK;
}

int gingr_fitter_set_state(gingr_fitter *f, const double *alpha, const gingr_state_scalars *s) {
    if (!f || !alpha || !s) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int32_t r = f->m->r, rp = f->m->rp;
    memset(f->pin, 0, ((size_t)rp + kScalarsDoubles) * sizeof(double));
    memcpy(f->pin, alpha, (size_t)r * sizeof(double));
    memcpy(f->pin + rp, s, sizeof(*s));
    HIP_TRY(ctx, hipMemcpyAsync(f->state_block, f->pin, ((size_t)rp + kScalarsDoubles) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_state_init(ctx, f->st, f->hs_dev, f->absmax + 1);
    if (!f->ws) {  // no target yet: allocate the sweep workspace so the fit can be instantiated
        f->ws_doubles = sweep_ws_doubles(f->m->M, rp);
        GINGR_TRY(dev_alloc(ctx, &f->ws, (size_t)f->ws_doubles));
    }
    refresh_fit(f);
    GINGR_TRY(check_launch(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    f->has_state = true;
    f->mh_saved = false;
    f->state_key.v.assign(alpha, alpha + r);
    for (int q = 0; q < 3; ++q) f->state_key.v.push_back(s->euler[q]);
    for (int q = 0; q < 3; ++q) f->state_key.v.push_back(s->center[q]);
    for (int q = 0; q < 3; ++q) f->state_key.v.push_back(s->translation[q]);
    f->state_key.v.push_back(s->scale);
    f->state_key.v.push_back(s->sigma2);
    f->state_key_valid = true;
    return GINGR_OK;
}

int gingr_fitter_set_fit_points(gingr_fitter *f, const double *fit_xyz) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (!fit_xyz) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "set_fit_points: null argument");
    if (!f->has_state) return gingr_set_error(ctx, GINGR_ERR_STATE, "set_fit_points: set a state first (its pose / sigma2 stay in force)");
    if (!f->aos) return gingr_set_error(ctx, GINGR_ERR_STATE, "set_fit_points: set a target first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = f->m->M;
    double *stage = reinterpret_cast<double *>(f->aos);
    HIP_TRY(ctx, hipMemcpyAsync(stage, fit_xyz, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, stage, M, f->fit, f->m->perm);
    fit_boxes_now(f);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the caller's buffer is free again
    // the shape on the device is no instance of the model any more: nothing memoised describes it
    f->forget_posteriors();
    f->state_key_valid = false;
    f->mh_saved = false;
    return GINGR_OK;
}

// n doubles from the device into the pinned buffer at `dst` (a pointer INTO f->pin) without a copy + stream synchronisation: one small
// launch writes them through the buffer's device address and stores the launch number into the flag word, the host spins on it (see
// mh_readback_kernel).  Everything enqueued before on the stream is complete when this returns.
static int pull_small(gingr_fitter *f, const double *src, int n, double *dst) {
    gingr_ctx *ctx = f->ctx;
    if (!f->pin_dev) {
        HIP_TRY(ctx, hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return GINGR_OK;
    }
    volatile double *flag = f->pin + f->pin_doubles - 1;
    const double epoch = (double)(++f->mh_epoch);
    hipLaunchKernelGGL(mh_readback_kernel, dim3(1), dim3(256), 0, ctx->stream, src, n, (const double *)nullptr, (int64_t)0, (const int32_t *)nullptr,
                       f->pin_dev + (dst - f->pin), f->pin_dev + f->pin_doubles - 1, f->mh_done, epoch);
    GINGR_TRY(check_launch(ctx));
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
    bool seen = false;
    for (unsigned spins = 0;; ++spins) {
        if (*flag == epoch) {
            seen = true;
            break;
        }
        if ((spins & 1023u) == 1023u) {
            if (std::chrono::steady_clock::now() > deadline) break;
            if (spins > 65536u) std::this_thread::yield();  // (a long wait: leave the core to whoever else needs it)
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GINGR_OK;
}

int gingr_fitter_get_state(gingr_fitter *f, double *alpha, gingr_state_scalars *s, double *fit_xyz) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (!f->has_state) return gingr_set_error(ctx, GINGR_ERR_STATE, "get_state: no state set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = f->m->M;
    const int32_t rp_ = f->m->rp;
    const size_t head = (size_t)rp_ + kScalarsDoubles + kDevStateDoubles;  // [alpha | scalars | DevState], one transfer
    bool seen = false;
    if (f->pin_dev && M <= 8192) {
        // small templates: one launch gathers state and fit straight into the pinned buffer and the call spins on the flag word (as
        // gingr_fitter_mh_step does) instead of two copies, a reordering launch and a stream synchronisation
        volatile double *flag = f->pin + f->pin_doubles - 1;
        const double epoch = (double)(++f->mh_epoch);
        const int64_t n = fit_xyz ? std::max<int64_t>(M, (int64_t)head) : (int64_t)head;
        hipLaunchKernelGGL(mh_readback_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, f->state_block, (int)head,
                           fit_xyz ? f->fit : (const double *)nullptr, M, f->m->perm, f->pin_dev, f->pin_dev + f->pin_doubles - 1, f->mh_done, epoch);
        GINGR_TRY(check_launch(ctx));
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
        for (unsigned spins = 0;; ++spins) {
            if (*flag == epoch) {
                seen = true;
                break;
            }
            if ((spins & 1023u) == 1023u) {
                if (std::chrono::steady_clock::now() > deadline) break;
                if (spins > 65536u) std::this_thread::yield();  // (a long wait: leave the core to whoever else needs it)
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!seen) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (a launch that never finished: the error is reported here)
    }
    if (!seen) HIP_TRY(ctx, hipMemcpyAsync(f->pin, f->state_block, head * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DevBuf tmp;
    if (fit_xyz && !seen) {
        double *stage = reinterpret_cast<double *>(f->aos);  // the fitter's interleaved staging buffer (max(3M, 3N) doubles)
        if (!stage) {                                        // no target yet: a temporary
            HIP_TRY(ctx, tmp.alloc((size_t)3 * M * sizeof(double)));
            stage = tmp.as<double>();
        }
        launch_soa_to_aos(ctx, f->fit, M, stage, f->m->perm);
        HIP_TRY(ctx, hipMemcpyAsync(f->pin + head, stage, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!seen) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    DevState hst;
    memcpy(&hst, f->pin + rp_ + kScalarsDoubles, sizeof(hst));
    if (alpha) memcpy(alpha, f->pin, (size_t)f->m->r * sizeof(double));
    if (fit_xyz) memcpy(fit_xyz, f->pin + head, (size_t)3 * M * sizeof(double));
    if (s) {
        for (int q = 0; q < 3; ++q) {
            s->euler[q] = hst.euler[q];
            s->center[q] = hst.center[q];
            s->translation[q] = hst.t[q];
        }
        s->scale = hst.scale;
        s->sigma2 = hst.sigma2;
        s->iteration = hst.iteration;
        s->status = hst.status;
    }
    f->stop_hit = hst.stopped;
    if (alpha) {  // what was just read IS the device state: the posterior memo can recognise it without a gingr_fitter_set_state
        f->state_key.v.assign(alpha, alpha + f->m->r);
        for (int q = 0; q < 3; ++q) f->state_key.v.push_back(hst.euler[q]);
        for (int q = 0; q < 3; ++q) f->state_key.v.push_back(hst.center[q]);
        for (int q = 0; q < 3; ++q) f->state_key.v.push_back(hst.t[q]);
        f->state_key.v.push_back(hst.scale);
        f->state_key.v.push_back(hst.sigma2);
        f->state_key_valid = true;
    }
    return GINGR_OK;
}

int gingr_fitter_get_cpd_stats(gingr_fitter *f, double *P1, double *PX, double *den, double *scalars6) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (f->corr_stale)  // see gingr_fitter::alt_seg
        return gingr_set_error(ctx, GINGR_ERR_STATE, "get_cpd_stats: a probabilistic query brought another state's posterior back; the correspondences on the device are not this state's -- run an update or a phase first");
    if (!f->target) return gingr_set_error(ctx, GINGR_ERR_STATE, "get_cpd_stats: no target");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = f->m->M;
    DevBuf tmp, tmp1, tmpd;
    if (P1) {  // back to the caller's point order
        HIP_TRY(ctx, tmp1.alloc((size_t)M * sizeof(double)));
        launch_scatter(ctx, f->P1, M, f->m->perm, tmp1.as<double>());
        HIP_TRY(ctx, hipMemcpyAsync(P1, tmp1.p, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (PX) {
        HIP_TRY(ctx, tmp.alloc((size_t)3 * M * sizeof(double)));
        launch_soa_to_aos(ctx, f->PX, M, tmp.as<double>(), f->m->perm);
        HIP_TRY(ctx, hipMemcpyAsync(PX, tmp.p, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (den) {
        HIP_TRY(ctx, tmpd.alloc((size_t)f->N * sizeof(double)));
        launch_scatter(ctx, f->xch + f->off[0], f->N, f->tperm, tmpd.as<double>());
        HIP_TRY(ctx, hipMemcpyAsync(den, tmpd.p, (size_t)f->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    double sc[8];
    const double *red = f->seg1_live() + (int64_t)f->m->rp * f->m->rp + f->m->rp;
    HIP_TRY(ctx, hipMemcpyAsync(sc, red, sizeof(sc), hipMemcpyDeviceToHost, ctx->stream));
    double loc[8];
    HIP_TRY(ctx, hipMemcpyAsync(loc, f->scalars, sizeof(loc), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (scalars6) {
        scalars6[0] = sc[0];
        scalars6[1] = sc[1];
        scalars6[2] = sc[2];
        scalars6[3] = sc[3];
        scalars6[4] = (sc[1] - 2 * sc[2] + sc[3]) / (sc[0] * 3.0);
        scalars6[5] = loc[5];
    }
    return GINGR_OK;
}

int gingr_fitter_get_icp_idx(gingr_fitter *f, int32_t *idx, double *d2) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (f->corr_stale)  // see gingr_fitter::alt_seg
        return gingr_set_error(ctx, GINGR_ERR_STATE, "get_icp_idx: a probabilistic query brought another state's posterior back; the correspondences on the device are not this state's -- run an update or a phase first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = f->m->M;
    std::vector<int32_t> hidx((size_t)M);
    std::vector<double> hd2((size_t)M);
    HIP_TRY(ctx, hipMemcpyAsync(hidx.data(), f->nn_idx, (size_t)M * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hd2.data(), f->nn_d2, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // device positions -> the caller's numbering (rows and targets are kept in Morton order on the device)
    for (int64_t sidx = 0; sidx < M; ++sidx) {
        const int32_t row = f->m->hperm[(size_t)sidx];
        const int32_t pos = hidx[(size_t)sidx];
        if (idx) idx[row] = (pos >= 0 && (size_t)pos < f->h_tperm.size()) ? f->h_tperm[(size_t)pos] : -1;
        if (d2) d2[row] = hd2[(size_t)sidx];
    }
    return GINGR_OK;
}

int gingr_fitter_retry_counter(gingr_fitter *f, int32_t set_to, int32_t *value_out) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (set_to >= 0) HIP_TRY(ctx, hipMemcpyAsync(f->retry, &set_to, sizeof(set_to), hipMemcpyHostToDevice, ctx->stream));
    int32_t v = set_to;
    if (value_out && set_to < 0) HIP_TRY(ctx, hipMemcpyAsync(&v, f->retry, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (value_out) *value_out = v;
    return GINGR_OK;
}

int gingr_fitter_exchange(gingr_fitter *f, void **dev_ptr, int64_t offsets[GINGR_NUM_SEGMENTS],
                          int64_t counts[GINGR_NUM_SEGMENTS]) {
    if (!f || !dev_ptr) return GINGR_ERR_BAD_ARGUMENT;
    if (!f->xch) return gingr_set_error(f->ctx, GINGR_ERR_STATE, "exchange: no target set");
    *dev_ptr = f->xch;
    for (int s = 0; s < GINGR_NUM_SEGMENTS; ++s) {
        if (offsets) offsets[s] = f->off[s];
        if (counts) counts[s] = f->cnt[s];
    }
    return GINGR_OK;
}
// idx / d2 = nearest TARGET vertex of every query (positions in the target's device order; lowest original index on ties): the grid
// search over the fixed target cloud first (nn_grid.hip), then the tile scan masked to the queries the grid could not certify -- a
// launch that exits at once when there are none.  warm: idx holds the previous matches of the same queries.
static void nearest_target_vertex(gingr_ctx *ctx, gingr_fitter *f, Cloud query, Cloud tgt, int32_t *idx, double *d2, bool warm) {
    const int32_t *w = warm ? idx : nullptr;
    if (ctx->nn_grid && f->tgrid.ready && ctx->cull && query.n <= f->tgrid.max_queries) {
        if (!launch_nn_grid(ctx, query, tgt, f->tperm, f->tgrid, w, idx, d2))  // (true: a small cloud, nothing left to scan)
            launch_nn(ctx, query, tgt, f->tperm, f->tboxes, f->ws, idx, d2, idx, f->tgrid.flag, f->tgrid.cur_nflag());
    } else {
        launch_nn(ctx, query, tgt, f->tperm, f->tboxes, f->ws, idx, d2, w);
    }
}


}  // extern "C"

// --------------------------------------------------------------------------------------------------- internal hooks (group.hip)
void fitter_set_partial_output(gingr_fitter *f, double *base) { f->partial_out = base; }
void fitter_set_partial_fullfit(gingr_fitter *f, double *base) { f->partial_fullfit = base; }
double *fitter_fullfit(gingr_fitter *f) { return f->fullfit; }
void fitter_set_partial_revsum(gingr_fitter *f, double *base) { f->partial_revsum = base; }
double *fitter_revsum(gingr_fitter *f) { return f->revsum; }
// z (r standard normals, host) -> f->zrand (rp doubles, zero padded) on the context's stream, without waiting for the stream
int fitter_upload_zrand(gingr_fitter *f, const double *z) {
    gingr_ctx *ctx = f->ctx;
    const size_t bytes = (size_t)f->m->rp * sizeof(double);
    if (!f->zpin) {
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&f->zpin), bytes, hipHostMallocDefault));
        if (hipEventCreateWithFlags(&f->zpin_done, hipEventDisableTiming) != hipSuccess) {
            (void)hipHostFree(f->zpin);
            f->zpin = nullptr, f->zpin_done = nullptr;
            return gingr_set_error(ctx, GINGR_ERR_HIP, "upload of the posterior draws: hipEventCreate failed");
        }
    } else {
        HIP_TRY(ctx, hipEventSynchronize(f->zpin_done));  // the previous upload has read the buffer (long ago, in practice)
    }
    memset(f->zpin, 0, bytes);
    memcpy(f->zpin, z, (size_t)f->m->r * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(f->zrand, f->zpin, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(f->zpin_done, ctx->stream));
    return GINGR_OK;
}
int fitter_set_zrand(gingr_fitter *f, const double *z) {
    f->zrand_active = false;
    if (!z) return GINGR_OK;
    GINGR_TRY(fitter_upload_zrand(f, z));
    f->zrand_active = true;  // only once the draws are on their way: a failed upload must not sample from stale ones
    return GINGR_OK;
}
gingr_ctx *fitter_ctx(gingr_fitter *f) { return f->ctx; }
// this shard's rows are shard `rank` of the balanced partition over `world` shards, and it has the gathered-fit buffer (pure check)
bool fitter_gather_possible(gingr_fitter *f, int32_t world, int32_t rank) {
    if (!f || !f->fullfit || world < 1 || rank < 0 || rank >= world) return false;
    const gingr_model *m = f->m;
    const int64_t Mt = m->M_total, base = Mt / world, extra = Mt % world;
    const int64_t b = rank * base + (rank < extra ? rank : extra), e = b + base + (rank < extra ? 1 : 0);
    return b == m->row_begin && e - b == m->M;
}
int fitter_gather_agreed(gingr_fitter *f, int32_t world) { return f->gather_agreed_world == world ? f->gather_agreed : -1; }
void fitter_set_gather_agreed(gingr_fitter *f, int32_t world, int agreed) {
    f->gather_agreed = agreed;
    f->gather_agreed_world = world;
}
bool fitter_reversed(gingr_fitter *f) { return f->reversed; }
const gingr_model *fitter_model(gingr_fitter *f) { return f->m; }

// --------------------------------------------------------------------------------------------------- phases
namespace {

int run_phase(gingr_fitter *f, bool icp, const gingr_cpd_params *cp, const gingr_icp_params *ip, int phase) {
    gingr_ctx *ctx = f->ctx;
    const gingr_model *m = f->m;
    const int64_t M = m->M;
    const int32_t r = m->r, rp = m->rp;
    if (f->seg_swapped && !f->allow_alt) {  // this entry point works on the exchange buffer itself: the live segment moves back
        const int64_t seg = (int64_t)rp * rp + rp + 8;
        hipLaunchKernelGGL(swap_segments_kernel, dim3((unsigned)ceil_div(seg, 256)), dim3(256), 0, ctx->stream, f->xch + f->off[1], f->alt_seg,
                           seg, 0);
        f->seg_swapped = false;
        f->alt_stage = 0;  // (what was parked there is given up)
        f->fx_valid[f->live ^ 1] = f->nf_valid[f->live ^ 1] = false;
    }
    // reduced (summed over shards) segments, read by phases 1 and 2 ...
    double *seg0 = f->xch + f->off[0];
    double *G = f->seg1_live();
    double *rhs = G + (int64_t)rp * rp;
    double *sc8 = rhs + rp;
    // ... and where this shard's partial sums are written by phases 0 and 1
    double *wbase = f->partial_out ? f->partial_out : f->xch;
    double *seg0w = wbase + f->off[0];
    double *Gw = f->partial_out ? wbase + f->off[1] : G;
    double *rhsw = Gw + (int64_t)rp * rp;
    double *sc8w = rhsw + rp;
    const Cloud fit = cloud_of(f->fit, M);
    const Cloud tgt = cloud_of(f->target, f->N);
    if (phase == GINGR_PHASE_GATHER) {  // sharded surface ICP: this shard's rows of the fit into the full-fit buffer (gingr_fitter::fullfit)
        if (!f->sharded()) return GINGR_OK;
        if (!f->fullfit) return gingr_set_error(ctx, GINGR_ERR_STATE, "gather phase: no meshes set (gingr_fitter_set_meshes)");
        hipLaunchKernelGGL(fit_contribution_kernel, dim3((unsigned)ceil_div(m->M_total, 256)), dim3(256), 0, ctx->stream, f->fit, m->iperm, M,
                           m->row_begin, m->M_total, f->partial_fullfit ? f->partial_fullfit : f->fullfit);
        return check_launch(ctx);
    }
    // the template mesh of the surface tests: the fit itself, or -- on a row shard -- the gathered fit of all shards (original order)
    const Cloud meshc = f->sharded() && f->fullfit ? cloud_of(f->fullfit, m->M_total) : fit;
    // posterior memo (see gingr_fitter::Key): skip phases 0 and 1 when their results for exactly this state are still in place
    if (phase == 0) {
        f->skip_phase1 = false;
        gingr_fitter::Key k;
        k.flavour = !icp ? 0 : ((f->icp_surface ? 2 : 1) + 4 * f->surface_method + 16 * (f->reversed ? 1 : 0));
        if (!icp) {
            k.p0 = cp->w;
            k.p1 = cp->lambda;
        }
        const bool single = m->M == m->M_total;
        if (single && f->state_key_valid) {
            k.v = f->state_key.v;
            if (f->post_stage == 2 && f->post_key.same(k)) {
                f->skip_phase1 = true;
                return GINGR_OK;
            }
            if (f->allow_alt && !f->partial_out && f->alt_stage == 2 && f->alt_key.same(k)) {
                // the other slot holds this state: the two slots exchange roles (no copy)
                const bool both = f->post_stage == 2;
                f->seg_swapped = !f->seg_swapped;
                if (both) {
                    std::swap(f->post_key, f->alt_key);
                } else {  // the live slot held nothing finished: nothing is parked now
                    f->post_key = f->alt_key;
                    f->alt_stage = 0;
                    f->fx_valid[f->live] = f->nf_valid[f->live] = false;
                }
                f->live ^= 1;
                f->post_stage = 2;
                f->corr_stale = true;
                f->skip_phase1 = true;
                return GINGR_OK;
            }
            if (f->allow_alt && !f->partial_out && f->post_stage == 2) {  // keep what is about to be overwritten: it becomes the parked slot
                f->seg_swapped = !f->seg_swapped;
                f->alt_key = f->post_key;
                f->alt_stage = 2;
                f->live ^= 1;  // its factors stay with it
            }
            f->fx_valid[f->live] = f->nf_valid[f->live] = false;
            f->post_key = k;
            f->post_stage = 1;
        } else {
            f->post_stage = 0;
            f->fx_valid[f->live] = f->nf_valid[f->live] = false;
        }
        f->corr_stale = false;  // phase 0 recomputes the correspondences of this state
    } else if (phase == 1) {
        if (f->skip_phase1) {
            f->skip_phase1 = false;
            return GINGR_OK;
        }
        if (f->post_stage == 1) f->post_stage = 2;
    } else {
        f->state_key_valid = false;  // the commit moves the device state away from the key
        f->mh_saved = false;
    }
    switch (phase) {
        case 0: {
            if (icp && f->reversed && f->sharded()) {
                // The same correspondence as below against the GATHERED template (meshc: original vertex order, so a matched vertex IS
                // its original id and ties go to the lowest id as on a single shard), for THIS shard's range of the target queries
                // only; what leaves the phase is the per-template-vertex sums of the range (see gingr_fitter::revsum).  The tests that
                // involve the target mesh itself (self-intersection) see the whole target.
                if (!f->fullfit || !f->revsum)
                    return gingr_set_error(ctx, GINGR_ERR_STATE, "reversed correspondence direction on a row shard: meshes / direction not set");
                const int64_t Mt = m->M_total, q0 = f->rq0, nq = f->rqn;
                const Cloud tq{tgt.x + q0, tgt.y + q0, tgt.z + q0, nq};
                double *sums = f->partial_revsum ? f->partial_revsum : f->revsum;
                if (!f->gperm) {  // one-off (synchronises once): the spatial order of the gathered template for the vertex search
                    std::vector<double> soa((size_t)3 * Mt), aosv((size_t)3 * Mt);
                    HIP_TRY(ctx, hipMemcpyAsync(soa.data(), f->fullfit, soa.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    for (int64_t g = 0; g < Mt; ++g)
                        for (int d = 0; d < 3; ++d) aosv[(size_t)(3 * g + d)] = soa[(size_t)(d * Mt + g)];
                    std::vector<int32_t> order;
                    morton_order(aosv.data(), Mt, order);
                    // (failure-atomic: the three members are set together, once everything exists and the order is on the device --
                    // a half-built set would make every later phase 0 skip this block and index with garbage)
                    int32_t *gperm = nullptr, *rnn_pos = nullptr;
                    double *gsorted = nullptr;
                    int rc = dev_alloc(ctx, &gperm, (size_t)Mt);
                    if (!rc) rc = dev_alloc(ctx, &gsorted, (size_t)3 * Mt);
                    if (!rc) rc = dev_alloc(ctx, &rnn_pos, (size_t)(f->N > 0 ? f->N : 1));
                    if (!rc && hipMemcpy(gperm, order.data(), (size_t)Mt * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
                        rc = gingr_set_error(ctx, GINGR_ERR_HIP, "reversed direction: copying the template order failed");
                    if (rc) {
                        dev_free(gperm), dev_free(gsorted), dev_free(rnn_pos);
                        return rc;
                    }
                    f->gperm = gperm, f->gsorted = gsorted, f->rnn_pos = rnn_pos;
                }
                hipLaunchKernelGGL(soa_permute_kernel, dim3((unsigned)ceil_div(Mt, 256)), dim3(256), 0, ctx->stream, f->fullfit, f->gperm, Mt, f->gsorted);
                const Cloud msort = cloud_of(f->gsorted, Mt);
                launch_tile_bbox(ctx, msort, f->rfboxes);
                auto nearest_template_vertex = [&](Cloud q) {  // -> f->rnn: ORIGINAL vertex ids (lowest id on exact ties)
                    // (warm start from the last search: the queries are the same target vertices, the template moved a little -- with it the
                    // chunks a small query range is split into all start from a tight bound)
                    launch_nn(ctx, q, msort, f->gperm, f->rfboxes, f->rws, f->rnn_pos, f->rnnd2, f->rnn_warm ? f->rnn_pos : nullptr);
                    f->rnn_warm = true;
                    hipLaunchKernelGGL(index_map_kernel, dim3((unsigned)ceil_div(q.n, 256)), dim3(256), 0, ctx->stream, f->rnn_pos, q.n, f->gperm, f->rnn);
                };
                if (f->icp_surface) {
                    const bool along = f->surface_method == 1;
                    launch_cell_normals(ctx, meshc, f->mtri, f->Tm, f->mcn);
                    launch_vertex_normals(ctx, f->radj_ptr, f->radj_tri, f->mcn, f->Tm, Mt, f->rmvn);
                    launch_tri_tile_bbox(ctx, meshc, f->mtri, f->Tm, f->mtboxes, f->mtribox);
                    if (nq > 0) {
                        if (along)
                            launch_line_nearest(ctx, tq, f->rtvn_loc, meshc, f->mtri, f->mtri_orig, f->Tm, f->mtboxes, f->rcp, f->rhit);
                        else
                        {
                            launch_surface_closest_point(ctx, tq, meshc, f->mtri, f->mtri_orig, f->Tm, f->mtboxes, f->rcp, f->rd2, nullptr, f->rtri_pos,
                                                         f->rtri_warm, f->mtribox);
                            f->rtri_warm = true;
                        }
                        nearest_template_vertex(cloud_of(f->rcp, nq));
                        SelfIntersectFuse fu;  // (the first two rejection tests ride in the self-intersection launch)
                        fu.nn_vertex = f->rnn, fu.boundary = f->rmbnd, fu.q_vn = f->rtvn_loc, fu.t_vn = f->rmvn, fu.Nt = Mt;
                        fu.found = along ? f->rhit : nullptr, fu.pre_out = f->rpre;
                        launch_self_intersect(ctx, tq, f->rcp, f->ttri, f->Tt, f->ttboxes, nullptr, f->rhit, f->ttribox, &tgt, nullptr, nullptr, &fu);
                    }
                    launch_reversal_sums(ctx, Mt, tq, f->rnn, f->rpre, f->rhit, f->rkeys, f->rvals, f->rskeys, f->rsvals, f->rsort, f->rsort_bytes,
                                         f->rw01 + q0, sums);
                } else {
                    if (nq > 0) nearest_template_vertex(tq);
                    launch_reversal_sums(ctx, Mt, tq, f->rnn, nullptr, nullptr, f->rkeys, f->rvals, f->rskeys, f->rsvals, f->rsort, f->rsort_bytes,
                                         f->rw01 + q0, sums);
                }
            } else if (icp && f->reversed) {
                // closestPointCorrespondenceReversal (ClosestPointRegistrator.scala:34-49): the roles of the two meshes are swapped,
                // then every accepted target vertex becomes an observation of the template vertex nearest to its match
                const int64_t N = f->N;
                launch_tile_bbox(ctx, fit, f->fboxes);
                if (f->icp_surface) {
                    const bool along = f->surface_method == 1;
                    launch_cell_normals(ctx, fit, f->mtri, f->Tm, f->mcn);
                    launch_vertex_normals(ctx, f->madj_ptr, f->madj_tri, f->mcn, f->Tm, M, f->mvn);
                    launch_tri_tile_bbox(ctx, fit, f->mtri, f->Tm, f->mtboxes, f->mtribox);
                    if (along)
                        launch_line_nearest(ctx, tgt, f->tvn, fit, f->mtri, f->mtri_orig, f->Tm, f->mtboxes, f->rcp, f->rhit);
                    else
                    {
                        launch_surface_closest_point(ctx, tgt, fit, f->mtri, f->mtri_orig, f->Tm, f->mtboxes, f->rcp, f->rd2, nullptr, f->rtri_pos,
                                                     f->rtri_warm, f->mtribox);
                        f->rtri_warm = true;
                    }
                    launch_nn(ctx, cloud_of(f->rcp, N), fit, f->m->perm, f->fboxes, f->ws, f->rnn, f->rnnd2, f->rnn_warm ? f->rnn : nullptr);
                    f->rnn_warm = true;  // (rnn: positions in the fit's device order -- last iteration's matches start this one's scan)
                    SelfIntersectFuse fu;
                    fu.nn_vertex = f->rnn, fu.boundary = f->mboundary, fu.q_vn = f->tvn, fu.t_vn = f->mvn, fu.Nt = M;
                    fu.found = along ? f->rhit : nullptr, fu.pre_out = f->rpre;
                    launch_self_intersect(ctx, tgt, f->rcp, f->ttri, f->Tt, f->ttboxes, nullptr, f->rhit, f->ttribox, nullptr, nullptr, nullptr, &fu);
                    launch_reversal_observations(ctx, M, tgt, f->rnn, f->rpre, f->rhit, &f->st->sigma2, f->rkeys, f->rvals, f->rskeys,
                                                 f->rsvals, f->rsort, f->rsort_bytes, f->rw01, f->robs, f->rwin);
                } else {  // ClosestPointTriangleMesh3DSimple: nearest template vertex, weight 1
                    launch_nn(ctx, tgt, fit, f->m->perm, f->fboxes, f->ws, f->rnn, f->rnnd2, f->rnn_warm ? f->rnn : nullptr);
                    f->rnn_warm = true;
                    launch_reversal_observations(ctx, M, tgt, f->rnn, nullptr, nullptr, &f->st->sigma2, f->rkeys, f->rvals, f->rskeys,
                                                 f->rsvals, f->rsort, f->rsort_bytes, f->rw01, f->robs, f->rwin);
                }
            } else if (icp && f->icp_surface) {
                // ClosestPointTriangleMesh3D.closestPointCorrespondence (ClosestPointRegistrator.scala:75-100)
                launch_tri_tile_bbox(ctx, meshc, f->mtri, f->Tm, f->mtboxes, f->mtribox, f->mcn);  // boxes + cell normals of the template
                launch_vertex_normals(ctx, f->madj_ptr, f->madj_tri, f->mcn, f->Tm, M, f->mvn);
                const bool along = f->surface_method == 1;  // ClosestPointAlongNormalTriangleMesh3D (:102-131)
                if (along && ctx->tri_grid && (ctx->tri_grid == 2 || f->Tt >= kTriGridMinTriangles) && ctx->cull && f->ttgrid.ready)
                    launch_line_nearest_grid(ctx, fit, f->mvn, f->ttgrid, f->surf_cp, f->surf_hit);
                else if (along)
                    launch_line_nearest(ctx, fit, f->mvn, tgt, f->ttri, f->ttri_orig, f->Tt, f->ttboxes, f->surf_cp, f->surf_hit);
                else if (ctx->tri_grid && (ctx->tri_grid == 2 || f->Tt >= kTriGridMinTriangles) && ctx->cull && f->ttgrid.ready &&
                         f->surf_tri_warm && M <= f->ttgrid.max_queries) {
                    // grid search from the previous iteration's triangles, then the masked tile scan for what it flagged
                    launch_surface_cp_grid(ctx, fit, tgt, f->ttri, f->ttri_orig, f->Tt, f->ttgrid, f->surf_cp, f->surf_d2, nullptr, f->surf_tri_pos);
                    launch_surface_closest_point(ctx, fit, tgt, f->ttri, f->ttri_orig, f->Tt, f->ttboxes, f->surf_cp, f->surf_d2, nullptr,
                                                 f->surf_tri_pos, true, f->ttribox, f->ttgrid.flag, f->ttgrid.cur_nflag());
                } else {
                    launch_surface_closest_point(ctx, fit, tgt, f->ttri, f->ttri_orig, f->Tt, f->ttboxes, f->surf_cp, f->surf_d2, nullptr,
                                                 f->surf_tri_pos, f->surf_tri_warm, f->ttribox);
                    f->surf_tri_warm = true;
                }
                nearest_target_vertex(ctx, f, cloud_of(f->surf_cp, M), tgt, f->surf_nn, f->surf_nnd2, f->surf_nn_warm);
                f->surf_nn_warm = true;
                if (ctx->tri_grid == 2 && ctx->cull && f->mgrid.ready && M <= f->mgrid.max_queries) {
                    launch_surface_prereject(ctx, M, f->surf_nn, f->tboundary, f->mvn, f->tvn, f->N, along ? f->surf_hit : nullptr,
                                             f->surf_pre);
                    // GINGR_OPT_TRI_GRID = 2 only: the template's triangles binned for THIS iteration (boxes and tile boxes are the ones
                    // computed above), the test over the cells the segment's ball reaches, the tile scan for what that could not
                    // certify.  Same decisions; NOT the default -- at 41k x 82k the four build launches (setup, count, scan, fill:
                    // ~30 us) + the query (31 us) lose to the barrier-free tile scan (50 us): tools/experiments/README.md, round 5
                    launch_mov_grid_build(ctx, f->mgrid, meshc, f->mtri, nullptr, f->mtribox, f->mtboxes);
                    launch_self_intersect_grid(ctx, fit, f->surf_cp, f->mgrid, f->surf_pre, f->surf_hit);
                    launch_self_intersect(ctx, fit, f->surf_cp, f->mtri, f->Tm, f->mtboxes, f->surf_pre, f->surf_hit, f->mtribox, &meshc,
                                          f->mgrid.flag, f->mgrid.cur_nflag());
                    launch_surface_weight(ctx, M, f->surf_pre, f->surf_hit, &f->st->sigma2, f->surf_w01, f->surf_win);
                } else {
                    // one launch: the first two rejection tests in its prologue, the third (self-intersection) in its tile scan, the
                    // weights in its epilogue (until round 5: surface_prereject_kernel + this + surface_weight_kernel)
                    SelfIntersectFuse fu;
                    fu.nn_vertex = f->surf_nn, fu.boundary = f->tboundary, fu.q_vn = f->mvn, fu.t_vn = f->tvn, fu.Nt = f->N;
                    fu.found = along ? f->surf_hit : nullptr, fu.pre_out = f->surf_pre;
                    fu.sigma2 = &f->st->sigma2, fu.w01 = f->surf_w01, fu.weight_in = f->surf_win;
                    launch_self_intersect(ctx, fit, f->surf_cp, f->mtri, f->Tm, f->mtboxes, nullptr, f->surf_hit, f->mtribox, &meshc, nullptr, nullptr, &fu);
                }
            } else if (icp) {
                nearest_target_vertex(ctx, f, fit, tgt, f->nn_idx, f->nn_d2, f->nn_warm);
                f->nn_warm = true;
            } else {
                // (the quarter boxes of the fit and its |coordinate - centroid| maximum were left by the pass that wrote the fit:
                // refresh_fit / fit_boxes_now)
                f->cpd_seen = true;
                if (!f->fit_boxes_valid) fit_boxes_now(f);  // (first CPD phase of this fitter, or the fit was written while it ran ICP)
                // single shard: nothing is exchanged, so the chunk partials stay in ws and phase 1's den_finalize adds them up
                const bool alone = m->M == m->M_total && !f->partial_out;
                if (f->split_half != 0 && !alone) {
                    // one half of the target tiles (tile-aligned cut): its own launch, chunk plan and slice of the workspace
                    const int64_t NA = split_cut(tgt.n);
                    const bool first = f->split_half == 1;
                    const Cloud th = first ? Cloud{tgt.x, tgt.y, tgt.z, NA} : Cloud{tgt.x + NA, tgt.y + NA, tgt.z + NA, tgt.n - NA};
                    // (twice the chunks of the whole pass: half the targets x half-length chunks = the same number of workgroups)
                    const int nch2 = 2 * cpd_colsum_chunks(M, tgt.n);
                    (void)launch_cpd_colsum(ctx, fit, th, &f->st->sigma2, f->absmax, f->fboxes, first ? f->ws : f->ws + (int64_t)nch2 * NA,
                                            seg0w + (first ? 0 : NA), nch2);
                    f->colsum_chunks = 0;
                    break;
                }
                f->colsum_chunks = launch_cpd_colsum(ctx, fit, tgt, &f->st->sigma2, f->absmax, f->fboxes, f->ws, alone ? nullptr : seg0w);
                if (!alone) f->colsum_chunks = 0;
            }
            break;
        }
        case 1: {
            Phase1FinalizeArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.rp = rp;
            fa.G = Gw;
            fa.rhs = rhsw;
            fa.sc8 = sc8w;
            if (icp) {
                if (f->reversed && f->sharded())  // the totals over all shards' query ranges are in place: this shard's rows of them
                    hipLaunchKernelGGL(reversal_local_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, M, m->row_begin, m->perm,
                                       m->M_total, f->revsum, &f->st->sigma2, f->robs, f->rwin);
                if (f->reversed)  // one observation per template vertex: mean of its accepted targets, weight count / sigma2
                    launch_obs_points(ctx, m, f->st, f->robs, f->rwin, f->weight, f->evec, f->lm_mask);
                else if (f->icp_surface)  // only the weight-1 pairs are observed (ICP.scala:50): weight 0 drops the row
                    launch_obs_points(ctx, m, f->st, f->surf_cp, f->surf_win, f->weight, f->evec, f->lm_mask, f->zero_counts);
                else if (f->n_lm != 0)  // (without landmarks the observation is formed inside the right-hand-side pass below)
                    launch_obs_icp(ctx, m, f->st, tgt, f->nn_idx, f->lm_mask, f->weight, f->evec);
                fa.scalar_mode = 0;
            } else {
                launch_cpd_den_finalize(ctx, tgt, &f->st->sigma2, cp->w, m->M_total, seg0, f->inv_den, f->Pt1, f->tile_bad, f->part,
                                        f->scalars, f->colsum_chunks > 0 ? f->ws : nullptr, f->colsum_chunks);
                f->colsum_chunks = 0;
                // observations (correspondence point, uncertainty) come out of the row-statistics reduction; the scalar sums are
                // finished by the finalize kernel below
                CpdObsArgs ob;
                memset(&ob, 0, sizeof(ob));
                ob.ref = m->ref;
                ob.mean = m->mean;
                ob.sigma2 = &f->st->sigma2;
                ob.R = f->st->R;
                ob.center = f->st->center;
                ob.t = f->st->t;
                ob.lambda = cp->lambda;
                ob.lm_mask = f->lm_mask;
                ob.weight = f->weight;
                ob.evec = f->evec;
                launch_cpd_rowstats(ctx, fit, tgt, &f->st->sigma2, f->absmax, f->inv_den, f->tboxes, f->tile_bad, f->ws, f->P1,
                                    f->PX, f->part, f->scalars, nullptr, 0, &ob, false);
                fa.scalar_mode = 1;
                fa.part = f->part;
                fa.scalars_local = f->scalars;
                fa.contribute_xpx = m->row_begin == 0 ? 1 : 0;
            }
            double *gram_ws = f->ws, *sweep_ws = f->ws + gram_ws_doubles(M, rp);
            bool rhs_done = false;
            ZeroGate rhs_gate{};  // (no gate)
            if (icp && !f->icp_surface && !f->reversed && f->n_lm == 0) {
                // point-cloud ICP without landmarks: every row has the same weight 1 / sigma2 (ICP.scala:90-92), so the weighted Gram
                // is the model's one-off moment Q^T Q scaled -- no pass over the basis.  mom holds the total over ALL shards: the
                // shard that owns row 0 contributes it, the others contribute zero to the exchange.
                // (written by the phase-1 finalize kernel below: one launch less than a copy kernel of its own)
                fa.nslabs = 0;
                fa.scaled_src = m->mom + MomentLayout{rp}.stot();
                fa.sigma2 = &f->st->sigma2;
                fa.scaled_contribute = m->row_begin == 0 ? 1 : 0;
            } else if (icp && f->icp_surface && !f->reversed &&
                       (ctx->gram_downdate == 1 || (ctx->gram_downdate < 0 && M >= kGramDowndateMinRows))) {
                // surface correspondence: an accepted pair has the weight 1 / sigma2, a rejected one (and a vertex a landmark overrides) 0
                // (ICP.scala:50,90-92) -- the weighted Gram is the model's moment minus the rows of the zero-weight vertices, scaled.
                // One pass over THOSE rows (0.2 % of them at 41k x 82k) instead of the MFMA pass over the whole basis (44 us); the
                // right-hand side takes the sweep below.  On row shards the moment is the total: the shard of row 0 contributes it.
                // By size (the default) the choice is made again on the DEVICE, per iteration: the observation launch counted the
                // zero-weight vertices; with more than one in eight of them (open targets, partial overlap:
                // ClosestPointRegistrator.scala:84-91) the downdate launch and the right-hand-side sweep leave at once and the
                // weighted pass over the basis -- launched behind them, gated the other way -- does the work, as without the option.
                const bool gated = ctx->gram_downdate != 1;
                ZeroGate few{f->zero_counts, (int32_t)ceil_div(M, 256), 0, M}, many = few;
                many.run_if_many = 1;
                fa.gram_partial = gram_ws;
                fa.nslabs = launch_gram_downdate(ctx, m->Q0, M, rp, f->weight, gram_ws, gated ? &few : nullptr);
                fa.scaled_src = m->mom + MomentLayout{rp}.stot();
                fa.sigma2 = &f->st->sigma2;
                fa.scaled_contribute = m->row_begin == 0 ? 1 : 0;
                if (gated) {
                    bool alt_rhs = false;
                    fa.alt_nslabs = launch_gram(ctx, m->Q0, M, rp, f->weight, gram_ws, nullptr, f->evec, sweep_ws, &alt_rhs, &many);
                    fa.gate = many;
                    rhs_gate = few;
                }
            } else {
                fa.gram_partial = gram_ws;
                fa.nslabs = launch_gram(ctx, m->Q0, M, rp, f->weight, gram_ws, nullptr, f->evec, sweep_ws, &rhs_done);
            }
            fa.sweep_partial = sweep_ws;
            if (rhs_done) {  // the Gram pass left the right-hand-side partials, one row per slab
                fa.sweep_blocks = fa.nslabs;
            } else {
                SweepArgs a = base_args(f);
                a.evec = f->evec;
                a.partial = sweep_ws;
                a.no_reduce = 1;
                a.gate = rhs_gate;
                if (icp && !f->icp_surface && !f->reversed && f->n_lm == 0) {  // point-cloud ICP: observation + Q^T e in one pass
                    a.state = f->st;
                    a.icp_idx = f->nn_idx;
                    a.tx = tgt.x, a.ty = tgt.y, a.tz = tgt.z;
                    a.n_targets = tgt.n;
                    a.lm_mask = f->lm_mask;
                    a.weight_out = f->weight, a.evec_out = f->evec;
                    launch_sweep(ctx, SWEEP_RHS_ICP, a);
                } else {
                    launch_sweep(ctx, SWEEP_RHS, a);
                }
                fa.sweep_blocks = sweep_num_blocks(M);
            }
            launch_phase1_finalize(ctx, fa);
            launch_landmarks(ctx, m, f->st, f->n_lm, f->lm_pid, f->lm_xyz, f->lm_cov, Gw, rhsw);
            break;
        }
        case 2: {
            // the posterior mean of the uniform-weight case comes from the model's eigen-decomposition (no factorisation); a sampled
            // proposal needs the Cholesky factor itself (its square root of the covariance is part of the parity contract)
            const bool eig = icp && !f->icp_surface && !f->reversed && f->n_lm == 0 && !f->zrand_active && m->eig_ready;
            if (eig)
                launch_posterior_solve_eig(ctx, r, rp, m->eigV, m->eigL, &f->st->sigma2, rhs, f->acoef, f->st);
            else if (f->zrand_active && f->allow_alt && f->post_stage == 2 && f->nf_valid[f->live] && m->M == m->M_total && !f->partial_out)
                // the log-density query that first met this state left the factor of I + G and the posterior coefficients behind
                launch_posterior_sample_cached(ctx, r, rp, f->nfac[f->live], f->fxbuf[f->live] + (int64_t)rp * rp, f->zrand, f->acoef, f->st);
            else
                launch_posterior_solve(ctx, r, rp, G, rhs, f->zrand_active ? f->zrand : nullptr, f->work, f->acoef, f->st);
            launch_post_matvecs(ctx, m, f->alpha, f->acoef, f->zbuf);
            PostSolveArgs a;
            memset(&a, 0, sizeof(a));
            a.r = r;
            a.rp = rp;
            a.pvec = m->pvec;
            a.zbuf = f->zbuf;
            a.alpha = f->alpha;
            a.scalars = sc8;
            a.is_icp = icp ? 1 : 0;
            if (icp) {
                a.icp_step = (ip->initial_sigma - ip->end_sigma) / (double)ip->max_iterations;  // ICP.scala:65
                a.icp_end = ip->end_sigma;
            }
            a.step = f->step_length;
            a.global_transform = f->global_transform;
            a.state = f->st;
            a.retry = f->retry;
            a.zero_slot = f->absmax + 1;
            a.probabilistic = f->zrand_active ? 1 : 0;
            a.stop_threshold = f->stop_threshold;
            launch_post_solve(ctx, a);
            refresh_fit(f);
            break;
        }
        default:
            return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "phase %d out of range", phase);
    }
    return check_launch(ctx);
}

int check_ready(gingr_fitter *f) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    if (!f->target) return gingr_set_error(f->ctx, GINGR_ERR_STATE, "update: no target set (gingr_fitter_set_target)");
    if (!f->has_state) return gingr_set_error(f->ctx, GINGR_ERR_STATE, "update: no state set (gingr_fitter_set_state)");
    if (hipSetDevice(f->ctx->device) != hipSuccess) return gingr_set_error(f->ctx, GINGR_ERR_HIP, "hipSetDevice failed");
    return GINGR_OK;
}

}  // namespace

extern "C" {

int gingr_fitter_cpd_phase_async(gingr_fitter *f, const gingr_cpd_params *p, int32_t phase) {
    GINGR_TRY(check_ready(f));
    if (!p || !(p->w >= 0.0 && p->w < 1.0) || !(p->lambda > 0.0))
        return gingr_set_error(f->ctx, GINGR_ERR_BAD_ARGUMENT, "cpd params: need 0 <= w < 1 and lambda > 0");
    return run_phase(f, false, p, nullptr, phase);
}

int gingr_fitter_icp_phase_async(gingr_fitter *f, const gingr_icp_params *p, int32_t phase) {
    GINGR_TRY(check_ready(f));
    if (!p || p->max_iterations < 1) return gingr_set_error(f->ctx, GINGR_ERR_BAD_ARGUMENT, "icp params: max_iterations < 1");
    f->icp_surface = false;
    return run_phase(f, true, nullptr, p, phase);
}

int gingr_fitter_update_cpd_async(gingr_fitter *f, const gingr_cpd_params *p, int32_t n_iterations) {
    GINGR_TRY(check_ready(f));
    if (f->m->M != f->m->M_total)
        return gingr_set_error(f->ctx, GINGR_ERR_STATE, "update_cpd_async: sharded model needs the phase API + exchange");
    for (int32_t it = 0; it < n_iterations; ++it) {
        TimerScope ts(f->ctx, 3);
        for (int ph = 0; ph < GINGR_NUM_PHASES; ++ph) GINGR_TRY(gingr_fitter_cpd_phase_async(f, p, ph));
    }
    return GINGR_OK;
}

int gingr_fitter_update_icp_async(gingr_fitter *f, const gingr_icp_params *p, int32_t n_iterations) {
    GINGR_TRY(check_ready(f));
    if (f->m->M != f->m->M_total)
        return gingr_set_error(f->ctx, GINGR_ERR_STATE, "update_icp_async: sharded model needs the phase API + exchange");
    for (int32_t it = 0; it < n_iterations; ++it) {
        TimerScope ts(f->ctx, 3);
        for (int ph = 0; ph < GINGR_NUM_PHASES; ++ph) GINGR_TRY(gingr_fitter_icp_phase_async(f, p, ph));
    }
    return GINGR_OK;
}

}  // extern "C"

// One phase of flavour 0 CPD / 1 ICP point cloud / 2 ICP surface (GINGR_PHASE_GATHER included)
int fitter_run_phase(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int phase) {
    if (flavour == 0) return gingr_fitter_cpd_phase_async(f, cp, phase);
    if (flavour == 1) return gingr_fitter_icp_phase_async(f, ip, phase);
    return gingr_fitter_icp_surface_phase_async(f, ip, phase);
}

// The row-sharded update of any flavour, deterministic (z == nullptr) or with a sampled proposal (z: rank standard normals, one
// iteration): per iteration [surface: gather phase, all-reduce of the full fit], phase 0, [CPD: all-reduce of the column sums],
// phase 1, all-reduce of the Gram bundle, phase 2.  The posterior solve, the sample a + L^-T z and everything behind them are
// replicated r x r algebra, so z is the same on every shard and nothing else is exchanged.
// gather (nullable): does the whole gather of the fit itself (stage, all-gather, unpack: rccl_exchange.hip) and returns 0; a positive
// value means "not possible here" and the zero-padded all-reduce through `reduce` is used instead
static int gather_fit(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, gingr_allreduce_fn reduce, void *user,
                      fitter_gather_fn gather, const char *who) {
    gingr_ctx *ctx = f->ctx;
    if (gather) {
        const int g = gather(user, f);
        if (g == 0) return GINGR_OK;
        if (g < 0) return gingr_set_error(ctx, GINGR_ERR_STATE, "%s: the all-gather of the fit failed", who);
    }
    GINGR_TRY(fitter_run_phase(f, flavour, cp, ip, GINGR_PHASE_GATHER));
    if (reduce(user, GINGR_SEGMENT_FULLFIT, f->fullfit, 3 * f->m->M_total) != 0)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "%s: the all-reduce callback failed (full fit)", who);
    return GINGR_OK;
}

int fitter_sharded_update(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int32_t n_iterations,
                          const double *z, gingr_allreduce_fn reduce, void *user, fitter_gather_fn gather, bool split_native) {
    GINGR_TRY(check_ready(f));
    gingr_ctx *ctx = f->ctx;
    if (n_iterations < 0 || !reduce || flavour < 0 || flavour > 2) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "sharded update: bad arguments");
    if (f->partial_out) return gingr_set_error(ctx, GINGR_ERR_STATE, "sharded update: this fitter belongs to a device group");
    if (z && n_iterations != 1) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "sharded update: a sampled proposal is one iteration");
    if (z) GINGR_TRY(fitter_upload_zrand(f, z));
    f->zrand_active = z != nullptr;
    int rc = GINGR_OK;
    // GINGR_OPT_SPLIT_EXCHANGE: pass 1 in two halves of the target tiles; the all-reduce of the first half runs on the context's second
    // stream (ordered by events, same communicator) while the second half computes, so only the second half's all-reduce is exposed
    bool split = split_native && flavour == 0 && f->sharded() && f->N >= 8192;
    // (the halves take twice the chunks of the whole pass)
    if (split && (int64_t)2 * cpd_colsum_chunks(f->m->M, f->N) * f->N > f->ws_doubles) split = false;
    if (split && !ctx->side_stream) {
        if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->split_ev[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->split_ev[1], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            split = false;
        }
    }
    for (int32_t it = 0; it < n_iterations && rc == GINGR_OK; ++it) {
        TimerScope ts(ctx, 3);
        if ((flavour == 2 || (flavour == 1 && f->reversed)) && f->sharded()) rc = gather_fit(f, flavour, cp, ip, reduce, user, gather, "sharded update");
        for (int ph = 0; ph < GINGR_NUM_PHASES && rc == GINGR_OK; ++ph) {
            if (ph == 0 && split) {
                const int64_t NA = split_cut(f->N);
                double *seg0 = f->xch + f->off[0];
                f->split_half = 1;
                rc = fitter_run_phase(f, flavour, cp, ip, 0);
                {
                    TimerScope tx(ctx, 6);
                    if (!rc && (hipEventRecord(ctx->split_ev[0], ctx->stream) != hipSuccess ||
                                hipStreamWaitEvent(ctx->side_stream, ctx->split_ev[0], 0) != hipSuccess))
                        rc = gingr_set_error(ctx, GINGR_ERR_HIP, "sharded update: event ordering of the split exchange failed");
                    if (!rc) {
                        ctx->exchange_stream = ctx->side_stream;
                        const int xr = reduce(user, 0, seg0, NA);
                        ctx->exchange_stream = nullptr;
                        if (xr != 0) rc = gingr_set_error(ctx, GINGR_ERR_STATE, "sharded update: the all-reduce callback failed (segment 0, first half)");
                    }
                    if (!rc && hipEventRecord(ctx->split_ev[1], ctx->side_stream) != hipSuccess)
                        rc = gingr_set_error(ctx, GINGR_ERR_HIP, "sharded update: event ordering of the split exchange failed");
                }
                f->split_half = 2;
                if (!rc) rc = fitter_run_phase(f, flavour, cp, ip, 0);
                f->split_half = 0;
                {
                    TimerScope tx(ctx, 6);
                    if (!rc && reduce(user, 0, seg0 + NA, f->N - NA) != 0)
                        rc = gingr_set_error(ctx, GINGR_ERR_STATE, "sharded update: the all-reduce callback failed (segment 0, second half)");
                    if (!rc && hipStreamWaitEvent(ctx->stream, ctx->split_ev[1], 0) != hipSuccess)
                        rc = gingr_set_error(ctx, GINGR_ERR_HIP, "sharded update: event ordering of the split exchange failed");
                }
                continue;
            }
            rc = fitter_run_phase(f, flavour, cp, ip, ph);
            if (!rc && ph < GINGR_NUM_SEGMENTS && !(flavour != 0 && ph == 0)) {
                TimerScope tx(ctx, 6 + ph);  // the exchange of segment ph as this shard sees it (includes waiting for the peers)
                if (reduce(user, ph, f->xch + f->off[ph], f->cnt[ph]) != 0)
                    rc = gingr_set_error(ctx, GINGR_ERR_STATE, "sharded update: the all-reduce callback failed (segment %d)", ph);
            }
            if (!rc && ph == 0 && flavour != 0 && f->reversed && f->sharded() &&
                reduce(user, GINGR_SEGMENT_REVSUM, f->revsum, 4 * f->m->M_total) != 0)
                rc = gingr_set_error(ctx, GINGR_ERR_STATE, "sharded update: the all-reduce callback failed (reversal sums)");
        }
    }
    f->zrand_active = false;
    return rc;
}

extern "C" {

int gingr_fitter_update_cpd_sharded_async(gingr_fitter *f, const gingr_cpd_params *p, int32_t n_iterations, gingr_allreduce_fn reduce,
                                          void *user) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    return fitter_sharded_update(f, 0, p, nullptr, n_iterations, nullptr, reduce, user, nullptr, false);
}

int gingr_fitter_update_icp_sharded_async(gingr_fitter *f, const gingr_icp_params *p, int32_t n_iterations, gingr_allreduce_fn reduce,
                                          void *user) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    return fitter_sharded_update(f, 1, nullptr, p, n_iterations, nullptr, reduce, user, nullptr, false);
}

int gingr_fitter_update_sharded_async(gingr_fitter *f, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                      int32_t n_iterations, const double *z, gingr_allreduce_fn reduce, void *user) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    return fitter_sharded_update(f, flavour, cp, ip, n_iterations, z, reduce, user, nullptr, false);
}

int gingr_fitter_gather_stage(gingr_fitter *f, int32_t world, int32_t rank, void **send_ptr, void **recv_ptr, int64_t *count_per_rank) {
    if (!f || !send_ptr || !recv_ptr || !count_per_rank) return GINGR_ERR_BAD_ARGUMENT;
    GINGR_TRY(check_ready(f));
    gingr_ctx *ctx = f->ctx;
    const gingr_model *m = f->m;
    if (!f->fullfit) return gingr_set_error(ctx, GINGR_ERR_STATE, "gather_stage: not a row shard with meshes (gingr_fitter_set_meshes)");
    if (world < 1 || rank < 0 || rank >= world) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "gather_stage: need 0 <= rank < world");
    const int64_t Mt = m->M_total, base = Mt / world, extra = Mt % world;
    const int64_t b = rank * base + (rank < extra ? rank : extra), e = b + base + (rank < extra ? 1 : 0);
    if (b != m->row_begin || e - b != m->M)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "gather_stage: rows [%lld, %lld) are not shard %d of the balanced partition over %d shards",
                               (long long)m->row_begin, (long long)(m->row_begin + m->M), (int)rank, (int)world);
    const int64_t chunk = ceil_div(Mt, world);
    if (!f->gstage || f->gstage_world != world) {
        dev_free(f->gstage);
        f->gstage = nullptr;
        GINGR_TRY(dev_alloc(ctx, &f->gstage, (size_t)world * 3 * chunk));
        HIP_TRY(ctx, hipMemsetAsync(f->gstage, 0, (size_t)world * 3 * chunk * sizeof(double), ctx->stream));
        f->gstage_world = world;
    }
    double *mine = f->gstage + (int64_t)rank * 3 * chunk;
    hipLaunchKernelGGL(fit_to_stage_kernel, dim3((unsigned)ceil_div(m->M, 256)), dim3(256), 0, ctx->stream, f->fit, m->iperm, m->M, chunk, mine);
    *send_ptr = mine;
    *recv_ptr = f->gstage;
    *count_per_rank = 3 * chunk;
    return check_launch(ctx);
}

int gingr_fitter_gather_finish(gingr_fitter *f, int32_t world) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (!f->fullfit || !f->gstage || f->gstage_world != world)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "gather_finish: no gather staged for %d shards (gingr_fitter_gather_stage)", (int)world);
    const int64_t Mt = f->m->M_total;
    hipLaunchKernelGGL(stage_to_fullfit_kernel, dim3((unsigned)ceil_div(Mt, 256)), dim3(256), 0, ctx->stream, f->gstage, (int)world, ceil_div(Mt, world),
                       Mt, f->fullfit);
    return check_launch(ctx);
}

int gingr_fitter_reversal_exchange(gingr_fitter *f, void **dev_ptr, int64_t *count) {
    if (!f || !dev_ptr || !count) return GINGR_ERR_BAD_ARGUMENT;
    if (!f->revsum)
        return gingr_set_error(f->ctx, GINGR_ERR_STATE, "reversal_exchange: not a row shard with the reversed direction set (gingr_fitter_set_correspondence_direction)");
    *dev_ptr = f->revsum;
    *count = 4 * f->m->M_total;
    return GINGR_OK;
}

int gingr_fitter_fullfit_exchange(gingr_fitter *f, void **dev_ptr, int64_t *count) {
    if (!f || !dev_ptr || !count) return GINGR_ERR_BAD_ARGUMENT;
    if (!f->fullfit) return gingr_set_error(f->ctx, GINGR_ERR_STATE, "fullfit_exchange: not a row shard with meshes (gingr_fitter_set_meshes)");
    *dev_ptr = f->fullfit;
    *count = 3 * f->m->M_total;
    return GINGR_OK;
}

// ------------------------------------------------------------------------------------------ ICP, surface correspondence
int gingr_fitter_set_meshes(gingr_fitter *f, int64_t n_model_tri, const int32_t *model_tri, int64_t n_target_tri,
                            const int32_t *target_tri) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    f->forget_posteriors();  // the posterior memos describe other inputs
    gingr_ctx *ctx = f->ctx;
    if (!f->target) return gingr_set_error(ctx, GINGR_ERR_STATE, "set_meshes: no target set (gingr_fitter_set_target)");
    if (n_model_tri < 1 || n_target_tri < 1 || !model_tri || !target_tri)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "set_meshes: need at least one triangle per mesh");
    // A row shard takes the triangles of the WHOLE template (vertex ids of the full model): its queries are its own rows, but the
    // tests against the template itself -- vertex normals, self-intersection -- see all of it, through the gathered fit
    // (gingr_fitter::fullfit, original point order).  A single shard indexes its own fit (device order).
    const bool sharded = f->sharded();
    const int64_t M = f->m->M, N = f->N, Mt = f->m->M_total, rb = f->m->row_begin;
    for (int64_t k = 0; k < 3 * n_model_tri; ++k)
        if (model_tri[k] < 0 || model_tri[k] >= Mt) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "set_meshes: model vertex id out of range");
    for (int64_t k = 0; k < 3 * n_target_tri; ++k)
        if (target_tri[k] < 0 || target_tri[k] >= N) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "set_meshes: target vertex id out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_meshes(f);
    // device vertex positions of the two clouds (orig -> device) and their coordinates in device order
    std::vector<int32_t> tinv((size_t)N);
    for (int64_t s2 = 0; s2 < N; ++s2) tinv[(size_t)f->h_tperm[(size_t)s2]] = (int32_t)s2;
    std::vector<double> mpos((size_t)3 * M), mmean((size_t)3 * M), tpos((size_t)3 * N);
    HIP_TRY(ctx, hipMemcpy(mpos.data(), f->m->ref, mpos.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(mmean.data(), f->m->mean, mmean.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(tpos.data(), f->target, tpos.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < mpos.size(); ++k) mpos[k] += mmean[k];
    struct Built {
        std::vector<int32_t> tri, orig, adj_ptr, adj_tri;
    };
    // coord(v, d): coordinate d of vertex v; mesh_pos(v): its position in the cloud the triangles index; adj_slot(v): its slot in the
    // vertex -> triangles lists (n_adj slots) or -1 for a vertex this fitter does not own
    auto build = [&](int64_t T, const int32_t *tri, auto coord, auto mesh_pos, auto adj_slot, int64_t n_adj, Built &b) {
        std::vector<double> cen((size_t)3 * T);
        for (int64_t t = 0; t < T; ++t)
            for (int d = 0; d < 3; ++d) {
                double c = 0.0;
                for (int k = 0; k < 3; ++k) c += coord(tri[3 * t + k], d);
                cen[(size_t)3 * t + d] = c / 3.0;
            }
        morton_order(cen.data(), T, b.orig);  // orig[s] = original index of the triangle at device position s
        std::vector<int32_t> tpos2((size_t)T);
        b.tri.resize((size_t)3 * T);
        for (int64_t s2 = 0; s2 < T; ++s2) {
            const int32_t t = b.orig[(size_t)s2];
            tpos2[(size_t)t] = (int32_t)s2;
            for (int k = 0; k < 3; ++k) b.tri[(size_t)3 * s2 + k] = mesh_pos(tri[3 * t + k]);
        }
        // vertex -> triangles, in ascending ORIGINAL triangle index (the order the normals are averaged in)
        std::vector<int32_t> cnt((size_t)n_adj + 1, 0);
        int64_t total = 0;
        for (int64_t k = 0; k < 3 * T; ++k) {
            const int32_t sl = adj_slot(tri[k]);
            if (sl >= 0) cnt[(size_t)sl + 1]++, ++total;
        }
        for (int64_t v = 0; v < n_adj; ++v) cnt[(size_t)v + 1] += cnt[(size_t)v];
        b.adj_ptr = cnt;
        b.adj_tri.resize((size_t)(total > 0 ? total : 1));
        std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
        for (int64_t t = 0; t < T; ++t)
            for (int k = 0; k < 3; ++k) {
                const int32_t sl = adj_slot(tri[3 * t + k]);
                if (sl >= 0) b.adj_tri[(size_t)fill[(size_t)sl]++] = tpos2[(size_t)t];
            }
    };
    Built bm, bt;
    const std::vector<int32_t> &hip = f->m->hiperm;
    if (sharded) {
        const std::vector<double> &full = f->m->h_full_pts;
        if ((int64_t)full.size() != 3 * Mt) return gingr_set_error(ctx, GINGR_ERR_STATE, "set_meshes: the shard holds no copy of the full mean shape");
        build(n_model_tri, model_tri, [&](int32_t v, int d) { return full[(size_t)3 * v + d]; }, [&](int32_t v) { return v; },
              [&](int32_t v) { return (v >= rb && v < rb + M) ? hip[(size_t)(v - rb)] : -1; }, M, bm);
    } else {
        build(n_model_tri, model_tri, [&](int32_t v, int d) { return mpos[(size_t)d * M + hip[(size_t)v]]; }, [&](int32_t v) { return hip[(size_t)v]; },
              [&](int32_t v) { return hip[(size_t)v]; }, M, bm);
    }
    build(n_target_tri, target_tri, [&](int32_t v, int d) { return tpos[(size_t)d * N + tinv[(size_t)v]]; }, [&](int32_t v) { return tinv[(size_t)v]; },
          [&](int32_t v) { return tinv[(size_t)v]; }, N, bt);
    // target boundary vertices: on an edge with exactly one adjacent triangle (TriangleMesh3DOperations.pointIsOnBoundary)
    std::vector<int32_t> bnd((size_t)N, 0);
    {
        std::vector<std::pair<uint64_t, int>> edges;
        edges.reserve((size_t)3 * n_target_tri);
        for (int64_t t = 0; t < n_target_tri; ++t)
            for (int k = 0; k < 3; ++k) {
                const uint64_t a = (uint64_t)target_tri[3 * t + k], b = (uint64_t)target_tri[3 * t + (k + 1) % 3];
                edges.push_back({(a < b ? a : b) << 32 | (a < b ? b : a), 0});
            }
        std::sort(edges.begin(), edges.end());
        for (size_t i = 0; i < edges.size();) {
            size_t j = i;
            while (j < edges.size() && edges[j].first == edges[i].first) ++j;
            if (j - i == 1) {
                bnd[(size_t)tinv[(size_t)(edges[i].first >> 32)]] = 1;
                bnd[(size_t)tinv[(size_t)(edges[i].first & 0xffffffffu)]] = 1;
            }
            i = j;
        }
    }
    // model boundary vertices (reversed direction: the rejection rules run on the template side)
    std::vector<int32_t> mbnd((size_t)M, 0);
    if (!sharded) {
        std::vector<uint64_t> edges;
        edges.reserve((size_t)3 * n_model_tri);
        for (int64_t t = 0; t < n_model_tri; ++t)
            for (int k = 0; k < 3; ++k) {
                const uint64_t a = (uint64_t)model_tri[3 * t + k], b = (uint64_t)model_tri[3 * t + (k + 1) % 3];
                edges.push_back((a < b ? a : b) << 32 | (a < b ? b : a));
            }
        std::sort(edges.begin(), edges.end());
        for (size_t i = 0; i < edges.size();) {
            size_t j = i;
            while (j < edges.size() && edges[j] == edges[i]) ++j;
            if (j - i == 1) {
                mbnd[(size_t)f->m->hiperm[(size_t)(edges[i] >> 32)]] = 1;
                mbnd[(size_t)f->m->hiperm[(size_t)(edges[i] & 0xffffffffu)]] = 1;
            }
            i = j;
        }
    }
    // (row shard) the whole template's vertex -> triangle lists and boundary flags in original vertex order: the reversed
    // correspondence direction tests the template vertex nearest to a match, which may belong to any shard
    Built bfull;
    std::vector<int32_t> mbnd_full;
    if (sharded) {
        const std::vector<double> &full = f->m->h_full_pts;
        build(n_model_tri, model_tri, [&](int32_t v, int d) { return full[(size_t)3 * v + d]; }, [&](int32_t v) { return v; },
              [&](int32_t v) { return v; }, Mt, bfull);
        mbnd_full.assign((size_t)Mt, 0);
        std::vector<uint64_t> edges;
        edges.reserve((size_t)3 * n_model_tri);
        for (int64_t t = 0; t < n_model_tri; ++t)
            for (int k = 0; k < 3; ++k) {
                const uint64_t a = (uint64_t)model_tri[3 * t + k], b = (uint64_t)model_tri[3 * t + (k + 1) % 3];
                edges.push_back((a < b ? a : b) << 32 | (a < b ? b : a));
            }
        std::sort(edges.begin(), edges.end());
        for (size_t i = 0; i < edges.size();) {
            size_t j = i;
            while (j < edges.size() && edges[j] == edges[i]) ++j;
            if (j - i == 1) mbnd_full[(size_t)(edges[i] >> 32)] = 1, mbnd_full[(size_t)(edges[i] & 0xffffffffu)] = 1;
            i = j;
        }
    }
    f->Tm = n_model_tri;
    f->Tt = n_target_tri;
    const int64_t ntm = ceil_div(f->Tm, 256), ntt = ceil_div(f->Tt, 256);
    int rc;
    if ((rc = dev_alloc(ctx, &f->mtri, (size_t)3 * f->Tm)) || (rc = dev_alloc(ctx, &f->ttri, (size_t)3 * f->Tt)) ||
        (rc = dev_alloc(ctx, &f->ttri_orig, (size_t)f->Tt)) || (rc = dev_alloc(ctx, &f->madj_ptr, (size_t)M + 1)) ||
        (rc = dev_alloc(ctx, &f->madj_tri, bm.adj_tri.size())) || (rc = dev_alloc(ctx, &f->tadj_ptr, (size_t)N + 1)) ||
        (rc = dev_alloc(ctx, &f->tadj_tri, (size_t)3 * f->Tt)) || (rc = dev_alloc(ctx, &f->mcn, (size_t)3 * f->Tm)) ||
        (rc = dev_alloc(ctx, &f->tcn, (size_t)3 * f->Tt)) || (rc = dev_alloc(ctx, &f->mvn, (size_t)3 * M)) ||
        (rc = dev_alloc(ctx, &f->tvn, (size_t)3 * N)) || (rc = dev_alloc(ctx, &f->mtboxes, (size_t)30 * ntm + 6 * (ntm / 16 + 1))) ||
        (rc = dev_alloc(ctx, &f->ttboxes, (size_t)30 * ntt + 6 * (ntt / 16 + 1))) || (rc = dev_alloc(ctx, &f->tboundary, (size_t)N)) ||
        (rc = dev_alloc(ctx, &f->surf_cp, (size_t)3 * M)) || (rc = dev_alloc(ctx, &f->surf_d2, (size_t)M)) ||
        (rc = dev_alloc(ctx, &f->surf_w01, (size_t)M)) || (rc = dev_alloc(ctx, &f->surf_win, (size_t)M)) ||
        (rc = dev_alloc(ctx, &f->surf_nnd2, (size_t)M)) || (rc = dev_alloc(ctx, &f->surf_nn, (size_t)M)) ||
        (rc = dev_alloc(ctx, &f->surf_pre, (size_t)M)) || (rc = dev_alloc(ctx, &f->surf_hit, (size_t)M)) ||
        (rc = dev_alloc(ctx, &f->surf_tri_pos, (size_t)M)) || (rc = dev_alloc(ctx, &f->mtribox, (size_t)6 * f->Tm)) ||
        (rc = dev_alloc(ctx, &f->ttribox, (size_t)6 * f->Tt)) ||
        (rc = dev_alloc(ctx, &f->mtri_orig, (size_t)f->Tm)) || (rc = dev_alloc(ctx, &f->mboundary, (size_t)M)))
        return rc;
    auto up = [&](int32_t *dst, const std::vector<int32_t> &src) {
        return hipMemcpy(dst, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    };
    HIP_TRY(ctx, up(f->mtri, bm.tri));
    HIP_TRY(ctx, up(f->ttri, bt.tri));
    HIP_TRY(ctx, up(f->ttri_orig, bt.orig));
    HIP_TRY(ctx, up(f->madj_ptr, bm.adj_ptr));
    HIP_TRY(ctx, up(f->madj_tri, bm.adj_tri));
    HIP_TRY(ctx, up(f->tadj_ptr, bt.adj_ptr));
    HIP_TRY(ctx, up(f->tadj_tri, bt.adj_tri));
    HIP_TRY(ctx, up(f->tboundary, bnd));
    HIP_TRY(ctx, up(f->mtri_orig, bm.orig));
    HIP_TRY(ctx, up(f->mboundary, mbnd));
    if (sharded && !f->fullfit) {
        GINGR_TRY(dev_alloc(ctx, &f->fullfit, (size_t)3 * Mt));
        HIP_TRY(ctx, hipMemsetAsync(f->fullfit, 0, (size_t)3 * Mt * sizeof(double), ctx->stream));
    }
    if (sharded) {
        if ((rc = dev_alloc(ctx, &f->radj_ptr, (size_t)Mt + 1)) || (rc = dev_alloc(ctx, &f->radj_tri, bfull.adj_tri.size())) ||
            (rc = dev_alloc(ctx, &f->rmbnd, (size_t)Mt)) || (rc = dev_alloc(ctx, &f->rmvn, (size_t)3 * Mt)) ||
            (rc = dev_alloc(ctx, &f->rfboxes, (size_t)ceil_div(Mt, 256) * 30)))
            return rc;
        HIP_TRY(ctx, up(f->radj_ptr, bfull.adj_ptr));
        HIP_TRY(ctx, up(f->radj_tri, bfull.adj_tri));
        HIP_TRY(ctx, up(f->rmbnd, mbnd_full));
    }
    // static target side: cell normals, vertex normals, triangle tile boxes
    const Cloud tgt = cloud_of(f->target, N);
    launch_cell_normals(ctx, tgt, f->ttri, f->Tt, f->tcn);
    launch_vertex_normals(ctx, f->tadj_ptr, f->tadj_tri, f->tcn, f->Tt, N, f->tvn);
    launch_tri_tile_bbox(ctx, tgt, f->ttri, f->Tt, f->ttboxes, f->ttribox);
    GINGR_TRY(check_launch(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // the target triangles do not move: bin them once (the surface ICP's warm-started closest-point search)
    GINGR_TRY(tri_grid_build(ctx, tpos.data(), N, bt.tri.data(), bt.orig.data(), n_target_tri, M, &f->ttgrid));
    // the template's triangles move: their grid is rebuilt on the device every iteration (self-intersection test); buffers only here
    GINGR_TRY(mov_grid_alloc(ctx, n_model_tri, M, &f->mgrid));
    return GINGR_OK;
}

int gingr_fitter_set_surface_method(gingr_fitter *f, int32_t method) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    if (method != 0 && method != 1) return gingr_set_error(f->ctx, GINGR_ERR_BAD_ARGUMENT, "set_surface_method: 0 or 1");
    f->surface_method = method;
    return GINGR_OK;
}

int gingr_fitter_set_correspondence_direction(gingr_fitter *f, int32_t reversed) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (!reversed) {
        f->reversed = false;
        return GINGR_OK;
    }
    if (!f->target) return gingr_set_error(ctx, GINGR_ERR_STATE, "set_correspondence_direction: no target set");
    if (f->sharded() && !f->radj_ptr)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "set_correspondence_direction: a row shard needs the meshes first (gingr_fitter_set_meshes: "
                                                     "the reversed direction works on the gathered template)");
    if (f->sharded() && !f->revsum) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        const int64_t Mt = f->m->M_total, N = f->N;
        // this shard's range of the target queries: the same fraction of the (replicated) target as its rows are of the template --
        // the row ranges tile [0, M_total), so the query ranges tile [0, N), whatever the number of shards
        const int64_t q0 = (int64_t)((__int128)N * f->m->row_begin / Mt), q1 = (int64_t)((__int128)N * (f->m->row_begin + f->m->M) / Mt);
        f->rq0 = q0;
        f->rqn = q1 - q0;
        int rc;
        if ((rc = dev_alloc(ctx, &f->revsum, (size_t)4 * Mt)) || (rc = dev_alloc(ctx, &f->rtvn_loc, (size_t)3 * (f->rqn > 0 ? f->rqn : 1)))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(f->revsum, 0, (size_t)4 * Mt * sizeof(double), ctx->stream));
        if (f->tvn && f->rqn > 0)
            hipLaunchKernelGGL(soa_range_kernel, dim3((unsigned)ceil_div(f->rqn, 256)), dim3(256), 0, ctx->stream, f->tvn, N, q0, f->rqn, f->rtvn_loc);
        HIP_TRY(ctx, hipMalloc(&f->rws, (size_t)nn_ws_bytes(f->N, Mt)));
    }
    if (!f->rnn) {  // buffers per target vertex + the sort workspace, once per target
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        const int64_t M = f->m->M, N = f->N;
        int rc;
        if ((rc = dev_alloc(ctx, &f->rcp, (size_t)3 * N)) || (rc = dev_alloc(ctx, &f->rd2, (size_t)N)) ||
            (rc = dev_alloc(ctx, &f->rnnd2, (size_t)N)) || (rc = dev_alloc(ctx, &f->rw01, (size_t)N)) ||
            (rc = dev_alloc(ctx, &f->robs, (size_t)3 * M)) || (rc = dev_alloc(ctx, &f->rwin, (size_t)M)) ||
            (rc = dev_alloc(ctx, &f->rnn, (size_t)N)) || (rc = dev_alloc(ctx, &f->rpre, (size_t)N)) ||
            (rc = dev_alloc(ctx, &f->rhit, (size_t)N)) || (rc = dev_alloc(ctx, &f->rkeys, (size_t)N)) ||
            (rc = dev_alloc(ctx, &f->rvals, (size_t)N)) || (rc = dev_alloc(ctx, &f->rskeys, (size_t)N)) ||
            (rc = dev_alloc(ctx, &f->rsvals, (size_t)N)) || (rc = dev_alloc(ctx, &f->rtri_pos, (size_t)N)))
            return rc;
        f->rtri_warm = false;
        f->rsort_bytes = reversal_sort_temp_bytes(N);
        HIP_TRY(ctx, hipMalloc(&f->rsort, f->rsort_bytes ? f->rsort_bytes : 8));
    }
    f->reversed = true;
    return GINGR_OK;
}

int gingr_fitter_get_reversed_correspondence(gingr_fitter *f, int32_t *template_id, double *w) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (f->corr_stale)  // see gingr_fitter::alt_seg
        return gingr_set_error(ctx, GINGR_ERR_STATE, "get_reversed_correspondence: a probabilistic query brought another state's posterior back; the correspondences on the device are not this state's -- run an update or a phase first");
    if (!f->rnn) return gingr_set_error(ctx, GINGR_ERR_STATE, "get_reversed_correspondence: direction not reversed");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t N = f->N;
    // a row shard scanned ITS range of the target queries only (device positions [rq0, rq0 + rqn)): the others come back as
    // (-1, 0) -- the shards' answers are disjoint and together cover the target
    const int64_t q0 = f->sharded() ? f->rq0 : 0, nq = f->sharded() ? f->rqn : N;
    std::vector<int32_t> hid((size_t)(nq > 0 ? nq : 1));
    std::vector<double> hw((size_t)(nq > 0 ? nq : 1));
    if (nq > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(hid.data(), f->rnn, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(hw.data(), f->rw01 + q0, (size_t)nq * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t s2 = 0; s2 < N; ++s2) {  // device target position -> original target id; device model row -> original vertex id
        const int32_t j = f->h_tperm[(size_t)s2];
        const bool mine = s2 >= q0 && s2 < q0 + nq;
        const int32_t row = mine ? hid[(size_t)(s2 - q0)] : -1;
        if (template_id)  // (a row shard searched the gathered template: original vertex ids already)
            template_id[j] = f->sharded() ? ((row >= 0 && row < f->m->M_total) ? row : -1)
                                          : ((row >= 0 && row < f->m->M) ? f->m->hperm[(size_t)row] : -1);
        if (w) w[j] = mine ? hw[(size_t)(s2 - q0)] : 0.0;
    }
    return GINGR_OK;
}

int gingr_fitter_icp_surface_phase_async(gingr_fitter *f, const gingr_icp_params *p, int32_t phase) {
    GINGR_TRY(check_ready(f));
    if (!p || p->max_iterations < 1) return gingr_set_error(f->ctx, GINGR_ERR_BAD_ARGUMENT, "icp params: max_iterations < 1");
    if (!f->Tm || !f->Tt) return gingr_set_error(f->ctx, GINGR_ERR_STATE, "icp surface: no meshes set (gingr_fitter_set_meshes)");
    f->icp_surface = true;
    return run_phase(f, true, nullptr, p, phase);
}

int gingr_fitter_update_icp_surface_async(gingr_fitter *f, const gingr_icp_params *p, int32_t n_iterations) {
    GINGR_TRY(check_ready(f));
    if (f->sharded())
        return gingr_set_error(f->ctx, GINGR_ERR_STATE, "update_icp_surface_async: a row shard needs the sharded update (gingr_fitter_update_sharded_async / _rccl_async / the device group)");
    for (int32_t it = 0; it < n_iterations; ++it) {
        TimerScope ts(f->ctx, 3);
        for (int ph = 0; ph < GINGR_NUM_PHASES; ++ph) GINGR_TRY(gingr_fitter_icp_surface_phase_async(f, p, ph));
    }
    return GINGR_OK;
}

int gingr_fitter_get_surface_correspondence(gingr_fitter *f, double *cp_xyz, double *w) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    gingr_ctx *ctx = f->ctx;
    if (f->corr_stale)  // see gingr_fitter::alt_seg
        return gingr_set_error(ctx, GINGR_ERR_STATE, "get_surface_correspondence: a probabilistic query brought another state's posterior back; the correspondences on the device are not this state's -- run an update or a phase first");
    if (!f->surf_cp) return gingr_set_error(ctx, GINGR_ERR_STATE, "get_surface_correspondence: no meshes set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = f->m->M;
    if (cp_xyz) {
        launch_soa_to_aos(ctx, f->surf_cp, M, reinterpret_cast<double *>(f->aos), f->m->perm);
        HIP_TRY(ctx, hipMemcpyAsync(cp_xyz, f->aos, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    std::vector<double> hw((size_t)M);
    HIP_TRY(ctx, hipMemcpyAsync(hw.data(), f->surf_w01, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (w)
        for (int64_t s2 = 0; s2 < M; ++s2) w[f->m->hperm[(size_t)s2]] = hw[(size_t)s2];
    return GINGR_OK;
}

// ------------------------------------------------------------------------------------------ probabilistic proposal
// correspondence flavour of a probabilistic query: 0 CPD, 1 ICP point cloud, 2 ICP surface
static int flavour_phase(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, int ph) {
    if (flavour == 0) return gingr_fitter_cpd_phase_async(f, cp, ph);
    if (flavour == 1) return gingr_fitter_icp_phase_async(f, ip, ph);
    return gingr_fitter_icp_surface_phase_async(f, ip, ph);
}

static int sample_update(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, const double *z) {
    GINGR_TRY(check_ready(f));
    gingr_ctx *ctx = f->ctx;
    if (!z) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "update_sample: z is null");
    if (f->m->M != f->m->M_total) return gingr_set_error(ctx, GINGR_ERR_STATE, "update_sample: single shard only");
    // (this entry point returns without synchronising: the draws go through the fitter's own event-guarded pinned buffer, not
    // through `pin`, which the synchronous entry points rewrite)
    GINGR_TRY(fitter_upload_zrand(f, z));
    f->zrand_active = true;
    int rc = GINGR_OK;
    f->allow_alt = true;
    for (int ph = 0; ph < GINGR_NUM_PHASES && rc == GINGR_OK; ++ph) rc = flavour_phase(f, flavour, cp, ip, ph);
    f->allow_alt = false;
    f->zrand_active = false;
    return rc;
}

int gingr_fitter_update_cpd_sample_async(gingr_fitter *f, const gingr_cpd_params *p, const double *z) {
    return sample_update(f, 0, p, nullptr, z);
}

int gingr_fitter_update_icp_sample_async(gingr_fitter *f, const gingr_icp_params *p, const double *z) {
    return sample_update(f, 1, nullptr, p, z);
}

int gingr_fitter_update_icp_surface_sample_async(gingr_fitter *f, const gingr_icp_params *p, const double *z) {
    return sample_update(f, 2, nullptr, p, z);
}

static int posterior_logpdf(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                            const double *mesh_xyz, double *logpdf) {
    GINGR_TRY(check_ready(f));
    gingr_ctx *ctx = f->ctx;
    const gingr_model *m = f->m;
    if (!mesh_xyz || !logpdf) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "posterior_logpdf: null argument");
    if (m->M != m->M_total) return gingr_set_error(ctx, GINGR_ERR_STATE, "posterior_logpdf: single shard only");
    const int64_t M = m->M;
    const int32_t r = m->r, rp = m->rp;
    // posterior of the current state: correspondences, Gram, right-hand side (phases 0 and 1 do not touch the state)
    f->allow_alt = true;
    int prc = GINGR_OK;
    for (int ph = 0; ph < 2 && prc == GINGR_OK; ++ph) prc = flavour_phase(f, flavour, cp, ip, ph);
    f->allow_alt = false;
    GINGR_TRY(prc);
    if (f->lp_epoch == 0) HIP_TRY(ctx, hipMemsetAsync(f->lp_sync, 0, 2 * sizeof(unsigned), ctx->stream));  // before the first hand-over
    const bool cached = f->fx_valid[f->live];  // this state's factors are on the device: only the mesh-dependent part is left
    double *G = f->seg1_live();
    double *rhs = G + (int64_t)rp * rp;
    // Q0^T e with e = R^T(mesh - c - t) - (ref - c) - mean in the pose of the state (copied on the device, no host round trip)
    double *out2 = f->small;
    double *aos = reinterpret_cast<double *>(f->aos);
    memcpy(f->pin, mesh_xyz, (size_t)3 * M * sizeof(double));  // pinned: the copy is a plain asynchronous DMA
    HIP_TRY(ctx, hipMemcpyAsync(aos, f->pin, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, aos, M, f->newshape, m->perm);
    hipLaunchKernelGGL(pose_of_state_kernel, dim3(1), dim3(64), 0, ctx->stream, f->st, f->pose);
    SweepArgs a = base_args(f);
    a.shape_in = f->newshape;
    a.out = f->alpha_c;
    launch_sweep(ctx, SWEEP_PROJ2, a);
    // one kernel: posterior coefficients a = (I + G)^-1 rhs, then the ridge projection of the mesh and its log-density
    GINGR_TRY(launch_posterior_logpdf(ctx, r, rp, G, rhs, m->mom + MomentLayout{rp}.stot(), f->alpha_c, f->fxbuf[f->live], cached, f->work,
                                      out2, f->lp_sync, ++f->lp_epoch));
    GINGR_TRY(check_launch(ctx));
    double *res = f->pin + (size_t)3 * M;  // behind the mesh (pin holds 3M + rp + ... doubles)
    GINGR_TRY(pull_small(f, out2, 2, res));
    if (res[1] != 0.0) return gingr_set_error(ctx, GINGR_ERR_NOT_SPD, "posterior_logpdf: posterior of the current state failed");
    if (!std::isfinite(res[0])) return gingr_set_error(ctx, GINGR_ERR_NONFINITE, "posterior_logpdf: non-finite result");
    if (f->post_stage == 2) f->fx_valid[f->live] = true;  // (not memoised: sharded / state unknown to the host -> nothing to key it by)
    *logpdf = res[0];
    return GINGR_OK;
}

int gingr_fitter_posterior_logpdf_cpd(gingr_fitter *f, const gingr_cpd_params *p, const double *mesh_xyz, double *logpdf) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    if (!p || !(p->w >= 0.0 && p->w < 1.0) || !(p->lambda > 0.0))
        return gingr_set_error(f->ctx, GINGR_ERR_BAD_ARGUMENT, "cpd params: need 0 <= w < 1 and lambda > 0");
    return posterior_logpdf(f, 0, p, nullptr, mesh_xyz, logpdf);
}

int gingr_fitter_posterior_logpdf_icp(gingr_fitter *f, const gingr_icp_params *p, const double *mesh_xyz, double *logpdf) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    if (!p || p->max_iterations < 1) return gingr_set_error(f->ctx, GINGR_ERR_BAD_ARGUMENT, "icp params: max_iterations < 1");
    return posterior_logpdf(f, 1, nullptr, p, mesh_xyz, logpdf);
}

int gingr_fitter_posterior_logpdf_icp_surface(gingr_fitter *f, const gingr_icp_params *p, const double *mesh_xyz, double *logpdf) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    if (!p || p->max_iterations < 1) return gingr_set_error(f->ctx, GINGR_ERR_BAD_ARGUMENT, "icp params: max_iterations < 1");
    return posterior_logpdf(f, 2, nullptr, p, mesh_xyz, logpdf);
}

}  // extern "C"

// ---- transition density on a row shard: the two halves around the exchange of segment 1 (the device group drives them itself)
// prepare: this shard's rows of the mesh (host, the FULL mesh in the caller's point order) -> e = R^T (mesh - c - t) - (ref - c) - mean
// in the pose of the state -> the partial Q0^T e into the tail of exchange segment 1 (summed with the Gram bundle).
int fitter_logpdf_prepare(gingr_fitter *f, const double *mesh_xyz_full) {
    gingr_ctx *ctx = f->ctx;
    const gingr_model *m = f->m;
    const int64_t M = m->M;
    const int32_t rp = m->rp;
    double *aos = reinterpret_cast<double *>(f->aos);
    memcpy(f->pin, mesh_xyz_full + 3 * m->row_begin, (size_t)3 * M * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(aos, f->pin, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_aos_to_soa(ctx, aos, M, f->newshape, m->perm);
    hipLaunchKernelGGL(pose_of_state_kernel, dim3(1), dim3(64), 0, ctx->stream, f->st, f->pose);
    SweepArgs a = base_args(f);
    a.shape_in = f->newshape;
    a.out = (f->partial_out ? f->partial_out : f->xch) + f->off[1] + (int64_t)rp * rp + rp + 8;
    launch_sweep(ctx, SWEEP_PROJ2, a);
    return check_launch(ctx);
}

// finish (segment 1 reduced): the replicated log-density kernel, read-back, and the tail of segment 1 back to zero
int fitter_logpdf_finish(gingr_fitter *f, double *logpdf) {
    gingr_ctx *ctx = f->ctx;
    const gingr_model *m = f->m;
    const int32_t r = m->r, rp = m->rp;
    double *G = f->xch + f->off[1];
    double *rhs = G + (int64_t)rp * rp, *qte = rhs + rp + 8;
    double *fx = nullptr;
    unsigned *sync = nullptr;
    unsigned epoch = 0;
    if (rp >= 128) {  // the two factorisations side by side on the super-panel solve (gp.hip: posterior_logpdf_wide_kernel)
        if (!f->lp_scratch) GINGR_TRY(dev_alloc(ctx, &f->lp_scratch, (size_t)rp * rp + 2 * rp));
        if (f->lp_epoch == 0) HIP_TRY(ctx, hipMemsetAsync(f->lp_sync, 0, 2 * sizeof(unsigned), ctx->stream));  // before the first hand-over
        fx = f->lp_scratch, sync = f->lp_sync, epoch = ++f->lp_epoch;
    }
    GINGR_TRY(launch_posterior_logpdf(ctx, r, rp, G, rhs, m->mom + MomentLayout{rp}.stot(), qte, fx, false, f->work, f->small, sync, epoch));
    GINGR_TRY(check_launch(ctx));
    double *res = f->pin + (size_t)3 * m->M;
    // in-place exchanges (RCCL, host callback) would keep adding a stale tail up, so it goes back to zero.  NOT the device group's send
    // buffer: a slower peer may still be reading it (double buffering protects the next WRITE, two exchanges later, not a write now);
    // its stale partial is harmless -- the group's sum is out of place, and updates never read the tail.
    if (!f->partial_out) HIP_TRY(ctx, hipMemsetAsync(qte, 0, (size_t)rp * sizeof(double), ctx->stream));
    GINGR_TRY(pull_small(f, f->small, 2, res));
    if (res[1] != 0.0) return gingr_set_error(ctx, GINGR_ERR_NOT_SPD, "posterior_logpdf: posterior of the current state failed");
    if (!std::isfinite(res[0])) return gingr_set_error(ctx, GINGR_ERR_NONFINITE, "posterior_logpdf: non-finite result");
    *logpdf = res[0];
    return GINGR_OK;
}

// posterior(of the current state).gp.logpdf(posterior.coefficients(mesh)) on a row shard (GeneratorWrapperStochastic.scala:42-63):
// phases 0 and 1 with their exchanges; Q0^T e rides in segment 1; the log-density kernel is replicated.
int fitter_sharded_logpdf(gingr_fitter *f, int flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip, const double *mesh_xyz_full,
                          gingr_allreduce_fn reduce, void *user, double *logpdf, fitter_gather_fn gather) {
    GINGR_TRY(check_ready(f));
    gingr_ctx *ctx = f->ctx;
    if (!mesh_xyz_full || !logpdf || !reduce || flavour < 0 || flavour > 2)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "sharded posterior_logpdf: bad arguments");
    if (f->partial_out) return gingr_set_error(ctx, GINGR_ERR_STATE, "sharded posterior_logpdf: this fitter belongs to a device group");
    if ((flavour == 2 || (flavour == 1 && f->reversed)) && f->sharded()) GINGR_TRY(gather_fit(f, flavour, cp, ip, reduce, user, gather, "sharded posterior_logpdf"));
    GINGR_TRY(fitter_run_phase(f, flavour, cp, ip, 0));
    if (flavour == 0 && reduce(user, 0, f->xch + f->off[0], f->cnt[0]) != 0)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "sharded posterior_logpdf: the all-reduce callback failed (segment 0)");
    if (flavour != 0 && f->reversed && f->sharded() && reduce(user, GINGR_SEGMENT_REVSUM, f->revsum, 4 * f->m->M_total) != 0)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "sharded posterior_logpdf: the all-reduce callback failed (reversal sums)");
    GINGR_TRY(fitter_run_phase(f, flavour, cp, ip, 1));
    GINGR_TRY(fitter_logpdf_prepare(f, mesh_xyz_full));
    if (reduce(user, 1, f->xch + f->off[1], f->cnt[1]) != 0)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "sharded posterior_logpdf: the all-reduce callback failed (segment 1)");
    return fitter_logpdf_finish(f, logpdf);
}

extern "C" {

int gingr_fitter_posterior_logpdf_sharded(gingr_fitter *f, int32_t flavour, const gingr_cpd_params *cp, const gingr_icp_params *ip,
                                          const double *mesh_xyz_full, gingr_allreduce_fn reduce, void *user, double *logpdf) {
    if (!f) return GINGR_ERR_BAD_ARGUMENT;
    return fitter_sharded_logpdf(f, flavour, cp, ip, mesh_xyz_full, reduce, user, logpdf, nullptr);
}

// ===================================================================================== stateless model operators
static void fill_scalars(gingr_state_scalars *s, const double euler[3], const double center[3], const double translation[3],
                         double scale) {
    memset(s, 0, sizeof(*s));
    for (int q = 0; q < 3; ++q) {
        s->euler[q] = euler[q];
        s->center[q] = center[q];
        s->translation[q] = translation[q];
    }
    s->scale = scale;
    s->sigma2 = 1.0;
}

int gingr_model_instance(gingr_ctx *ctx, const gingr_model *model, const double *alpha, const double euler[3],
                         const double center[3], const double translation[3], double scale, double *out_xyz) {
    if (!ctx || !model || !alpha || !euler || !center || !translation || !out_xyz) return GINGR_ERR_BAD_ARGUMENT;
    gingr_fitter *f = nullptr;
    // a non-finalized shard can still be instantiated: bypass the finalize check through a local flag
    gingr_model *mm = const_cast<gingr_model *>(model);
    const bool was = mm->finalized;
    mm->finalized = true;
    int rc = gingr_fitter_create(ctx, model, &f);
    mm->finalized = was;
    if (rc) return rc;
    gingr_state_scalars s;
    fill_scalars(&s, euler, center, translation, scale);
    rc = gingr_fitter_set_state(f, alpha, &s);
    if (!rc) rc = gingr_fitter_get_state(f, nullptr, nullptr, out_xyz);
    gingr_fitter_destroy(f);
    return rc;
}

int gingr_model_coefficients(gingr_ctx *ctx, const gingr_model *model, const double euler[3], const double center[3],
                             const double translation[3], const double *mesh_xyz, double *alpha) {
    if (!ctx || !model || !euler || !center || !translation || !mesh_xyz || !alpha) return GINGR_ERR_BAD_ARGUMENT;
    if (model->M != model->M_total)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "model_coefficients: single-shard models only");
    gingr_fitter *f = nullptr;
    GINGR_TRY(gingr_fitter_create(ctx, model, &f));
    const int64_t M = model->M;
    const int32_t r = model->r, rp = model->rp;
    int rc = GINGR_OK;
    std::vector<double> zero((size_t)r, 0.0);
    gingr_state_scalars s;
    fill_scalars(&s, euler, center, translation, 1.0);
    rc = gingr_fitter_set_state(f, zero.data(), &s);
    DevBuf aos, pose_h;
    if (!rc && aos.alloc((size_t)3 * M * sizeof(double)) != hipSuccess) rc = gingr_set_error(ctx, GINGR_ERR_HIP, "out of memory");
    if (!rc) {
        (void)hipMemcpyAsync(aos.p, mesh_xyz, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        launch_aos_to_soa(ctx, aos.as<double>(), M, f->newshape, model->perm);
        // pose := the state's rigid transform
        DevState hst;
        (void)hipMemcpyAsync(&hst, f->st, sizeof(hst), hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        DevPose hp;
        memcpy(hp.R, hst.R, sizeof(hp.R));
        memcpy(hp.euler, hst.euler, sizeof(hp.euler));
        memcpy(hp.t, hst.t, sizeof(hp.t));
        memcpy(hp.center, hst.center, sizeof(hp.center));
        hp.scale = 1.0;
        (void)hipMemcpyAsync(f->pose, &hp, sizeof(hp), hipMemcpyHostToDevice, ctx->stream);
        SweepArgs a = base_args(f);
        a.shape_in = f->newshape;
        a.out = f->acoef;
        launch_sweep(ctx, SWEEP_PROJ2, a);
        launch_coeff_solve(ctx, r, rp, model->Binv, f->acoef, f->alpha_c);
        rc = check_launch(ctx);
        if (!rc && hipMemcpyAsync(alpha, f->alpha_c, (size_t)r * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
            rc = gingr_set_error(ctx, GINGR_ERR_HIP, "copy failed");
        (void)hipStreamSynchronize(ctx->stream);
        if (!rc)
            for (int32_t k = 0; k < r; ++k)
                if (!std::isfinite(alpha[k])) {
                    rc = gingr_set_error(ctx, GINGR_ERR_NONFINITE, "model_coefficients: non-finite coefficient");
                    break;
                }
    }
    gingr_fitter_destroy(f);
    return rc;
}

int gingr_model_posterior_mean(gingr_ctx *ctx, const gingr_model *model, const double euler[3], const double center[3],
                               const double translation[3], const double *obs_xyz, const double *weight, int32_t n_lm,
                               const int32_t *lm_pid, const double *lm_xyz, const double *lm_cov, double *mean_xyz,
                               double *coeffs) {
    if (!ctx || !model || !euler || !center || !translation || !obs_xyz || !weight) return GINGR_ERR_BAD_ARGUMENT;
    if (model->M != model->M_total)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "model_posterior_mean: single-shard models only");
    gingr_fitter *f = nullptr;
    GINGR_TRY(gingr_fitter_create(ctx, model, &f));
    const int64_t M = model->M;
    const int32_t r = model->r, rp = model->rp;
    int rc = GINGR_OK;
    std::vector<double> zero((size_t)r, 0.0);
    gingr_state_scalars s;
    fill_scalars(&s, euler, center, translation, 1.0);
    rc = gingr_fitter_set_state(f, zero.data(), &s);
    if (!rc) rc = gingr_fitter_set_landmarks(f, n_lm, lm_pid, lm_xyz, lm_cov);
    DevBuf aos, obs, win, G, gws;
    if (!rc && (aos.alloc((size_t)3 * M * sizeof(double)) != hipSuccess || obs.alloc((size_t)3 * M * sizeof(double)) != hipSuccess ||
                win.alloc((size_t)M * sizeof(double)) != hipSuccess ||
                G.alloc(((size_t)rp * rp + rp) * sizeof(double)) != hipSuccess ||
                gws.alloc((size_t)gram_ws_doubles(M, rp) * sizeof(double)) != hipSuccess))
        rc = gingr_set_error(ctx, GINGR_ERR_HIP, "out of memory");
    if (!rc) {
        std::vector<double> wo((size_t)M), wh((size_t)M);
        for (int64_t i = 0; i < M; ++i) wo[(size_t)i] = weight[i];
        for (int32_t l = 0; l < n_lm; ++l) wo[(size_t)lm_pid[l]] = 0.0;  // landmark pids carry weight 0
        for (int64_t sidx = 0; sidx < M; ++sidx) wh[(size_t)sidx] = wo[(size_t)model->hperm[(size_t)sidx]];  // device order
        (void)hipMemcpyAsync(aos.p, obs_xyz, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        launch_aos_to_soa(ctx, aos.as<double>(), M, obs.as<double>(), model->perm);
        (void)hipMemcpyAsync(win.p, wh.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        launch_obs_points(ctx, model, f->st, obs.as<double>(), win.as<double>(), f->weight, f->evec);
        double *Gd = G.as<double>(), *rhs = Gd + (int64_t)rp * rp;
        launch_gram(ctx, model->Q0, M, rp, f->weight, gws.as<double>(), Gd);
        SweepArgs a = base_args(f);
        a.evec = f->evec;
        a.out = rhs;
        launch_sweep(ctx, SWEEP_RHS, a);
        launch_landmarks(ctx, model, f->st, f->n_lm, f->lm_pid, f->lm_xyz, f->lm_cov, Gd, rhs);
        launch_posterior_solve(ctx, r, rp, Gd, rhs, nullptr, f->work, f->acoef, f->st);
        SweepArgs b = base_args(f);
        b.coef0 = f->acoef;
        b.shape_out = f->newshape;
        launch_sweep(ctx, SWEEP_POSED, b);
        launch_soa_to_aos(ctx, f->newshape, M, aos.as<double>(), model->perm);
        rc = check_launch(ctx);
        DevState hst;
        (void)hipMemcpyAsync(&hst, f->st, sizeof(hst), hipMemcpyDeviceToHost, ctx->stream);
        if (mean_xyz) (void)hipMemcpyAsync(mean_xyz, aos.p, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (coeffs) (void)hipMemcpyAsync(coeffs, f->acoef, (size_t)r * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = gingr_set_error(ctx, GINGR_ERR_HIP, "synchronize failed");
        if (!rc && hst.err) rc = gingr_set_error(ctx, hst.err, "model_posterior_mean: posterior solve failed (%s)",
                                                  hst.err == GINGR_ERR_NOT_SPD ? "not SPD" : "non-finite");
    }
    gingr_fitter_destroy(f);
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ surface distance statistics
namespace {

// out4 = {sum d, max d, count, sum log N(d; 0, sdev)} of d = |q - closest point of the mesh (v, tri)| over the queries q.
// `nn_orig` / `nn_boxes` / `boundary` (all three or none): the boundary-aware variant.  `scratch` holds what the kernels write.
struct StatScratch {
    DevBuf cp, d2, nn, nnd2, ws, part, out;
    // warm start of the closest-point scan: last call's winning triangle per query, valid for the same number of queries against the
    // same triangle array (successive likelihood evaluations of a chain look at nearby shapes)
    DevBuf pos;
    int64_t pos_K = -1, pos_T = -1;
    const int32_t *pos_tri = nullptr;
};

}  // namespace
static void free_stat_scratch(void *p) { delete static_cast<StatScratch *>(p); }
namespace {

// grow-only: steady-state queries (one likelihood evaluation per Metropolis-Hastings step) do not allocate
hipError_t ensure(DevBuf &b, size_t bytes) { return b.p && b.bytes >= bytes ? hipSuccess : b.alloc(bytes); }

int run_distance_stats(gingr_ctx *ctx, Cloud q, Cloud v, const int32_t *tri, const int32_t *tri_orig, int64_t T, const double *tboxes,
                       const int32_t *q_orig, int64_t q_limit, const int32_t *v_orig, const double *v_boxes,
                       const int32_t *boundary, double sdev, StatScratch &sc, double out4[4], double *pinned4 = nullptr,
                       const double *tribox = nullptr, gingr_fitter *spin_on = nullptr) {
    const int64_t K = q.n;
    HIP_TRY(ctx, ensure(sc.cp, (size_t)3 * K * sizeof(double)));
    HIP_TRY(ctx, ensure(sc.d2, (size_t)K * sizeof(double)));
    HIP_TRY(ctx, ensure(sc.part, (size_t)distance_stats_ws_doubles() * sizeof(double)));
    HIP_TRY(ctx, ensure(sc.out, 4 * sizeof(double)));
    HIP_TRY(ctx, ensure(sc.pos, (size_t)K * sizeof(int32_t)));
    const bool warm = sc.pos_K == K && sc.pos_T == T && sc.pos_tri == tri;
    launch_surface_closest_point(ctx, q, v, tri, tri_orig, T, tboxes, sc.cp.as<double>(), sc.d2.as<double>(), nullptr, sc.pos.as<int32_t>(),
                                 warm, tribox);
    sc.pos_K = K, sc.pos_T = T, sc.pos_tri = tri;
    if (boundary) {
        HIP_TRY(ctx, ensure(sc.nn, (size_t)K * sizeof(int32_t)));
        HIP_TRY(ctx, ensure(sc.nnd2, (size_t)K * sizeof(double)));
        HIP_TRY(ctx, ensure(sc.ws, (size_t)nn_ws_bytes(K, v.n)));
        launch_nn(ctx, cloud_of(sc.cp.as<double>(), K), v, v_orig, v_boxes, sc.ws.p, sc.nn.as<int32_t>(), sc.nnd2.as<double>());
    }
    launch_distance_stats(ctx, K, sc.d2.as<double>(), q_orig, q_limit, boundary ? sc.nn.as<int32_t>() : nullptr, boundary, sdev,
                          sc.part.as<double>(), sc.out.as<double>());
    GINGR_TRY(check_launch(ctx));
    if (spin_on && pinned4) {  // (the fitter's pinned buffer: pull_small)
        GINGR_TRY(pull_small(spin_on, sc.out.as<double>(), 4, pinned4));
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(pinned4 ? pinned4 : out4, sc.out.p, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (pinned4) memcpy(out4, pinned4, 4 * sizeof(double));
    return GINGR_OK;
}

// SoA planes of host points taken in the order `order` (device position -> input index)
void gather_soa(const double *xyz, const std::vector<int32_t> &order, std::vector<double> &soa) {
    const size_t n = order.size();
    soa.resize(3 * n);
    for (size_t s2 = 0; s2 < n; ++s2)
        for (int d = 0; d < 3; ++d) soa[(size_t)d * n + s2] = xyz[(size_t)3 * order[s2] + d];
}

}  // namespace

extern "C" {

int gingr_fitter_surface_distance_stats(gingr_fitter *f, int32_t direction, int64_t n_points, const double *points,
                                        int32_t boundary_aware, double sdev, double out[4]) {
    GINGR_TRY(check_ready(f));
    gingr_ctx *ctx = f->ctx;
    if (!out || (direction != 0 && direction != 1) || n_points < 0 || !(sdev >= 0.0))
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "surface_distance_stats: bad argument");
    if (!f->Tm || !f->Tt) return gingr_set_error(ctx, GINGR_ERR_STATE, "surface_distance_stats: no meshes set (gingr_fitter_set_meshes)");
    const gingr_model *m = f->m;
    const int64_t M = m->M, N = f->N;
    const Cloud fit = cloud_of(f->fit, M), tgt = cloud_of(f->target, N);
    if (!f->stat_scratch) f->stat_scratch = new StatScratch;
    StatScratch &sc = *static_cast<StatScratch *>(f->stat_scratch);
    if (direction == 0) {
        // the first n_points vertices of the current fit (original numbering; 0 = all) against the target surface
        if (points) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "surface_distance_stats: model -> target takes no point list");
        if (n_points > M) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "surface_distance_stats: more points than model vertices");
        const bool all = n_points == 0 || n_points == M;
        return run_distance_stats(ctx, fit, tgt, f->ttri, f->ttri_orig, f->Tt, f->ttboxes, all ? nullptr : m->perm, n_points, f->tperm,
                                  f->tboxes, boundary_aware ? f->tboundary : nullptr, sdev, sc, out, f->pin, f->ttribox, f);
    }
    // `points` (null: every target vertex) against the surface of the current fit
    launch_tri_tile_bbox(ctx, fit, f->mtri, f->Tm, f->mtboxes, f->mtribox);
    if (boundary_aware) launch_tile_bbox(ctx, fit, f->fboxes);
    const int32_t *bnd = boundary_aware ? f->mboundary : nullptr;
    if (!points)
        return run_distance_stats(ctx, tgt, fit, f->mtri, f->mtri_orig, f->Tm, f->mtboxes, nullptr, 0, m->perm, f->fboxes, bnd, sdev, sc,
                                  out, f->pin, f->mtribox, f);
    if (n_points < 1) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "surface_distance_stats: empty point list");
    std::vector<int32_t> order;
    morton_order(points, n_points, order);
    std::vector<double> soa;
    gather_soa(points, order, soa);
    DevBuf q;
    HIP_TRY(ctx, q.alloc(soa.size() * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(q.p, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return run_distance_stats(ctx, cloud_of(q.as<double>(), n_points), fit, f->mtri, f->mtri_orig, f->Tm, f->mtboxes, nullptr, 0, m->perm,
                              f->fboxes, bnd, sdev, sc, out);
}

// ------------------------------------------------------------------------------------------ one Metropolis-Hastings step
// What MetropolisHastings.next asks of the device for ONE step of GingrAlgorithm.run's chain (G/api/GingrAlgorithm.scala:115-190,
// generators/GeneratorWrapperStochastic.scala:28-63, evaluators/IndependentPointDistanceEvaluator.scala:54-70), enqueued as one
// sequence with one synchronisation at the end:
//   x  = the device state                         (its posterior inputs come from the memo: every state's are computed once)
//   x' = update(x, probabilistic = true) with z   (kind 0)   or   the parameters the host's random walk proposes (kind 1)
//   q(x'|x)  = posterior(x).logpdf(coefficients(x.fit))       -- with step length 1 the reference projects from.fit, NOT to.fit
//              (GeneratorWrapperStochastic.scala:50-55): a function of x alone, so only asked for when the host does not hold it
//   posterior inputs of x' (correspondences, Gram, right-hand side)
//   L(x')    = sum over the first n fit vertices of log N(|v - closest point of the target surface|; 0, sdev)
//              -- with the surface correspondence these distances ARE the ones the correspondences of x' just measured
//   q(x|x')  = posterior(x').logpdf(coefficients(x'.fit))     -- the q(.|x') of every later step that starts from x'
// The host decides; a rejection is gingr_fitter_mh_restore (x becomes the device state again, nothing waits).
static void mh_tag_state(gingr_fitter *f) {
    f->state_key.v.assign({-1.2345678901234567e300, (double)++f->mh_serial});
    f->state_key_valid = true;
}

static int mh_logpdf_enqueue(gingr_fitter *f, const DevState *frame, const double *mesh_soa, double *out2, bool keep_factor) {
    gingr_ctx *ctx = f->ctx;
    const gingr_model *m = f->m;
    const int32_t r = m->r, rp = m->rp;
    double *G = f->seg1_live(), *rhs = G + (int64_t)rp * rp;
    if (f->lp_epoch == 0) HIP_TRY(ctx, hipMemsetAsync(f->lp_sync, 0, 2 * sizeof(unsigned), ctx->stream));
    const bool cached = f->fx_valid[f->live];
    SweepArgs a = base_args(f);
    a.frame = frame;
    a.shape_in = mesh_soa;
    a.out = f->alpha_c;
    launch_sweep(ctx, SWEEP_PROJ2, a);
    const bool split = !cached && rp <= 112;  // (launch_posterior_logpdf's own choice: the two-workgroup form also leaves the factor of I + G)
    GINGR_TRY(launch_posterior_logpdf(ctx, r, rp, G, rhs, m->mom + MomentLayout{rp}.stot(), f->alpha_c, f->fxbuf[f->live], cached, f->work, out2,
                                      f->lp_sync, ++f->lp_epoch, keep_factor, f->nfac[f->live]));
    if (split && f->post_stage == 2) f->nf_valid[f->live] = true;
    // (taken back after the synchronisation when the kernel reports a failure; ranks above 112 always leave the factor behind)
    if (f->post_stage == 2 && (keep_factor || rp > 112)) f->fx_valid[f->live] = true;
    return check_launch(ctx);
}

#ifdef GINGR_MH_TRACE  // diagnostic build only (tools/mkvar.sh mhtrace fitter -DGINGR_MH_TRACE): where the host side of a step goes
static double g_mh_t[4];  // between calls, enqueue, wait, after the wait (seconds)
static long g_mh_n;
static std::chrono::steady_clock::time_point g_mh_last;
static bool g_mh_has_last;
struct MhTrace {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), t1, t2;
    MhTrace() {
        if (g_mh_has_last) g_mh_t[0] += std::chrono::duration<double>(t0 - g_mh_last).count();
    }
    ~MhTrace() {
        const auto t3 = std::chrono::steady_clock::now();
        g_mh_t[1] += std::chrono::duration<double>(t1 - t0).count();
        g_mh_t[2] += std::chrono::duration<double>(t2 - t1).count();
        g_mh_t[3] += std::chrono::duration<double>(t3 - t2).count();
        g_mh_last = t3, g_mh_has_last = true;
        if (++g_mh_n % 100 == 0)
            fprintf(stderr, "mh_step x%ld: between calls %.1f us, enqueue %.1f, wait %.1f, after %.1f\n", g_mh_n, 1e6 * g_mh_t[0] / g_mh_n,
                    1e6 * g_mh_t[1] / g_mh_n, 1e6 * g_mh_t[2] / g_mh_n, 1e6 * g_mh_t[3] / g_mh_n);
    }
};
#endif

int gingr_fitter_mh_step(gingr_fitter *f, const gingr_mh_request *q, double *alpha_out, double *fit_out, gingr_mh_result *res) {
#ifdef GINGR_MH_TRACE
    MhTrace trace;
#endif
    GINGR_TRY(check_ready(f));
    gingr_ctx *ctx = f->ctx;
    const gingr_model *m = f->m;
    if (!q || !res || !alpha_out || q->flavour < 0 || q->flavour > 2 || (q->kind != 0 && q->kind != 1) || !(q->eval_sdev > 0.0) ||
        q->eval_points < 0 || q->eval_points > m->M || (q->flavour == 0 ? !q->cpd : !q->icp) || (q->kind == 0 ? !q->z : (!q->alpha || !q->scalars)))
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "mh_step: bad request");
    if (q->flavour == 0 && (!(q->cpd->w >= 0.0 && q->cpd->w < 1.0) || !(q->cpd->lambda > 0.0)))
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "cpd params: need 0 <= w < 1 and lambda > 0");
    if (q->flavour != 0 && q->icp->max_iterations < 1) return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "icp params: max_iterations < 1");
    if (m->M != m->M_total || f->partial_out) return gingr_set_error(ctx, GINGR_ERR_STATE, "mh_step: single shard only");
    if (!f->Tm || !f->Tt) return gingr_set_error(ctx, GINGR_ERR_STATE, "mh_step: no meshes set (gingr_fitter_set_meshes)");
    if (f->step_length != 1.0) return gingr_set_error(ctx, GINGR_ERR_STATE, "mh_step: step length 1 only (the transition density projects from.fit)");
    if (!f->state_key_valid)
        return gingr_set_error(ctx, GINGR_ERR_STATE, "mh_step: the device state is not one the host set or read (gingr_fitter_set_state)");
    const int64_t M = m->M;
    const int32_t r = m->r, rp = m->rp;
    const size_t head = (size_t)rp + kScalarsDoubles + kDevStateDoubles;
    const int flavour = q->flavour;
    int rc = GINGR_OK;
    f->allow_alt = true;
    struct Restore {  // whatever happens below, the entry-point-scoped switches go back
        gingr_fitter *f;
        ~Restore() { f->allow_alt = false, f->zrand_active = false; }
    } restore{f};
    // (1) the posterior inputs of x (memo: they exist unless x is the first state of the chain)
    for (int ph = 0; ph < 2 && rc == GINGR_OK; ++ph) rc = flavour_phase(f, flavour, q->cpd, q->icp, ph);
    GINGR_TRY(rc);
    // (2) x stays: parameters + device state in mh_save (parked by the launch that also brings the proposal's draws / parameters, or
    // by a copy where those do not fit a kernel argument), the fit by exchanging the two fit buffers
    const bool by_kernel = (size_t)rp + kScalarsDoubles <= (size_t)kMhPayload && head <= 256;
    if (!by_kernel) HIP_TRY(ctx, hipMemcpyAsync(f->mh_save, f->state_block, head * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    f->mh_key = f->state_key;
    std::swap(f->fit, f->fit_alt);
    // From here on the device state is in flux (fit pointers exchanged, state block and memo keys about to be rewritten): a failure
    // on the way must not leave something behind that the next mh_step / mh_restore would take for a consistent state -- the
    // caller is sent back through gingr_fitter_set_state.
    struct Poison {
        gingr_fitter *f;
        bool armed = true;
        ~Poison() {
            if (!armed) return;
            f->state_key_valid = false;
            f->mh_saved = false;
            f->forget_posteriors();
        }
    } poison{f};
    const DevState *x_state = reinterpret_cast<const DevState *>(f->mh_save + rp + kScalarsDoubles);
    // (3) the proposal
    memset(f->pin, 0, ((size_t)rp + kScalarsDoubles) * sizeof(double));
    MhPayload payload;
    if (by_kernel) memset(&payload, 0, sizeof(payload));
    if (q->kind == 0) {
        if (by_kernel) {
            memcpy(payload.v, q->z, (size_t)r * sizeof(double));
            hipLaunchKernelGGL(mh_begin_kernel, dim3(1), dim3(256), 0, ctx->stream, payload, (int)rp, f->zrand, f->state_block, (int)head, f->mh_save,
                               (DevState *)nullptr, (const gingr_state_scalars *)nullptr, (double *)nullptr);
        } else {
            memcpy(f->pin, q->z, (size_t)r * sizeof(double));
            HIP_TRY(ctx, hipMemcpyAsync(f->zrand, f->pin, (size_t)rp * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        }
        f->zrand_active = true;
        rc = flavour_phase(f, flavour, q->cpd, q->icp, 2);
        f->zrand_active = false;
        GINGR_TRY(rc);
        mh_tag_state(f);
    } else {
        if (by_kernel) {
            memcpy(payload.v, q->alpha, (size_t)r * sizeof(double));
            memcpy(payload.v + rp, q->scalars, sizeof(*q->scalars));
            hipLaunchKernelGGL(mh_begin_kernel, dim3(1), dim3(256), 0, ctx->stream, payload, (int)(rp + kScalarsDoubles), f->state_block, f->state_block,
                               (int)head, f->mh_save, f->st, (const gingr_state_scalars *)f->hs_dev, f->absmax + 1);
        } else {
            memcpy(f->pin, q->alpha, (size_t)r * sizeof(double));
            memcpy(f->pin + rp, q->scalars, sizeof(*q->scalars));
            HIP_TRY(ctx, hipMemcpyAsync(f->state_block, f->pin, ((size_t)rp + kScalarsDoubles) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            launch_state_init(ctx, f->st, f->hs_dev, f->absmax + 1);
        }
        refresh_fit(f);
        GINGR_TRY(check_launch(ctx));
        const gingr_state_scalars *s = q->scalars;  // the host knows this state: keyed by value, like gingr_fitter_set_state
        f->state_key.v.assign(q->alpha, q->alpha + r);
        for (int k = 0; k < 3; ++k) f->state_key.v.push_back(s->euler[k]);
        for (int k = 0; k < 3; ++k) f->state_key.v.push_back(s->center[k]);
        for (int k = 0; k < 3; ++k) f->state_key.v.push_back(s->translation[k]);
        f->state_key.v.push_back(s->scale);
        f->state_key.v.push_back(s->sigma2);
        f->state_key_valid = true;
    }
    // (4) q(x'|x): the live posterior slot still holds x; frame and mesh of x
    const int slot_fw = f->live;
    if (q->need_forward) GINGR_TRY(mh_logpdf_enqueue(f, x_state, f->fit_alt, f->small, true));
    // (5) the posterior inputs of x' (x's are parked in the second slot)
    rc = flavour_phase(f, flavour, q->cpd, q->icp, 0);
    const bool memo_hit = f->skip_phase1;
    if (rc == GINGR_OK) rc = flavour_phase(f, flavour, q->cpd, q->icp, 1);
    GINGR_TRY(rc);
    // (6) the likelihood of x'
    if (!f->stat_scratch) f->stat_scratch = new StatScratch;
    StatScratch &sc = *static_cast<StatScratch *>(f->stat_scratch);
    HIP_TRY(ctx, ensure(sc.part, (size_t)distance_stats_ws_doubles() * sizeof(double)));
    const double *d2 = f->surf_d2;
    if (!(flavour == 2 && !memo_hit && !f->reversed && f->surface_method == 0)) {  // no fresh closest-point scan of x' to share
        const Cloud fit = cloud_of(f->fit, M), tgt = cloud_of(f->target, f->N);
        HIP_TRY(ctx, ensure(sc.cp, (size_t)3 * M * sizeof(double)));
        HIP_TRY(ctx, ensure(sc.d2, (size_t)M * sizeof(double)));
        HIP_TRY(ctx, ensure(sc.pos, (size_t)M * sizeof(int32_t)));
        const bool warm = sc.pos_K == M && sc.pos_T == f->Tt && sc.pos_tri == f->ttri;
        launch_surface_closest_point(ctx, fit, tgt, f->ttri, f->ttri_orig, f->Tt, f->ttboxes, sc.cp.as<double>(), sc.d2.as<double>(), nullptr,
                                     sc.pos.as<int32_t>(), warm, f->ttribox);
        sc.pos_K = M, sc.pos_T = f->Tt, sc.pos_tri = f->ttri;
        d2 = sc.d2.as<double>();
    }
    const bool all = q->eval_points == 0 || q->eval_points == M;
    launch_distance_stats(ctx, M, d2, all ? nullptr : m->perm, q->eval_points, nullptr, nullptr, q->eval_sdev, sc.part.as<double>(), f->small + 4);
    // (7) q(x|x'): frame, posterior and mesh of x'
    const int slot_bw = f->live;
    // (this state's density is asked for once: the host keeps the number, so the factor need not be left behind)
    GINGR_TRY(mh_logpdf_enqueue(f, f->st, f->fit, f->small + 2, false));
    // (8) ONE transfer back: [alpha | scalars | DevState] of x', the eight results (small follows the state block) and, on request, the fit,
    // gathered by one launch
    // Small results (always) and the fit of small templates go straight into the pinned buffer, the host spins on the flag word; the
    // fit of a large template (scattered 24-byte stores over the host link) keeps the gather on the device + one copy.
    const bool direct = f->pin_dev != nullptr && (!fit_out || M <= 8192);
    if (!direct && !f->mh_rb) GINGR_TRY(dev_alloc(ctx, &f->mh_rb, head + 8 + (size_t)3 * M));
    volatile double *flag = f->pin + f->pin_doubles - 1;
    const double epoch = (double)(++f->mh_epoch);
    {
        const int64_t n = fit_out ? std::max<int64_t>(M, (int64_t)head + 8) : (int64_t)head + 8;
        if (direct) {
            hipLaunchKernelGGL(mh_readback_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, f->state_block, (int)(head + 8),
                               fit_out ? f->fit : (const double *)nullptr, M, m->perm, f->pin_dev, f->pin_dev + f->pin_doubles - 1, f->mh_done, epoch);
        } else {
            hipLaunchKernelGGL(mh_readback_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, f->state_block, (int)(head + 8),
                               fit_out ? f->fit : (const double *)nullptr, M, m->perm, f->mh_rb, (double *)nullptr, (int32_t *)nullptr, 0.0);
            HIP_TRY(ctx, hipMemcpyAsync(f->pin, f->mh_rb, (head + 8 + (fit_out ? (size_t)3 * M : 0)) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    GINGR_TRY(check_launch(ctx));
#ifdef GINGR_MH_TRACE
    trace.t1 = std::chrono::steady_clock::now();
#endif
    bool seen = false;
    if (direct) {  // spin on the flag; a launch that never finishes (a fault) is left to the stream synchronisation below to report
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
        for (unsigned spins = 0;; ++spins) {
            if (*flag == epoch) {
                seen = true;
                break;
            }
            if ((spins & 1023u) == 1023u) {
                if (std::chrono::steady_clock::now() > deadline) break;
                if (spins > 65536u) std::this_thread::yield();  // (a long wait: leave the core to whoever else needs it)
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
#ifdef GINGR_MH_TRACE
    trace.t2 = std::chrono::steady_clock::now();
#endif
    poison.armed = false;
    f->mh_saved = true;
    DevState hst;
    memcpy(&hst, f->pin + rp + kScalarsDoubles, sizeof(hst));
    memcpy(alpha_out, f->pin, (size_t)r * sizeof(double));
    if (fit_out) memcpy(fit_out, f->pin + head + 8, (size_t)3 * M * sizeof(double));
    memset(res, 0, sizeof(*res));
    for (int k = 0; k < 3; ++k) {
        res->scalars.euler[k] = hst.euler[k];
        res->scalars.center[k] = hst.center[k];
        res->scalars.translation[k] = hst.t[k];
    }
    res->scalars.scale = hst.scale;
    res->scalars.sigma2 = hst.sigma2;
    res->scalars.iteration = hst.iteration;
    res->scalars.status = hst.status;
    const double *o = f->pin + head;
    auto density = [&](const double *v, int slot, double *lp, int32_t *status) {
        *status = v[1] != 0.0 ? GINGR_ERR_NOT_SPD : (std::isfinite(v[0]) ? GINGR_OK : GINGR_ERR_NONFINITE);
        *lp = *status == GINGR_OK ? v[0] : -INFINITY;
        if (*status != GINGR_OK) f->fx_valid[slot] = f->nf_valid[slot] = false;  // nothing usable was left behind for the cached forms
    };
    if (q->need_forward) {
        density(o, slot_fw, &res->log_q_forward, &res->forward_status);
    } else {
        res->log_q_forward = NAN;
        res->forward_status = -1;  // not asked for
    }
    density(o + 2, slot_bw, &res->log_q_backward, &res->backward_status);
    res->dist_sum = o[4];
    res->dist_max = o[5];
    res->count = (int64_t)o[6];
    res->log_value = o[7];
    if (q->kind == 0) {  // the host now knows the state the update produced: value key, so that set_state of the same numbers finds its memo
        gingr_fitter::Key k = f->state_key;
        k.v.assign(alpha_out, alpha_out + r);
        for (int d = 0; d < 3; ++d) k.v.push_back(hst.euler[d]);
        for (int d = 0; d < 3; ++d) k.v.push_back(hst.center[d]);
        for (int d = 0; d < 3; ++d) k.v.push_back(hst.t[d]);
        k.v.push_back(hst.scale);
        k.v.push_back(hst.sigma2);
        if (f->post_stage == 2 && f->post_key.v == f->state_key.v) f->post_key.v = k.v;
        if (f->alt_stage == 2 && f->alt_key.v == f->state_key.v) f->alt_key.v = k.v;
        f->state_key.v = k.v;
    }
    return GINGR_OK;
}

int gingr_fitter_mh_restore(gingr_fitter *f) {
    GINGR_TRY(check_ready(f));
    gingr_ctx *ctx = f->ctx;
    if (!f->mh_saved) return gingr_set_error(ctx, GINGR_ERR_STATE, "mh_restore: no gingr_fitter_mh_step since the state was last set");
    const size_t head = (size_t)f->m->rp + kScalarsDoubles + kDevStateDoubles;
    HIP_TRY(ctx, hipMemcpyAsync(f->state_block, f->mh_save, head * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    refresh_fit(f);  // (recomputed rather than taken from fit_alt: the pass also leaves the boxes the CPD passes prune with)
    GINGR_TRY(check_launch(ctx));
    f->state_key = f->mh_key;
    f->state_key_valid = true;
    f->mh_saved = false;
    return GINGR_OK;
}

int gingr_mesh_distance_stats(gingr_ctx *ctx, int64_t n_points, const double *points, int64_t n_vertices, const double *vertices,
                              int64_t n_triangles, const int32_t *triangles, int32_t boundary_aware, double sdev, double out[4]) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if (!points || !vertices || !triangles || !out || n_points < 1 || n_vertices < 1 || n_triangles < 1 || !(sdev >= 0.0) ||
        n_vertices > INT32_MAX || n_triangles > INT32_MAX || n_points > INT32_MAX)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "mesh_distance_stats: bad argument");
    for (int64_t k = 0; k < 3 * n_triangles; ++k)
        if (triangles[k] < 0 || triangles[k] >= n_vertices)
            return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "mesh_distance_stats: vertex id out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // spatial orders: queries, vertices (device position -> original) and triangles (by centroid)
    std::vector<int32_t> qorder, vorder, torder;
    morton_order(points, n_points, qorder);
    morton_order(vertices, n_vertices, vorder);
    std::vector<int32_t> vinv((size_t)n_vertices);
    for (int64_t s2 = 0; s2 < n_vertices; ++s2) vinv[(size_t)vorder[(size_t)s2]] = (int32_t)s2;
    std::vector<double> cen((size_t)3 * n_triangles);
    for (int64_t t = 0; t < n_triangles; ++t)
        for (int d = 0; d < 3; ++d) {
            double c = 0.0;
            for (int k = 0; k < 3; ++k) c += vertices[(size_t)3 * triangles[3 * t + k] + d];
            cen[(size_t)3 * t + d] = c / 3.0;
        }
    morton_order(cen.data(), n_triangles, torder);
    std::vector<int32_t> tri((size_t)3 * n_triangles);
    for (int64_t s2 = 0; s2 < n_triangles; ++s2)
        for (int k = 0; k < 3; ++k) tri[(size_t)3 * s2 + k] = vinv[(size_t)triangles[(size_t)3 * torder[(size_t)s2] + k]];
    std::vector<int32_t> bnd;
    if (boundary_aware) {  // on an edge with exactly one adjacent triangle (TriangleMesh3DOperations.pointIsOnBoundary)
        bnd.assign((size_t)n_vertices, 0);
        std::vector<uint64_t> edges;
        edges.reserve((size_t)3 * n_triangles);
        for (int64_t t = 0; t < n_triangles; ++t)
            for (int k = 0; k < 3; ++k) {
                const uint64_t a = (uint64_t)triangles[3 * t + k], b = (uint64_t)triangles[3 * t + (k + 1) % 3];
                edges.push_back((a < b ? a : b) << 32 | (a < b ? b : a));
            }
        std::sort(edges.begin(), edges.end());
        for (size_t i = 0; i < edges.size();) {
            size_t j = i;
            while (j < edges.size() && edges[j] == edges[i]) ++j;
            if (j - i == 1) {
                bnd[(size_t)vinv[(size_t)(edges[i] >> 32)]] = 1;
                bnd[(size_t)vinv[(size_t)(edges[i] & 0xffffffffu)]] = 1;
            }
            i = j;
        }
    }
    std::vector<double> qsoa, vsoa;
    gather_soa(points, qorder, qsoa);
    gather_soa(vertices, vorder, vsoa);
    const int64_t ntiles = ceil_div(n_triangles, 256), nvt = ceil_div(n_vertices, 256);
    DevBuf dq, dv, dtri, dorig, dtb, dvorig, dvb, dbnd;
    HIP_TRY(ctx, dq.alloc(qsoa.size() * sizeof(double)));
    HIP_TRY(ctx, dv.alloc(vsoa.size() * sizeof(double)));
    HIP_TRY(ctx, dtri.alloc(tri.size() * sizeof(int32_t)));
    HIP_TRY(ctx, dorig.alloc(torder.size() * sizeof(int32_t)));
    HIP_TRY(ctx, dtb.alloc((size_t)30 * ntiles * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(dq.p, qsoa.data(), qsoa.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dv.p, vsoa.data(), vsoa.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dtri.p, tri.data(), tri.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dorig.p, torder.data(), torder.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const Cloud q = cloud_of(dq.as<double>(), n_points), v = cloud_of(dv.as<double>(), n_vertices);
    launch_tri_tile_bbox(ctx, v, dtri.as<int32_t>(), n_triangles, dtb.as<double>());
    if (boundary_aware) {
        HIP_TRY(ctx, dvorig.alloc(vorder.size() * sizeof(int32_t)));
        HIP_TRY(ctx, dvb.alloc((size_t)30 * nvt * sizeof(double)));
        HIP_TRY(ctx, dbnd.alloc(bnd.size() * sizeof(int32_t)));
        HIP_TRY(ctx, hipMemcpyAsync(dvorig.p, vorder.data(), vorder.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dbnd.p, bnd.data(), bnd.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        launch_tile_bbox(ctx, v, dvb.as<double>());
    }
    StatScratch sc;
    return run_distance_stats(ctx, q, v, dtri.as<int32_t>(), dorig.as<int32_t>(), n_triangles, dtb.as<double>(), nullptr, 0,
                              boundary_aware ? dvorig.as<int32_t>() : nullptr, boundary_aware ? dvb.as<double>() : nullptr,
                              boundary_aware ? dbnd.as<int32_t>() : nullptr, sdev, sc, out);
}

int gingr_mesh_closest_points(gingr_ctx *ctx, int64_t n_points, const double *points, int64_t n_vertices, const double *vertices,
                              int64_t n_triangles, const int32_t *triangles, double *cp_xyz, double *d2, int32_t *tri_id,
                              double *bary) {
    if (!ctx) return GINGR_ERR_BAD_ARGUMENT;
    if (!points || !vertices || !triangles || n_points < 1 || n_vertices < 1 || n_triangles < 1 || n_vertices > INT32_MAX ||
        n_triangles > INT32_MAX || n_points > INT32_MAX)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "mesh_closest_points: bad argument");
    for (int64_t k = 0; k < 3 * n_triangles; ++k)
        if (triangles[k] < 0 || triangles[k] >= n_vertices)
            return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "mesh_closest_points: vertex id out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> qorder, vorder, torder;
    morton_order(points, n_points, qorder);
    morton_order(vertices, n_vertices, vorder);
    std::vector<int32_t> vinv((size_t)n_vertices);
    for (int64_t s2 = 0; s2 < n_vertices; ++s2) vinv[(size_t)vorder[(size_t)s2]] = (int32_t)s2;
    std::vector<double> cen((size_t)3 * n_triangles);
    for (int64_t t = 0; t < n_triangles; ++t)
        for (int d = 0; d < 3; ++d) {
            double c = 0.0;
            for (int k = 0; k < 3; ++k) c += vertices[(size_t)3 * triangles[3 * t + k] + d];
            cen[(size_t)3 * t + d] = c / 3.0;
        }
    morton_order(cen.data(), n_triangles, torder);
    std::vector<int32_t> tri((size_t)3 * n_triangles), tri_by_orig((size_t)3 * n_triangles);
    for (int64_t s2 = 0; s2 < n_triangles; ++s2)
        for (int k = 0; k < 3; ++k) tri[(size_t)3 * s2 + k] = vinv[(size_t)triangles[(size_t)3 * torder[(size_t)s2] + k]];
    for (int64_t k = 0; k < 3 * n_triangles; ++k) tri_by_orig[(size_t)k] = vinv[(size_t)triangles[(size_t)k]];
    std::vector<double> qsoa, vsoa;
    gather_soa(points, qorder, qsoa);
    gather_soa(vertices, vorder, vsoa);
    const int64_t ntiles = ceil_div(n_triangles, 256);
    DevBuf dq, dv, dtri, dorig, dtb, dcp, dd2, dtid, dtbo, dbary;
    HIP_TRY(ctx, dq.alloc(qsoa.size() * sizeof(double)));
    HIP_TRY(ctx, dv.alloc(vsoa.size() * sizeof(double)));
    HIP_TRY(ctx, dtri.alloc(tri.size() * sizeof(int32_t)));
    HIP_TRY(ctx, dtbo.alloc(tri.size() * sizeof(int32_t)));
    HIP_TRY(ctx, dorig.alloc(torder.size() * sizeof(int32_t)));
    HIP_TRY(ctx, dtb.alloc((size_t)30 * ntiles * sizeof(double)));
    HIP_TRY(ctx, dcp.alloc((size_t)3 * n_points * sizeof(double)));
    HIP_TRY(ctx, dd2.alloc((size_t)n_points * sizeof(double)));
    HIP_TRY(ctx, dtid.alloc((size_t)n_points * sizeof(int32_t)));
    HIP_TRY(ctx, dbary.alloc((size_t)3 * n_points * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(dq.p, qsoa.data(), qsoa.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dv.p, vsoa.data(), vsoa.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dtri.p, tri.data(), tri.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dtbo.p, tri_by_orig.data(), tri.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dorig.p, torder.data(), torder.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const Cloud q = cloud_of(dq.as<double>(), n_points), v = cloud_of(dv.as<double>(), n_vertices);
    launch_tri_tile_bbox(ctx, v, dtri.as<int32_t>(), n_triangles, dtb.as<double>());
    launch_surface_closest_point(ctx, q, v, dtri.as<int32_t>(), dorig.as<int32_t>(), n_triangles, dtb.as<double>(), dcp.as<double>(),
                                 dd2.as<double>(), dtid.as<int32_t>());
    launch_barycentric(ctx, q, v, dtbo.as<int32_t>(), dtid.as<int32_t>(), dbary.as<double>());
    GINGR_TRY(check_launch(ctx));
    std::vector<double> hcp((size_t)3 * n_points), hd2((size_t)n_points), hb((size_t)3 * n_points);
    std::vector<int32_t> ht((size_t)n_points);
    HIP_TRY(ctx, hipMemcpyAsync(hcp.data(), dcp.p, hcp.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hd2.data(), dd2.p, hd2.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hb.data(), dbary.p, hb.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ht.data(), dtid.p, ht.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t s2 = 0; s2 < n_points; ++s2) {  // device position -> input index
        const size_t o = (size_t)qorder[(size_t)s2];
        if (cp_xyz)
            for (int d = 0; d < 3; ++d) cp_xyz[3 * o + d] = hcp[(size_t)d * n_points + s2];
        if (d2) d2[o] = hd2[(size_t)s2];
        if (tri_id) tri_id[o] = ht[(size_t)s2];
        if (bary)
            for (int d = 0; d < 3; ++d) bary[3 * o + d] = hb[(size_t)3 * s2 + d];
    }
    return GINGR_OK;
}

int gingr_model_new_reference(gingr_ctx *ctx, const gingr_model *src, int64_t M_new, const double *new_ref,
                              const int32_t *vertex_ids, const double *weights, int64_t row_begin, int64_t row_end,
                              gingr_model **out) {
    if (!ctx || !out) return GINGR_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (!src || !new_ref || !vertex_ids || !weights || M_new < 1)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "model_new_reference: bad argument");
    if (src->ctx != ctx || src->row_begin != 0 || src->row_end != src->M_total)
        return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "model_new_reference: the source must be a complete model of this context");
    const int64_t Ms = src->M;
    for (int64_t k = 0; k < 3 * M_new; ++k)
        if (vertex_ids[k] < 0 || vertex_ids[k] >= Ms)
            return gingr_set_error(ctx, GINGR_ERR_BAD_ARGUMENT, "model_new_reference: source vertex id out of range");
    if (row_end <= 0) row_end = M_new;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // mean displacement of the new points (host: 3 M_new values)
    std::vector<double> smean((size_t)3 * Ms), nmean((size_t)3 * M_new);
    GINGR_TRY(gingr_model_download(ctx, src, nullptr, smean.data(), nullptr, nullptr));
    for (int64_t i = 0; i < M_new; ++i)
        for (int d = 0; d < 3; ++d) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += weights[3 * i + k] * smean[(size_t)3 * vertex_ids[3 * i + k] + d];
            nmean[(size_t)3 * i + d] = acc;
        }
    DevBuf dids, dw, dinv;
    HIP_TRY(ctx, dids.alloc((size_t)3 * M_new * sizeof(int32_t)));
    HIP_TRY(ctx, dw.alloc((size_t)3 * M_new * sizeof(double)));
    HIP_TRY(ctx, dinv.alloc((size_t)Ms * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpyAsync(dids.p, vertex_ids, (size_t)3 * M_new * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dw.p, weights, (size_t)3 * M_new * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dinv.p, src->hiperm.data(), (size_t)Ms * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    auto fill = [&](gingr_model *m) -> int {
        launch_interp_pack(ctx, src->Q0, src->rp, dinv.as<int32_t>(), dids.as<int32_t>(), dw.as<double>(), m->perm, m->row_begin, m->M,
                           m->Q0);
        return check_launch(ctx);
    };
    return model_create_impl(ctx, M_new, src->r, new_ref, nmean.data(), src->variance.data(), row_begin, row_end, fill, out);
}

}  // extern "C"

// Internal definitions shared by the translation units of libgingr_hip.so (gfx950 only).
#pragma once
#include <cstdlib>

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gingr_hip.h"

#define GINGR_TIMERS 10

struct gingr_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    // timing hooks (bench roofline): per-kernel event pairs recorded on `stream`
    bool timing = false;
    struct Span {
        hipEvent_t a, b;
        int which;
    };
    std::vector<Span> spans;     // recorded, not yet resolved
    std::vector<hipEvent_t> pool;  // recycled events
    double t_ms[GINGR_TIMERS] = {0};
    int64_t t_n[GINGR_TIMERS] = {0};
    // exact-zero tile culling of the CPD passes and exact pruning of the closest-point scans (affinity.hip); gingr_ctx_set_option
    // (GINGR_OPT_CULL, 0) disables both (results must stay bit-identical: the culling test compares the two)
    int cull = 1;
    // GINGR_OPT_TRI_GRID: closest surface point over the target's triangle grid (surface.hip) in front of the tile scan: 0 never, 1 from
    // kTriGridMinTriangles target triangles on (default), 2 always.  Bit-identical results either way.
    int tri_grid = 1;
    int nn_grid = 1;  // GINGR_OPT_NN_GRID: closest point over the target's uniform grid (nn_grid.hip); 0 = the tile scan alone; 2 = gingr_nn too
    // Culling regime of the CPD passes as the DEVICE last saw it (0 plain, 1 quarter-tile culling pays): pinned host word the
    // all-pairs kernels write, read -- unsynchronised, possibly a few launches stale -- when the next launch picks its kernel
    // variant.  Both variants compute bit-identical results, so a stale value only costs time.  Null: always the plain variant.
    int32_t *regime_host = nullptr, *regime_dev = nullptr;
    int fine_override = -1;  // GINGR_OPT_FINE_CULL 0|1 pins the variant (tests: both must give bit-identical results); -1: by regime
    // diagnostics (gingr_ctx_nn_counting): device counter of the distance tests the nearest-neighbour launches really execute; null = off
    unsigned long long *nn_tests = nullptr;
    // native RCCL exchange (rccl_exchange.hip): the communicator of this rank, owned by the context; null = none
    void *rccl_comm = nullptr;
    int32_t rccl_world = 0, rccl_rank = 0;
    // GINGR_OPT_SPLIT_EXCHANGE: the column-sum exchange in two halves, the first on a second stream of the context behind the first
    // half of pass 1 (fitter.hip: fitter_sharded_update).  side_stream / the events are created on first use; exchange_stream is
    // where the native all-reduce is enqueued right now (nullptr = the context's stream).
    int split_exchange = 0;
    int gram_downdate = -1;  // GINGR_OPT_GRAM_DOWNDATE: 0 / 1 weights of the surface ICP -> the model's moment minus the rejected rows (fitter.hip, phase 1); -1: by size
    hipStream_t side_stream = nullptr, exchange_stream = nullptr;
    hipEvent_t split_ev[2] = {nullptr, nullptr};
    // scratch kept across calls (grown on demand, never shrunk) so steady-state updates do not allocate
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
};

int gingr_set_error(gingr_ctx *ctx, int code, const char *fmt, ...);
void gingr_ctx_rccl_release(gingr_ctx *ctx);  // rccl_exchange.hip: destroys the context's communicator, if any

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess)                                                                               \
            return gingr_set_error((ctx), GINGR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                   __FILE__, __LINE__);                                                      \
    } while (0)

#define GINGR_TRY(expr)          \
    do {                         \
        int s__ = (expr);        \
        if (s__ != GINGR_OK) return s__; \
    } while (0)

// RAII device buffer for the stateless operators
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = n;
        const hipError_t e = hipMalloc(&p, n ? n : 8);
        static const bool poison = getenv("GINGR_DEBUG_POISON") != nullptr;  // (diagnostic: see dev_alloc, fitter.hip)
        if (e == hipSuccess && poison) {
            (void)hipMemset(p, 0xFF, n ? n : 8);
            (void)hipDeviceSynchronize();
        }
        return e;
    }
    template <typename T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

// Raise a kernel's dynamic-LDS limit -- ONCE per device, function and size, under a lock.  hipFuncSetAttribute in front of every launch
// (rounds 1-5) is a race in a device group, where one host thread per shard launches the same kernels: a launch that overlaps another
// thread's hipFuncSetAttribute on the same function can be rejected, nothing runs, and the consumer reads the previous iteration's
// partials (a 0.2 % flake of the three-shard rank-150 tests, unmasked by tools/experiments/stress_group2.py).
void set_dynamic_lds(const void *func, int bytes);
template <typename K>
inline void set_dynamic_lds(K *kern, size_t bytes) {
    set_dynamic_lds(reinterpret_cast<const void *>(kern), (int)bytes);
}

struct TimerScope {
    gingr_ctx *ctx;
    int which;
    hipEvent_t a = nullptr, b = nullptr;
    TimerScope(gingr_ctx *c, int w);
    void stop();
    ~TimerScope();
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ---------------------------------------------------------------------------- affinity.hip (launchers, all async)
// Points on the device are SoA: x[n], y[n], z[n] contiguous planes of one allocation (plane stride = n).
struct Cloud {
    const double *x, *y, *z;
    int64_t n;
};

// workspace sizes (in doubles) the launchers need
int64_t cpd_colsum_ws_doubles(int64_t M, int64_t N);
int64_t cpd_rowstats_ws_doubles(int64_t M, int64_t N);
int64_t nn_ws_bytes(int64_t M, int64_t N);
// all pairs of two small clouds, both in the caller's order: one launch of ~4 000 single-wave workgroups + the combination of their
// slices (affinity.hip: nn_small_kernel); the stateless gingr_nn uses it where it applies
bool nn_small_applies(int64_t M, int64_t N);
int64_t nn_small_ws_bytes(int64_t M, int64_t N);
void launch_nn_small(gingr_ctx *ctx, Cloud query, Cloud target, void *ws, int32_t *idx, double *d2);

// den_partial[N] = sum_{i in fit} exp(-|x_j - y_i|^2 / (2 sigma2))   (no outlier constant)
// aux (GINGR_AUX doubles on the device): [2..4] centroid of the target cloud (launch_cloud_centroid); [0] / [1] largest
// |coordinate - centroid| of the target / fit cloud (launch_cloud_absmax).  From them the kernels decide, wave-uniformly,
// (a) whether the exponent argument needs clamping and (b) whether the cheaper norm-expansion form of c|x-y|^2 is
// accurate enough (affinity.hip: use_expansion).
#define GINGR_AUX 8
void launch_cloud_centroid(gingr_ctx *ctx, Cloud c, double *out3);
void launch_cloud_absmax(gingr_ctx *ctx, Cloud c, const double *ctr, double *slot);
// boxes[tile] = {lo[3], hi[3]} of every 256-point tile of a cloud: input of the exact-zero tile culling
// with absmax_slot != nullptr also *absmax_slot = max |coordinate - ctr| (same value launch_cloud_absmax produces); the slot
// must have been zeroed by an earlier launch on the stream
void launch_tile_bbox(gingr_ctx *ctx, Cloud c, double *boxes, const double *ctr = nullptr, double *absmax_slot = nullptr);
// returns the number of chunk partials left in ws ([chunk][N]); den_partial == nullptr skips their reduction (the caller passes
// ws and the count to launch_cpd_den_finalize, which then adds them up itself: single shard, one launch less)
// forced_chunks > 0: exactly that many chunks of the streamed rows, balanced to a quarter (a half of a split column-sum pass keeps
// the launch's workgroup count by taking twice the chunks of the whole pass)
int launch_cpd_colsum(gingr_ctx *ctx, Cloud fit, Cloud target, const double *sigma2_dev, const double *aux,
                      const double *fit_boxes, double *ws, double *den_partial, int forced_chunks = 0);
int cpd_colsum_chunks(int64_t M, int64_t N);  // chunks of the planner's own choice
// den[j] += c; inv_den[j] = 1/den[j]; Pt1[j] = (den[j]-c)/den[j]; xPx block partials -> part[0..256)
// M_total enters the outlier constant c = w/(1-w) (2 pi sigma2)^1.5 M_total/N.  part: GINGR_SCALAR_PART doubles.
// tile_bad[tile] (nullable) is set when a 1/den of the tile is not finite: such tiles are never culled.
#define GINGR_SCALAR_PART 1024
void launch_cpd_den_finalize(gingr_ctx *ctx, Cloud target, const double *sigma2_dev, double w, int64_t M_total,
                             double *den, double *inv_den, double *Pt1, int32_t *tile_bad, double *part,
                             double *scalars_dev, const double *chunk_partial = nullptr, int nchunks = 0);
// CPD observations fused into the row-statistics reduction (rowstats_reduce_kernel): pointers into the model / state
struct CpdObsArgs {
    const double *ref, *mean;       // SoA planes of the local rows
    const double *sigma2;           // device scalars of the state: sigma2, R (row-major 3x3), center, translation
    const double *R, *center, *t;
    double lambda;
    const int32_t *lm_mask;         // nullable
    double *weight, *evec;          // out: [M], SoA [3][M]; weight == nullptr: no observations
};
// the four block-partial arrays (GINGR_SCALAR_BLOCKS entries each) the CPD passes leave in `part`: slot 0 xPx (den_finalize),
// 1 Np, 2 trPXY, 3 yPy (rowstats_reduce); scalar index of slot q: {1, 0, 2, 3}[q]
#define GINGR_SCALAR_BLOCKS 256
// P1[i], PX (SoA planes px,py,pz of stride M) for the local rows; Np/xPx/trPXY/yPy sums into scalars_dev[0..3]
// xch8 (nullable): the 8 scalars of the exchange segment {Np, xPx (only when contribute_xpx), trPXY, yPy, 0...}
void launch_cpd_rowstats(gingr_ctx *ctx, Cloud fit, Cloud target, const double *sigma2_dev, const double *aux,
                         const double *inv_den, const double *tgt_boxes, const int32_t *tile_bad, double *ws, double *P1,
                         double *PX_soa, double *part, double *scalars_dev, double *xch8 = nullptr, int contribute_xpx = 1,
                         const CpdObsArgs *obs = nullptr, bool finish_scalars = true);
// idx[i] = POSITION (in the device order of `target`) of the nearest target; exact ties are broken by the lowest ORIGINAL
// index, taken from target_orig[position] (nullptr: the device order is the original order).
// tgt_boxes (nullable): bounding boxes of the 256-point target tiles (launch_tile_bbox) for exact nearest-first pruning.
// mask / nmask (nullable, device): only queries with mask[i] != 0 are answered (idx / d2 of the others are left alone) and the
// launch is a no-op when *nmask == 0 (launch_nn_grid leaves both behind: NNGrid::flag, NNGrid::cur_nflag()).
void launch_nn(gingr_ctx *ctx, Cloud query, Cloud target, const int32_t *target_orig, const double *tgt_boxes, void *ws,
               int32_t *idx, double *d2, const int32_t *warm = nullptr, const uint8_t *mask = nullptr,
               const int32_t *nmask = nullptr);
// ---------------------------------------------------------------------------- nn_grid.hip (closest point over a uniform grid)
struct GridPoint {  // one target, in cell order: coordinates, original index (tie rule), position in the device order of the cloud
    double x, y, z;
    int32_t orig, pos;
};
struct NNGridDev {  // what the kernel needs, by value
    double lo[3];
    double h, inv_h;
    int32_t g[3];
    const int32_t *cell_start;  // [g0 g1 g2 + 1], x fastest
    const GridPoint *pts;       // [n]
    int32_t brute_n;            // n when the cloud is small enough for uncertified queries to scan all of it in the kernel, else 0
};
struct NNGrid {  // owner
    NNGridDev v{};
    int32_t *cell_start = nullptr;
    void *pts = nullptr;
    uint8_t *flag = nullptr;   // [max_queries]: queries the grid could not certify (launch_nn_grid writes every entry)
    // how many of them: two counters used alternately -- a search counts into one and clears the other for the next search, so the
    // masked scan that follows only reads (no reset kernel, no reset by a kernel that others still read)
    int32_t *nflag = nullptr;
    int parity = 0;
    int64_t n = 0, max_queries = 0;
    bool ready = false;
    const int32_t *cur_nflag() const { return nflag + parity; }  // the counter of the LAST launch_nn_grid
};
int nn_grid_build(gingr_ctx *ctx, const double *target_xyz, int64_t N, const int32_t *perm, int64_t max_queries, NNGrid *g);
void nn_grid_free(NNGrid *g);
bool launch_nn_grid(gingr_ctx *ctx, Cloud query, Cloud target, const int32_t *target_orig, NNGrid &g, const int32_t *warm,
                    int32_t *idx, double *d2);
void launch_gauss_block(gingr_ctx *ctx, Cloud A, Cloud B, double sigma, double scaling, double *out);
// sum over all pairs of |a_i - b_j|^2 (computeInitialSigma2, CPD.scala:81-90); ws: sumsq_pairs_ws_doubles(A.n) doubles
int64_t sumsq_pairs_ws_doubles(int64_t nA);
void launch_sumsq_pairs(gingr_ctx *ctx, Cloud A, Cloud B, double *ws, double *out_scalar);
// ---------------------------------------------------------------------------- surface.hip (ICP surface correspondence)
// tri: [3*T] vertex POSITIONS in the cloud `v`, in a spatial triangle order; boxes: one {lo, hi} per 256 triangles
void launch_cell_normals(gingr_ctx *ctx, Cloud v, const int32_t *tri, int64_t T, double *cn_soa);
void launch_vertex_normals(gingr_ctx *ctx, const int32_t *adj_ptr, const int32_t *adj_tri, const double *cn_soa, int64_t T,
                           int64_t n, double *vn_soa);
void launch_tri_tile_bbox(gingr_ctx *ctx, Cloud v, const int32_t *tri, int64_t T, double *boxes, double *tribox = nullptr, double *cell_normals = nullptr);
// closest point of the triangle soup to every query (SoA out); exact ties: lowest tri_orig
// mask / nmask (nullable, device): only queries with mask[i] != 0 are answered and the launch is a no-op when *nmask == 0 (what
// launch_surface_cp_grid leaves behind: TriGrid::flag, TriGrid::cur_nflag())
void launch_surface_closest_point(gingr_ctx *ctx, Cloud q, Cloud v, const int32_t *tri, const int32_t *tri_orig, int64_t T,
                                  const double *boxes, double *cp_soa, double *d2, int32_t *tri_out = nullptr, int32_t *warm = nullptr,
                                  bool warm_valid = false, const double *tribox = nullptr, const uint8_t *mask = nullptr,
                                  const int32_t *nmask = nullptr);
// uniform grid over the (fixed) triangles of a mesh: cell -> triangles whose box overlaps it (surface.hip)
struct TriGridDev {
    double lo[3];
    double h, inv_h;
    int32_t g[3];
    int32_t span[3];            // largest extent (in cell steps) of a listed triangle's box per axis
    int32_t n_listed, n_big;    // entries [0, n_listed) are listed by cell, [n_listed, n_listed + n_big) is the short list of wide triangles
    const int32_t *cell_start;  // [g0 g1 g2 + 1], x fastest: a triangle is listed in the cell of its box's lower corner
    const double *boxes;        // [entries][6]: box lo / hi of every entry
    const double *recs;         // [entries][10]: corners A, B, C, {position | original index << 32} (surface.hip: kTriRec)
};
struct TriGrid {
    TriGridDev v{};
    int32_t *cell_start = nullptr;
    double *boxes = nullptr, *recs = nullptr;
    uint8_t *flag = nullptr;   // [max_queries]: queries the grid search could not certify
    int32_t *nflag = nullptr;  // two counters used alternately (as NNGrid)
    int parity = 0;
    int64_t max_queries = 0, list_entries = 0;
    bool ready = false;
    const int32_t *cur_nflag() const { return nflag + parity; }
};
// The same grid for a MOVING mesh (the template), rebuilt on the device in front of every use (round 5): the description lives in
// device memory (params: the TriGridDev the kernels read, + validity), geometry and lists are recomputed from the per-triangle boxes
// the iteration has computed anyway (tri_tile_bbox_kernel).  Four short launches: set-up (bounding box of the tile boxes, grid
// dimensions at the fixed cell edge h, counters zeroed), count (cell of every triangle's lower corner, wide triangles to the short
// list), scan (128 workgroups with a look-back over their totals, up to 2^20 cells), fill (entries, boxes and corner records in cell order).  Entry order inside a cell
// is whatever the atomics give: the queries' results do not depend on it (self-intersection is an OR; closest points break ties by
// original triangle id).
struct MovGridParams {
    TriGridDev v;
    int32_t valid, pad;
};
struct MovGrid {
    MovGridParams *params = nullptr;  // device
    int32_t *cell_cnt = nullptr, *cell_start = nullptr, *tri_cell = nullptr, *big = nullptr;
    double *boxes = nullptr, *recs = nullptr;
    uint8_t *flag = nullptr;          // [max_queries]: queries the grid could not certify
    int32_t *nflag = nullptr;         // two counters used alternately (as TriGrid)
    unsigned long long *scan_agg = nullptr;  // per scan workgroup: (epoch << 32 | total) of the build in flight
    unsigned epoch = 0;
    int parity = 0;
    int32_t ncap = 0;                 // cells allocated
    int64_t T = 0, max_queries = 0;
    double h = 0.0;                   // cell edge (mean extent of a triangle's box when the grid was set up; any value is correct)
    bool ready = false;
    const int32_t *cur_nflag() const { return nflag + parity; }
};
int mov_grid_alloc(gingr_ctx *ctx, int64_t T, int64_t max_queries, MovGrid *g);
void mov_grid_free(MovGrid *g);
// (re)build for the current vertex positions; tribox [T][6] and tile_boxes [ceil(T / 256)][6] as tri_tile_bbox_kernel left them
void launch_mov_grid_build(gingr_ctx *ctx, MovGrid &g, Cloud v, const int32_t *tri, const int32_t *tri_orig, const double *tribox,
                           const double *tile_boxes);
// self-intersection flags (see launch_self_intersect) over the moving grid: certified queries get their flag, the others are marked
// in g.flag / counted in g.cur_nflag() for the masked launch_self_intersect that must follow
void launch_self_intersect_grid(gingr_ctx *ctx, Cloud fit, const double *cp_soa, MovGrid &g, const int32_t *skip, int32_t *flag);
int tri_grid_build(gingr_ctx *ctx, const double *vsoa_host, int64_t n, const int32_t *tri_host, const int32_t *tri_orig_host, int64_t T,
                   int64_t max_queries, TriGrid *g);
void tri_grid_free(TriGrid *g);
void launch_surface_cp_grid(gingr_ctx *ctx, Cloud q, Cloud v, const int32_t *tri, const int32_t *tri_orig, int64_t T, TriGrid &g,
                            double *cp_soa, double *d2, int32_t *tri_out, int32_t *warm);
// bary[3 i + k] = weight of corner k of triangle tri_id[i] at the closest point of that triangle to query i; tri_by_orig [3 T]:
// corner positions in the cloud v, indexed by ORIGINAL triangle number
void launch_barycentric(gingr_ctx *ctx, Cloud q, Cloud v, const int32_t *tri_by_orig, const int32_t *tri_id, double *bary);
// factor the 64 x 64 diagonal block k of the row-major matrix Aw (leading dimension ld) in place and store its inverse in
// Linv[k] (gp.hip; the diagonal step of classic_cpd.hip's blocked Cholesky); *flag = GINGR_ERR_NOT_SPD on a bad pivot
void launch_chol_block64(gingr_ctx *ctx, double *Aw, int64_t ld, int k, double *Linv, int32_t *flag);
// classic_cpd.hip: the multi-workgroup blocked Cholesky solve of a bordered SPD system (64-column panels, one launch per stage)
void dense_spd_solve3(gingr_ctx *ctx, double *Aw, int64_t Mp, double *Linv, double *W, int32_t *flag);
// A^-1 of an SPD matrix: Aw = [lower triangle of A; identity] ((2 Mp) x Mp), C = Mp x Mp (classic_cpd.hip)
void dense_spd_inverse(gingr_ctx *ctx, double *Aw, int64_t Mp, double *Linv, double *C, int32_t *flag);
// out4 = {sum of sqrt(d2), max, count, sum of log N(sqrt(d2); 0, sdev)} over the counted points (surface.hip)
int distance_stats_ws_doubles();
void launch_distance_stats(gingr_ctx *ctx, int64_t n, const double *d2, const int32_t *orig, int64_t orig_limit, const int32_t *nn,
                           const int32_t *boundary, double sdev, double *partial, double *out4);
// What the self-intersection launch can take over from the launches around it (both are one value per query, computed where the query
// is held anyway):  nn_vertex != nullptr -- the first two rejection tests of launch_surface_prereject are made in the prologue (pre_out
// is written, `skip` is not read);  w01 != nullptr -- launch_surface_weight's outputs are written in the epilogue.
struct SelfIntersectFuse {
    const int32_t *nn_vertex = nullptr, *boundary = nullptr, *found = nullptr;
    const double *q_vn = nullptr, *t_vn = nullptr;  // vertex normals of the queries' mesh [3][n] and of the other mesh [3][Nt]
    int64_t Nt = 0;
    int32_t *pre_out = nullptr;
    const double *sigma2 = nullptr;
    double *w01 = nullptr, *weight_in = nullptr;
};
// mesh (nullable): the cloud the triangles index when it is not the query cloud itself (row shard: the gathered fit of all shards)
// only / nonly (nullable, device): only queries with only[i] != 0 are processed and written, and the launch is a no-op when *nonly == 0
void launch_self_intersect(gingr_ctx *ctx, Cloud fit, const double *cp_soa, const int32_t *tri, int64_t T, const double *boxes,
                           const int32_t *skip, int32_t *flag, const double *tribox = nullptr, const Cloud *mesh = nullptr,
                           const uint8_t *only = nullptr, const int32_t *nonly = nullptr, const SelfIntersectFuse *fuse = nullptr);
// found (nullable): along-normal flavour, 0 = no intersection (rejected)
void launch_surface_prereject(gingr_ctx *ctx, int64_t M, const int32_t *nn_vertex, const int32_t *tgt_boundary,
                              const double *fit_vn, const double *tgt_vn, int64_t N, const int32_t *found, int32_t *pre);
// nearest intersection (!= the vertex) of the line through every fit vertex along dirs with the mesh; cp = the vertex, found = 0 if none
void launch_line_nearest(gingr_ctx *ctx, Cloud fit, const double *dirs_soa, Cloud v, const int32_t *tri, const int32_t *tri_orig,
                         int64_t T, double *boxes, double *cp_soa, int32_t *found);
// the same over the triangle grid of the mesh (static meshes: the target of the forward direction)
void launch_line_nearest_grid(gingr_ctx *ctx, Cloud fit, const double *dirs_soa, const TriGrid &g, double *cp_soa, int32_t *found);
void launch_surface_weight(gingr_ctx *ctx, int64_t M, const int32_t *pre, const int32_t *hit, const double *sigma2_dev, double *w01,
                           double *weight_in);

// reversed correspondence direction: from (nearest template vertex, rejection flags) per TARGET vertex to one observation per
// template vertex (mean of the accepted targets that map to it, weight = count / sigma2); w01_targets (nullable) gets 0 / 1 per target
size_t reversal_sort_temp_bytes(int64_t N);
void launch_reversal_observations(gingr_ctx *ctx, int64_t M, Cloud tgt, const int32_t *nn_vertex, const int32_t *pre,
                                  const int32_t *hit, const double *sigma2_dev, int32_t *keys, int32_t *vals, int32_t *skeys,
                                  int32_t *svals, void *sort_temp, size_t sort_temp_bytes, double *w01_targets, double *obs_soa,
                                  double *weight_in);

// the same for a RANGE of the target queries (tgt = that range), left as sums: sums4 [4][M] = {sum x, sum y, sum z, count} per
// template vertex (a row shard's contribution to the all-reduce of the reversed direction); tgt.n may be 0
void launch_reversal_sums(gingr_ctx *ctx, int64_t M, Cloud tgt, const int32_t *nn_vertex, const int32_t *pre, const int32_t *hit,
                          int32_t *keys, int32_t *vals, int32_t *skeys, int32_t *svals, void *sort_temp, size_t sort_temp_bytes,
                          double *w01_targets, double *sums4);

// interleaved xyz (n*3) <-> SoA planes; perm (nullable) maps device position -> original index:
// soa[s] = aos[perm[s]] resp. aos[perm[s]] = soa[s]
void launch_aos_to_soa(gingr_ctx *ctx, const double *aos, int64_t n, double *soa, const int32_t *perm = nullptr);
void launch_soa_to_aos(gingr_ctx *ctx, const double *soa, int64_t n, double *aos, const int32_t *perm = nullptr);
// out[perm[s]] = in[s]  (perm nullable = copy)
void launch_scatter(gingr_ctx *ctx, const double *in, int64_t n, const int32_t *perm, double *out);
// spatial (balanced k-d tree, leaf = one 256-point tile) order of interleaved points: perm[s] = original index of the
// point stored at device position s
void morton_order(const double *xyz, int64_t n, std::vector<int32_t> &perm);

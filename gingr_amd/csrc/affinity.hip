// All-pairs kernels of the GiNGR update path for gfx950 (MI355X): CPD soft-assignment statistics in two
// streaming passes (P is never materialised), exact brute-force nearest neighbour, Gaussian kernel blocks.
//
// Reference loops replaced (G/ = src/main/scala/gingr/):
//   cpd_colsum   : K_ij and its column sums               G/api/registration/config/CPD.scala:63-68,71
//   cpd_rowstats : P_ij = K_ij/den_j, P1 = row sums, P*X  CPD.scala:36-45,74,138,144
//   nn           : findClosestPoint per fit vertex        G/api/registration/utils/ClosestPointRegistrator.scala:139-145
//   gauss_block  : GaussianKernel(sigma)*scaling          G/api/gpmm/GPMMHelper.scala:99-102
//
// Shape of both CPD passes: a thread keeps PT points of the "owned" side in registers (targets in pass 1, fit points
// in pass 2), the other side streams through LDS in 256-point tiles read with wave-uniform (broadcast) 16-byte
// reads, so HBM traffic is O(M+N) per block column and the kernels are bound by float64 VALU issue (the software
// exponential), not by memory.  The streamed dimension is split into chunks (gridDim.y) so that >> 256 workgroups
// exist; chunk partials are combined by a second kernel in a FIXED order (no float atomics: results are bitwise
// reproducible run to run).
#include "common.h"
#include "fastexp.h"

#include <algorithm>
#include <system_error>
#include <thread>
#include <cstdlib>
#include <utility>

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 256;
#ifndef GINGR_PT
#define GINGR_PT 4
#endif
#define GINGR_PT_DEFAULT GINGR_PT

struct __attribute__((aligned(32))) P4 {
    double x, y, z, w;
};

// Table of the two CPD passes: 2^11 entries (16 KB of LDS), floor form (fastexp.h).  An 8192-entry table (byte offset by one SDWA
// shift) was measured in round 1 and lost to its LDS footprint; the floor form gets the one-instruction offset with 2048 entries.
constexpr int kTB = 11;
constexpr int kTabN = 1 << kTB;

// -DGINGR_STAMPS (diagnostic builds only: `make variant NAME=stamps DEFS=-DGINGR_STAMPS`, tools/stamps_shard.py): every wave of the
// two CPD pair loops leaves eight 64-bit words per launch in a device buffer -- the 100 MHz wall clock at entry / owned points and
// boxes ready / table barrier passed / first quarter staged / pair loop done / exit, the core-clock cycles of the whole wave, and
// its hardware id -- so that a one-round launch (a short row shard) can be taken apart per wave.  Nothing of it exists in the product build.
#ifdef GINGR_STAMPS
__device__ unsigned long long *g_stamp_buf = nullptr;   // [2 kernels][kStampWaves][8]
constexpr int kStampWaves = 1 << 15;
struct Stamps {
    unsigned long long *p;
    long long c0;
    __device__ __forceinline__ void begin(int kernel) {
        const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned w = wg * 4 + (threadIdx.x >> 6);
        p = (g_stamp_buf && w < kStampWaves) ? g_stamp_buf + ((size_t)kernel * kStampWaves + w) * 8 : nullptr;
        c0 = clock64();
        mark(0);
    }
    __device__ __forceinline__ void mark(int slot) {
        const unsigned long long t = wall_clock64();
        if (p && (threadIdx.x & 63) == 0) p[slot] = t;
    }
    __device__ __forceinline__ void end() {
        mark(5);
        if (p && (threadIdx.x & 63) == 0) {
            p[6] = (unsigned long long)(clock64() - c0);
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            p[7] = ((unsigned long long)xcc << 32) | hw;
        }
    }
};
#define STAMP_DECL Stamps stamps__;
#define STAMP_BEGIN(k) stamps__.begin(k);
#define STAMP(slot) stamps__.mark(slot);
#define STAMP_END stamps__.end();
#else
#define STAMP_DECL
#define STAMP_BEGIN(k)
#define STAMP(slot)
#define STAMP_END
#endif

// ---------------------------------------------------------------- exact-zero culling
// K_ij = 2^(c d2 / table size) is flushed to exactly +0 by v_ldexp_f64 once c*d2/size < -1076, i.e. d2 > 1491.7 sigma2.  When the
// bounding boxes of the owned block and of a streamed 256-point tile are farther apart than that (with margin: 1500
// sigma2), every pair of the tile pair contributes exactly +0 to every sum and the tile is skipped: bit-identical results.
// This only triggers when the points are spatially coherent (the fitter keeps model rows and targets in Morton order).
// 1500 * sigma2 * (-c), with -c = table size * log2(e) / (2 sigma2)
#define GINGR_CULL_SCALED(entries) (1084.0 * (double)(entries))
constexpr double kFineCullRatio = 16.0;  // fine culling once the flush radius is below a quarter of the largest possible distance

struct Box {
    double lo[3], hi[3];
};

__device__ __forceinline__ double uniform_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double box_gap2(const Box &a, const double *__restrict__ b /* lo[3], hi[3] */) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double g = fmax(fmax(a.lo[d] - b[3 + d], b[d] - a.hi[d]), 0.0);
        s = __builtin_fma(g, g, s);
    }
    return s;
}

// bounding boxes of the owned points (invalid slots excluded): `wave` = this wave's 64*PT points, return value = the whole
// workgroup's.  Both are wave-uniform and kept in scalar registers.
template <int PT>
__device__ __forceinline__ Box block_bbox(const double (&x)[PT], const double (&y)[PT], const double (&z)[PT],
                                          const bool (&ok)[PT], double *sh /* >= 6*4 doubles */, Box *wave_box) {
    double lo[3] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val()};
    double hi[3] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
#pragma unroll
    for (int t = 0; t < PT; ++t)
        if (ok[t]) {
            lo[0] = fmin(lo[0], x[t]); hi[0] = fmax(hi[0], x[t]);
            lo[1] = fmin(lo[1], y[t]); hi[1] = fmax(hi[1], y[t]);
            lo[2] = fmin(lo[2], z[t]); hi[2] = fmax(hi[2], z[t]);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fmin(lo[d], __shfl_xor(lo[d], off));
            hi[d] = fmax(hi[d], __shfl_xor(hi[d], off));
        }
    if (wave_box)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            wave_box->lo[d] = uniform_d(lo[d]);
            wave_box->hi[d] = uniform_d(hi[d]);
        }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            sh[wave * 6 + d] = lo[d];
            sh[wave * 6 + 3 + d] = hi[d];
        }
    __syncthreads();
    Box b;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        b.lo[d] = uniform_d(fmin(fmin(sh[d], sh[6 + d]), fmin(sh[12 + d], sh[18 + d])));
        b.hi[d] = uniform_d(fmax(fmax(sh[3 + d], sh[9 + d]), fmax(sh[15 + d], sh[21 + d])));
    }
    __syncthreads();
    return b;
}

// bounding box of a wave's owned points (invalid slots excluded), wave-uniform, in scalar registers.  In the CPD passes all four
// waves of a workgroup hold the SAME owned points, so this is also the workgroup's box -- computed redundantly, no LDS, and
// bit-identical in every wave (block-uniform branches may depend on it).
template <int PT>
__device__ __forceinline__ Box wave_bbox(const double (&x)[PT], const double (&y)[PT], const double (&z)[PT], const bool (&ok)[PT]) {
    double lo[3] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val()};
    double hi[3] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
#pragma unroll
    for (int t = 0; t < PT; ++t)
        if (ok[t]) {
            lo[0] = fmin(lo[0], x[t]); hi[0] = fmax(hi[0], x[t]);
            lo[1] = fmin(lo[1], y[t]); hi[1] = fmax(hi[1], y[t]);
            lo[2] = fmin(lo[2], z[t]); hi[2] = fmax(hi[2], z[t]);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fmin(lo[d], __shfl_xor(lo[d], off));
            hi[d] = fmax(hi[d], __shfl_xor(hi[d], off));
        }
    Box b;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        b.lo[d] = uniform_d(lo[d]);
        b.hi[d] = uniform_d(hi[d]);
    }
    return b;
}

// bounding box of each owned slot t over the wave: the 64 consecutive points {base + 64 t + lane} (a quarter k-d leaf); wave
// The streamed side of an all-pairs launch is cut into chunks (one workgroup per owned block and chunk): n_big chunks of len_big
// points, then chunks of len_tail points.  Long chunks first and short ones last shorten the tail of the launch (the last
// workgroups to be dispatched are the cheap ones) without multiplying the per-chunk partials; len_tail == len_big is the uniform cut.
struct ChunkPlan {
    int64_t len_big, len_tail;
    int32_t n_big;
    int32_t fair = 0;  // one launch round: waves lower their issue priority as they advance (fair_priority)
    __host__ __device__ void range(int64_t y, int64_t n, int64_t *b, int64_t *e) const {
        const int64_t lo = y < n_big ? y * len_big : (int64_t)n_big * len_big + (y - n_big) * len_tail;
        const int64_t hi = lo + (y < n_big ? len_big : len_tail);
        *b = lo;
        *e = hi < n ? hi : n;
    }
    __host__ int chunks(int64_t n) const {
        const int64_t head = (int64_t)n_big * len_big;
        if (head >= n) return (int)((n + len_big - 1) / len_big);
        return n_big + (int)((n - head + len_tail - 1) / len_tail);
    }
};

// uniform, kept in scalar registers (measured faster than a round trip through LDS, spills included)
template <int PT>
__device__ __forceinline__ void slot_boxes(const double (&x)[PT], const double (&y)[PT], const double (&z)[PT],
                                           const bool (&ok)[PT], Box (&sb)[PT]) {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        double lo[3] = {ok[t] ? x[t] : __builtin_huge_val(), ok[t] ? y[t] : __builtin_huge_val(), ok[t] ? z[t] : __builtin_huge_val()};
        double hi[3] = {ok[t] ? x[t] : -__builtin_huge_val(), ok[t] ? y[t] : -__builtin_huge_val(), ok[t] ? z[t] : -__builtin_huge_val()};
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                lo[d] = fmin(lo[d], __shfl_xor(lo[d], off));
                hi[d] = fmax(hi[d], __shfl_xor(hi[d], off));
            }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            sb[t].lo[d] = uniform_d(lo[d]);
            sb[t].hi[d] = uniform_d(hi[d]);
        }
    }
}

// boxes[tile] = {lo[3], hi[3]} of the points [tile*256, tile*256+256) of a cloud, followed (at boxes + 6 * ntiles) by the boxes
// of its four 64-point quarters, [tile*4 + q]: the k-d leaf order makes those compact too (finer exact-zero culling).
// With slot != nullptr also slot = max over the cloud of |coordinate - ctr| (atomic max on the bit pattern of a non-negative
// double: order independent, deterministic); the slot must have been zeroed by an EARLIER launch on the stream.
constexpr int kBoxTilesPerBlock = 8;
// One workgroup handles kBoxTilesPerBlock consecutive tiles (wave q the quarter q of each), so the launch ends with one atomic per
// 2048 points: agent-scope atomics on one word are served at the memory side, one after the other (~0.13 us each: with one tile per
// workgroup the 196 tiles of 50k points took 26 us, all of it the atomics).  The 64-lane minima / maxima are not butterflies of
// cross-lane shuffles (6 dependent LDS round trips per quantity: 14 us for the 48 quantities of a wave) but column scans: the wave
// parks its values in LDS, [quantity][lane], and lane j < 48 scans the 64 entries of ITS quantity, starting at entry j so that the
// lanes of one read sit in different banks.  fmin ignores NaN like the butterfly did (a NaN point never widens a box).
__global__ __launch_bounds__(256) void tile_bbox_kernel(Cloud c, double *__restrict__ boxes, const double *__restrict__ ctr,
                                                        double *__restrict__ slot, int64_t ntiles) {
    constexpr int Q = kBoxTilesPerBlock * 3;          // quantities per wave: (tile, coordinate)
    __shared__ double park[4][Q][64];
    __shared__ double sh[kBoxTilesPerBlock][4][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile0 = (int64_t)blockIdx.x * kBoxTilesPerBlock;
    double v[Q];
#pragma unroll
    for (int t = 0; t < kBoxTilesPerBlock; ++t) {  // all loads in flight; an absent point is NaN: fmin / fmax skip it
        const int64_t i = (tile0 + t) * kTile + threadIdx.x;
        const bool ok = i < c.n;
        v[3 * t] = ok ? c.x[i] : __builtin_nan("");
        v[3 * t + 1] = ok ? c.y[i] : __builtin_nan("");
        v[3 * t + 2] = ok ? c.z[i] : __builtin_nan("");
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) park[wave][k][lane] = v[k];
    __builtin_amdgcn_wave_barrier();  // wave-local data: LDS serves one wave's accesses in order
    if (lane < 2 * Q) {
        const int k = lane >> 1;
        const double sgn = (lane & 1) ? -1.0 : 1.0;  // odd lanes: maximum as -min(-x)
        double m = __builtin_huge_val();
#pragma unroll 8
        for (int e = 0; e < 64; ++e) m = fmin(m, sgn * park[wave][k][(e + lane) & 63]);
        const int t = k / 3, d = k % 3;
        const double r = sgn * m;  // (+huge, -huge) for a quarter without points, as before
        if (tile0 + t < ntiles) {
            boxes[ntiles * 6 + ((tile0 + t) * 4 + wave) * 6 + (lane & 1) * 3 + d] = r;
            sh[t][wave][(lane & 1) * 3 + d] = r;
        }
    }
    __syncthreads();
    double m = 0.0;
    if (threadIdx.x < kBoxTilesPerBlock * 3) {  // thread (t, d): both bounds of coordinate d of tile t
        const int t = threadIdx.x / 3, d = threadIdx.x % 3;
        if (tile0 + t < ntiles) {
            const double lo = fmin(fmin(sh[t][0][d], sh[t][1][d]), fmin(sh[t][2][d], sh[t][3][d]));
            const double hi = fmax(fmax(sh[t][0][3 + d], sh[t][1][3 + d]), fmax(sh[t][2][3 + d], sh[t][3][3 + d]));
            boxes[(tile0 + t) * 6 + d] = lo;
            boxes[(tile0 + t) * 6 + 3 + d] = hi;
            const double cc = ctr ? ctr[d] : 0.0;
            m = fmax(fabs(lo - cc), fabs(hi - cc));
        }
    }
    if (slot && wave == 0) {  // kBoxTilesPerBlock * 3 <= 64: the candidates all sit in wave 0
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
        if (lane == 0) {
            // non-negative doubles order like their bit patterns; only a value above what is already there needs the atomic
            const unsigned long long mb = __builtin_bit_cast(unsigned long long, m);
            if (mb > __hip_atomic_load(reinterpret_cast<unsigned long long *>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(reinterpret_cast<unsigned long long *>(slot), mb);
        }
    }
}

// Lane-parallel exact-zero culling for one wave: the tile parts of a chunk (part p = box tile t0 + p, clipped to the chunk
// [c0, c1)) are tested 64 at a time -- lane l tests part pbase + l against the box of the wave's quarter q of that part -- and the
// outcome comes back as wave-uniform bit masks, so the walk over the parts (PartWalk::next) is scalar code with no per-part box
// arithmetic (uniform float64 arithmetic still costs full VALU instructions on gfx950: 12 per test, ~1.5 % of a dense pass).
//   act     bit l: the wave's quarter of part pbase + l exists and can receive non-zeros
//   slot[t] (FINE) bit l: owned slot t can receive non-zeros from it
// bad (nullable): per box tile, non-zero = never cull (a non-finite 1/den: 0 * inf must stay NaN).  boxes == nullptr: nothing is culled.
template <int PT, bool FINE>
struct PartWalk {
    int64_t c0, c1, t0;
    int nparts, pbase, q;
    unsigned long long act, slot[PT];
    const double *sub;      // quarter boxes, [6] per 64 points of the streamed cloud
    const int32_t *bad;
    double negc;

    __device__ __forceinline__ void init(int64_t c0_, int64_t c1_, int q_, const double *sub_, const int32_t *bad_, double negc_) {
        c0 = c0_;
        c1 = c1_;
        q = q_;
        sub = sub_;
        bad = bad_;
        negc = negc_;
        t0 = c0 / kTile;
        nparts = c1 > c0 ? (int)((c1 - 1) / kTile - t0 + 1) : 0;
        pbase = 0;
        act = 0;
    }
    __device__ __forceinline__ void bounds(int p, int64_t *ib, int64_t *ie) const {
        const int64_t a = (t0 + p) * kTile, b = a + kTile;
        *ib = a > c0 ? a : c0;
        *ie = b < c1 ? b : c1;
    }
    __device__ __forceinline__ void batch(const Box &own, const Box (&sown)[PT]) {
        const int lane = threadIdx.x & 63;
        const int p = pbase + lane;
        int64_t ib, ie;
        bounds(p, &ib, &ie);
        const bool has = p < nparts && q * 64 < (int)(ie - ib);
        bool on = has;
        bool son[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) son[t] = has;
        if (has && sub && !(bad && bad[(t0 + p)])) {
            const double *qbox = sub + (ib / 64 + q) * 6;
            if (FINE) {
                on = false;
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    son[t] = !(box_gap2(sown[t], qbox) * negc > GINGR_CULL_SCALED(kTabN));  // NaN boxes are never culled
                    on = on || son[t];
                }
            } else {
                on = !(box_gap2(own, qbox) * negc > GINGR_CULL_SCALED(kTabN));
            }
        }
        act = __ballot(on);
        if (FINE) {
#pragma unroll
            for (int t = 0; t < PT; ++t) slot[t] = __ballot(son[t]);
        }
        pbase += 64;
    }
    // next part with work for this wave: false when the chunk is exhausted
    __device__ __forceinline__ bool next(const Box &own, const Box (&sown)[PT], int64_t *ib, int64_t *ie, unsigned *mask) {
        while (act == 0) {
            if (pbase >= nparts) return false;
            batch(own, sown);
        }
        const int l = __builtin_ctzll(act);
        act &= act - 1;
        bounds(pbase - 64 + l, ib, ie);
        unsigned m = (1u << PT) - 1u;
        if (FINE) {
            m = 0;
#pragma unroll
            for (int t = 0; t < PT; ++t) m |= (unsigned)((slot[t] >> l) & 1ull) << t;
        }
        *mask = m;
        return true;
    }
};

// One launch round (every workgroup resident from the start, e.g. the row shard of an 8-rank job): the SIMD arbiter favours the
// oldest wave, so the four waves of a SIMD finish one after the other (time stamps: 53 / 80 / 107 / 141 us of a 148 us launch) and
// the last one runs alone at the single-wave rate of one instruction per ~5.3 cycles.  Lowering a wave's priority as it advances
// keeps the four abreast until the end.  Not used when rounds overlap: there the stagger hides the prologues.
__device__ __forceinline__ void fair_priority(int64_t done, int64_t total) {
    const int64_t f = total > 0 ? (4 * done) / total : 3;
    if (f <= 0)
        __builtin_amdgcn_s_setprio(3);
    else if (f == 1)
        __builtin_amdgcn_s_setprio(2);
    else if (f == 2)
        __builtin_amdgcn_s_setprio(1);
    else
        __builtin_amdgcn_s_setprio(0);
}

// ---------------------------------------------------------------- pass 1: column sums of K
// Tile loops.  [j0, j1) is a range of tile entries (the whole tile or one 64-point quarter); with MASKED only the owned slots
// t whose bit is set in `mask` (wave-uniform) are updated -- the others are known to receive exact zeros from this range.
template <int PT, bool CLAMP, bool MASKED>
__device__ __forceinline__ void colsum_tile(const P4 *tile, int j0, int j1, unsigned mask, const double (&x)[PT],
                                            const double (&y)[PT], const double (&z)[PT], double (&acc)[PT], double c, double lim,
                                            const double *T) {
#pragma unroll 2
    for (int ii = j0; ii < j1; ++ii) {
        const P4 p = tile[ii];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            if (MASKED && !((mask >> t) & 1u)) continue;
            const double dx = x[t] - p.x, dy = y[t] - p.y, dz = z[t] - p.z;
            double d2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
            if (CLAMP) d2 = fmin(d2, lim);
            acc[t] += fastexp2_floor_scaled(d2, c, T);
        }
    }
}

// Accuracy-guarded fast path: exponent argument from the norm expansion
//     t = c|x - y|^2 = c|x~|^2 + c|y~|^2 - 2c x~.y~     (x~, y~ centred on the target centroid)
// = 1 add + 3 FMAs per pair instead of the 6 + 1 instructions of the difference form.  Its cancellation error is a
// relative error of K_ij of about 7.7e-16 * R^2 / (2 sigma2) (R = cloud radius about the centroid), so it is used only
// while that bound stays below kExpandTol; otherwise the kernels fall back to the exact differences (wave-uniform choice).
constexpr double kExpandTol = 1e-12;

__device__ __forceinline__ bool use_expansion(double rmax_centered, double c) {
    // R^2 <= 3 rmax^2;  R^2 / (2 sigma2) = R^2 |c| ln2 / table size
    const double ratio = 3.0 * rmax_centered * rmax_centered * (-c) * (0.69314718055994530942 / kTabN);
    return ratio * 7.7e-16 < kExpandTol;
}

// Owned-side constants of the expansion form.  n = c|x~|^2 of an owned point is split into its nearest integer, folded into the
// magic constant of the range reduction (mg = MAGIC + rint(n): the table index and the exponent then come out for A + rint(n)
// although only A is ever added), and the fraction n - rint(n) in [-1/2, 1/2], which is a constant FACTOR 2^(frac/2048) of every
// K of that owned point and is applied once to the finished sums (expand_owned_scale).  One add per pair less.
__device__ __forceinline__ double expand_owned_magic(double n, double *frac) {
    const double ni = __builtin_rint(n);
    *frac = n - ni;  // exact
    return GINGR_EXP_MAGIC8 + ni;
}
__device__ __forceinline__ double expand_owned_scale(double frac) { return exp2(frac * (1.0 / kTabN)); }

// tile entries: (-2c y~, c|y~|^2); owned: x~ and mg = MAGIC + rint(c|x~|^2)   (float64 rounding mode: toward -inf)
template <int PT, bool MASKED>
__device__ __forceinline__ void colsum_tile_expand(const P4 *tile, int j0, int j1, unsigned mask, const double (&x)[PT],
                                                   const double (&y)[PT], const double (&z)[PT], const double (&mg)[PT],
                                                   double (&acc)[PT], const double *T) {
#pragma unroll 2
    for (int ii = j0; ii < j1; ++ii) {
        const P4 p = tile[ii];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            if (MASKED && !((mask >> t) & 1u)) continue;
            const double A = __builtin_fma(z[t], p.z, __builtin_fma(y[t], p.y, __builtin_fma(x[t], p.x, p.w)));
            acc[t] += fastexp2_floor_core(A + mg[t], __builtin_amdgcn_fract(A), T);
        }
    }
}

// Work split of both CPD passes.  A workgroup owns 64*PT points (one 256-point k-d leaf at PT = 4) and ALL FOUR of its waves hold
// the same owned points in registers; of every streamed 256-point tile wave q takes the 64-point quarter q.  The four waves'
// accumulators are added in LDS in a fixed order ((w0 + w1) + (w2 + w3)) before anything goes to memory, so a workgroup covers a
// chunk four times as long as it would with one accumulator set per wave and the launch writes a quarter of the chunk partials
// (50k x 50k: 25 instead of 98 chunks; 40 MB instead of 157 MB of row-statistics partials) for the same number of workgroups.
//
// FINE selects the variant with the quarter-tile x slot culling; it pays when the cull radius is small against the clouds (the
// regime, a property of sigma2 and the cloud extents that only the device knows).  The plain variant culls per (workgroup, tile)
// and per (wave's quarter).  Both variants give bit-identical results, so the host launches ONE of them, picked from the regime
// word the previous launches left in pinned host memory (regime_out; possibly stale -- that only costs time).  Two kernels
// rather than one with a switch: sharing one kernel cost the plain regime 7 % in the row-statistics pass.
template <int PT, bool FINE>
__global__ __launch_bounds__(kBlock) void cpd_colsum_kernel(Cloud fit, Cloud tgt, const double *__restrict__ sigma2,
                                                            const double *__restrict__ aux,
                                                            const double *__restrict__ fit_boxes, ChunkPlan plan,
                                                            double *__restrict__ partial, int32_t *regime_out) {
    __shared__ double T[kTabN];
    __shared__ P4 tile[kTile];
    __shared__ double sred[4][64 * PT];  // the waves' accumulators, combined in the epilogue
    __shared__ double sfrac[64 * PT];    // per owned point: fraction of c|x~|^2 (expansion form), parked until the epilogue
    STAMP_DECL
    STAMP_BEGIN(0)
    // Everything the prologue needs from memory is requested up front -- the exponential table, the owned points, the scalars -- so
    // that a workgroup of a one-round launch (a short row shard, a small cloud) waits for ONE round trip, not for a chain of them.
    FloorTableRegs trom;
    fastexp_floor_table_fetch256(trom);
    const double s2in = sigma2[0], aux0 = aux[0], aux1 = aux[1], cx = aux[2], cy = aux[3], cz = aux[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave's number, provably uniform: loop bounds and LDS bases stay scalar
    // the workgroup's 64*PT CONSECUTIVE points (one 256-point k-d leaf at PT = 4: a compact box), the same in every wave
    const int64_t jbase = (int64_t)blockIdx.x * (64 * PT) + lane;
    double x[PT], y[PT], z[PT], n[PT], acc[PT];
    bool okv[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int64_t j = jbase + (int64_t)t * 64;
        const bool ok = j < tgt.n;
        okv[t] = ok;
        x[t] = ok ? tgt.x[j] : 0.0;
        y[t] = ok ? tgt.y[j] : 0.0;
        z[t] = ok ? tgt.z[j] : 0.0;
    }
    // ... and the first quarter this wave will most likely stage (the first tile part of its chunk: the walk below confirms it or
    // picks another one when that part is culled)
    int64_t i0, i1;
    plan.range(blockIdx.y, fit.n, &i0, &i1);
    const int q0 = q * 64;
    double lx = 0.0, ly = 0.0, lz = 0.0;  // the staged point of the quarter about to be computed (in flight during the previous one)
    const int64_t spec_ib = i0;
    {
        const int64_t e = min((i0 / kTile + 1) * kTile, i1), i = i0 + q0 + lane;
        if (i < e) {
            lx = fit.x[i];
            ly = fit.y[i];
            lz = fit.z[i];
        }
    }
    const double c = fastexp_scale_for_variance<kTB>(2.0 * s2in);
    const double am = aux0 + aux1;
    // regime of the fine culling: the zero-flush radius is well inside the clouds' extent (3 am^2 bounds every squared distance)
    if (regime_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        *regime_out = (fit_boxes && 3.0 * am * am * (-c) > kFineCullRatio * GINGR_CULL_SCALED(kTabN)) ? 1 : 0;
        __threadfence_system();
    }
    const bool clamp = fastexp_needs_clamp(3.0 * am * am, c);               // wave-uniform
    const bool expand = use_expansion(fmax(aux0, aux1), c);                 // wave-uniform
    const double lim = fastexp_d2_limit<kTB>(c);
    const double m2c = -2.0 * c;
    fastexp_floor_table_park256(T, trom);
    const Box own = wave_bbox<PT>(x, y, z, okv);  // raw coordinates, before any centring
    constexpr unsigned kAllSlots = (1u << PT) - 1u;
    Box sown[PT];  // per owned slot (64 consecutive points)
    if (FINE) slot_boxes<PT>(x, y, z, okv, sown);
    const double *fit_sub = fit_boxes ? fit_boxes + ((fit.n + kTile - 1) / kTile) * 6 : nullptr;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (expand) {
            x[t] -= cx;
            y[t] -= cy;
            z[t] -= cz;
        }
        double fr;
        n[t] = expand_owned_magic(c * __builtin_fma(z[t], z[t], __builtin_fma(y[t], y[t], x[t] * x[t])), &fr);
        if (q == 0) sfrac[t * 64 + lane] = fr;
        acc[t] = 0.0;
    }
    STAMP(1)
    __syncthreads();       // the exponential table (filled by all four waves) and sfrac are complete
    STAMP(2)
    fastexp_round_down();  // the floor form of the exponential needs it; every float64 result up to the epilogue rounds down
    // A chunk starts and ends on 64-point quarters, not necessarily on tiles: every step handles the part of ONE box tile that lies
    // inside the chunk, so the tile / quarter boxes apply unchanged.  Of each part wave q takes the quarter q: it stages those 64
    // entries itself into its own slice of `tile` and is the only reader, so the pair loop has no workgroup barrier -- the waves
    // run decoupled -- and the loads of the wave's NEXT quarter are issued before the pairs of the current one are computed.
    struct Work {
        int64_t ib, ie;
        unsigned mask;
        bool valid;
    };
    PartWalk<PT, FINE> walk;
    walk.init(i0, i1, q, fit_boxes ? fit_sub : nullptr, nullptr, -c);
    auto find = [&]() {  // next tile part of the chunk in which this wave's quarter can receive non-zeros
        Work w{0, 0, kAllSlots, false};
        w.valid = walk.next(own, sown, &w.ib, &w.ie, &w.mask);
        return w;
    };
    auto issue = [&](const Work &w) {
        const int64_t i = w.ib + q0 + lane;
        if (i < w.ie) {
            lx = fit.x[i];
            ly = fit.y[i];
            lz = fit.z[i];
        }
    };
    Work cur = find();
    if (cur.valid && cur.ib != spec_ib) issue(cur);  // (otherwise the prologue's request was the right one)
#ifdef GINGR_STAMPS
    bool first__ = true;
#endif
    while (cur.valid) {
        if (plan.fair) fair_priority(cur.ib - i0, i1 - i0);
        __builtin_amdgcn_wave_barrier();  // (compiler fence) the previous quarter's reads are issued before the slice is rewritten
        if (cur.ib + q0 + lane < cur.ie) {
            if (expand) {
                const double fx = lx - cx, fy = ly - cy, fz = lz - cz;
                tile[q0 + lane] = P4{m2c * fx, m2c * fy, m2c * fz, c * __builtin_fma(fz, fz, __builtin_fma(fy, fy, fx * fx))};
            } else {
                tile[q0 + lane] = P4{lx, ly, lz, 0.0};
            }
        }
        __builtin_amdgcn_wave_barrier();  // LDS serves one wave's accesses in order: its reads below see its own writes
#ifdef GINGR_STAMPS
        if (first__) { STAMP(3) first__ = false; }
#endif
        const Work nxt = find();
        if (nxt.valid) issue(nxt);
        const int q1 = min((int)(cur.ie - cur.ib), q0 + 64);
        const unsigned mask = cur.mask;
        if (expand) {
            if (!FINE || mask == kAllSlots)
                colsum_tile_expand<PT, false>(tile, q0, q1, mask, x, y, z, n, acc, T);
            else
                colsum_tile_expand<PT, true>(tile, q0, q1, mask, x, y, z, n, acc, T);
        } else if (clamp) {
            if (!FINE || mask == kAllSlots)
                colsum_tile<PT, true, false>(tile, q0, q1, mask, x, y, z, acc, c, lim, T);
            else
                colsum_tile<PT, true, true>(tile, q0, q1, mask, x, y, z, acc, c, lim, T);
        } else {
            if (!FINE || mask == kAllSlots)
                colsum_tile<PT, false, false>(tile, q0, q1, mask, x, y, z, acc, c, lim, T);
            else
                colsum_tile<PT, false, true>(tile, q0, q1, mask, x, y, z, acc, c, lim, T);
        }
        cur = nxt;
    }
    fastexp_round_nearest();
    STAMP(4)
#pragma unroll
    for (int t = 0; t < PT; ++t) sred[q][t * 64 + lane] = acc[t];
    __syncthreads();
    for (int p = tid; p < 64 * PT; p += kBlock) {
        double v = (sred[0][p] + sred[1][p]) + (sred[2][p] + sred[3][p]);
        if (expand) v *= expand_owned_scale(sfrac[p]);
        const int64_t j = (int64_t)blockIdx.x * (64 * PT) + p;
        if (j < tgt.n) partial[(int64_t)blockIdx.y * tgt.n + j] = v;
    }
    STAMP_END
}

// centroid of a cloud into out[0..2] (single workgroup, fixed order)
__global__ __launch_bounds__(1024) void cloud_centroid_kernel(Cloud c, double *__restrict__ out) {
    __shared__ double sh[3][1024];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int64_t i = threadIdx.x; i < c.n; i += 1024) {
        sx += c.x[i];
        sy += c.y[i];
        sz += c.z[i];
    }
    sh[0][threadIdx.x] = sx;
    sh[1][threadIdx.x] = sy;
    sh[2][threadIdx.x] = sz;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int d = 0; d < 3; ++d) sh[d][threadIdx.x] += sh[d][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        const double v = sh[threadIdx.x][0] / (double)c.n;
        out[threadIdx.x] = (v == v && fabs(v) < 1e300) ? v : 0.0;  // a non-finite centroid would poison every pair
    }
}

// slot = max over the cloud of |x - cx|, |y - cy|, |z - cz| (ctr may be nullptr = origin).  Atomic max on the bit
// pattern of a non-negative double: order independent, hence deterministic.  The slot must be zeroed before the launch.
__global__ __launch_bounds__(256) void cloud_absmax_kernel(Cloud c, const double *__restrict__ ctr, double *__restrict__ slot) {
    __shared__ double sh[256];
    const double cx = ctr ? ctr[0] : 0.0, cy = ctr ? ctr[1] : 0.0, cz = ctr ? ctr[2] : 0.0;
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < c.n; i += (int64_t)gridDim.x * 256)
        m = fmax(m, fmax(fabs(c.x[i] - cx), fmax(fabs(c.y[i] - cy), fabs(c.z[i] - cz))));
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double v = sh[0];
        if (!(v == v)) v = __builtin_huge_val();  // NaN coordinates: force the clamped exact path
        atomicMax(reinterpret_cast<unsigned long long *>(slot), __builtin_bit_cast(unsigned long long, v));
    }
}

// out[j] = sum over chunks (ascending) of partial[chunk][j]
__global__ void chunk_reduce_kernel(const double *__restrict__ partial, int nchunks, int64_t n, double *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * n + j];
    out[j] = s;
}

// block-wide fixed-order sum; result valid in thread 0
template <int NT>
__device__ __forceinline__ double block_sum(double v, double *sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int s = NT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    return sh[0];
}

// den[j] = colsum[j] + c ; inv_den ; Pt1 ; per-block partial of xPx = sum_j Pt1_j |x_j|^2.
// Always launched with kScalarBlocks workgroups; part[0..kScalarBlocks) receives the xPx partials.
constexpr int kScalarBlocks = 256;

// chunk_partial != nullptr (single shard): the column sums are still spread over the nchunks chunk partials of the column-sum
// pass and are added up here, in ascending chunk order like chunk_reduce_kernel (one launch and one pass over den less).
__global__ __launch_bounds__(256) void cpd_den_finalize_kernel(Cloud tgt, const double *__restrict__ sigma2, double w,
                                                               double m_over_n, double *__restrict__ den,
                                                               double *__restrict__ inv_den, double *__restrict__ Pt1,
                                                               int32_t *__restrict__ tile_bad, double *__restrict__ part,
                                                               double *__restrict__ scalars,
                                                               const double *__restrict__ chunk_partial, int nchunks) {
    __shared__ double sh[256];
    const double s2 = sigma2[0];
    // c = w/(1-w) * (2 pi sigma2)^(3/2) * (M/N)     CPD.scala:69-70
    const double c = w / (1.0 - w) * pow(2.0 * 3.14159265358979323846 * s2, 1.5) * m_over_n;
    double xpx = 0.0;
    // one 256-point tile per workgroup and pass (block stride = tile size), so tile_bad needs no clearing beforehand
    for (int64_t jt = (int64_t)blockIdx.x * 256; jt < tgt.n; jt += (int64_t)kScalarBlocks * 256) {
        const int64_t j = jt + threadIdx.x;
        int bad = 0;
        if (j < tgt.n) {
        double colsum;
        if (chunk_partial) {
            colsum = 0.0;
#pragma unroll 8
            for (int ch = 0; ch < nchunks; ++ch) colsum += chunk_partial[(int64_t)ch * tgt.n + j];  // loads ahead, adds in order
        } else {
            colsum = den[j];
        }
        const double d = colsum + c;
        const double inv = 1.0 / d;
        const double pt1 = colsum / d;
        den[j] = d;
        inv_den[j] = inv;
        Pt1[j] = pt1;
        bad = !(fabs(inv) <= 1.79769313486231570815e308);  // never cull this tile
        const double xx = tgt.x[j], yy = tgt.y[j], zz = tgt.z[j];
        xpx += pt1 * (xx * xx + yy * yy + zz * zz);
        }
        const int any_bad = __syncthreads_or(bad);
        if (tile_bad && threadIdx.x == 0) tile_bad[jt / kTile] = any_bad;
    }
    const double tot = block_sum<256>(xpx, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = tot;
        if (blockIdx.x == 0) scalars[5] = c;
    }
}

// ---------------------------------------------------------------- pass 2: row statistics
// tile entries: (x, y, z, 1/den); tw entries: (x, y, z)/den, so that P1 and P.X are four FMAs on K_ij (the product
// K * (x/den) instead of (K/den) * x: one rounding placed differently, one instruction less per pair)
template <int PT, bool CLAMP, bool MASKED>
__device__ __forceinline__ void rowstats_tile(const P4 *tile, const P4 *tw, int j0, int j1, unsigned mask,
                                              const double (&x)[PT], const double (&y)[PT], const double (&z)[PT],
                                              double (&a1)[PT], double (&ax)[PT], double (&ay)[PT], double (&az)[PT], double c,
                                              double lim, const double *T) {
#pragma unroll 2
    for (int jj = j0; jj < j1; ++jj) {
        const P4 p = tile[jj];
        const P4 q = tw[jj];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            if (MASKED && !((mask >> t) & 1u)) continue;
            const double dx = p.x - x[t], dy = p.y - y[t], dz = p.z - z[t];
            double d2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
            if (CLAMP) d2 = fmin(d2, lim);
            const double k = fastexp2_floor_scaled(d2, c, T);
            a1[t] = __builtin_fma(k, p.w, a1[t]);
            ax[t] = __builtin_fma(k, q.x, ax[t]);
            ay[t] = __builtin_fma(k, q.y, ay[t]);
            az[t] = __builtin_fma(k, q.z, az[t]);
        }
    }
}

// expansion form of pass 2: tile entries (-2c x~, c|x~|^2) + 1/den; owned y~ and n = c|y~|^2.  P.X is accumulated as
// sum_j p a_j with a_j = -2c x~_j and rescaled once at the end: PX = ctr*P1 - (sum_j p a_j) / (2c).
// tw entries: (a_j / den_j, 1 / den_j)
template <int PT, bool MASKED>
__device__ __forceinline__ void rowstats_tile_expand(const P4 *tile, const P4 *tw, int j0, int j1, unsigned mask,
                                                     const double (&x)[PT], const double (&y)[PT], const double (&z)[PT],
                                                     const double (&mg)[PT], double (&a1)[PT], double (&ax)[PT], double (&ay)[PT],
                                                     double (&az)[PT], const double *T) {
#pragma unroll 2
    for (int jj = j0; jj < j1; ++jj) {
        const P4 p = tile[jj];
        const P4 q = tw[jj];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            if (MASKED && !((mask >> t) & 1u)) continue;
            const double A = __builtin_fma(z[t], p.z, __builtin_fma(y[t], p.y, __builtin_fma(x[t], p.x, p.w)));
            const double k = fastexp2_floor_core(A + mg[t], __builtin_amdgcn_fract(A), T);
            a1[t] = __builtin_fma(k, q.w, a1[t]);
            ax[t] = __builtin_fma(k, q.x, ax[t]);
            ay[t] = __builtin_fma(k, q.y, ay[t]);
            az[t] = __builtin_fma(k, q.z, az[t]);
        }
    }
}

// __launch_bounds__'s second argument (waves per SIMD): with four points per thread the kernel cannot hold four waves anyway; telling
// the compiler that three are enough lets it use up to 168 registers instead of parking values in accumulator registers inside
// the pair loop (17 extra v_accvgpr moves per 8 pairs at the default heuristic).
template <int PT, bool FINE>  // work split, FINE / regime_out: see cpd_colsum_kernel
__global__ __launch_bounds__(kBlock, (PT >= 4 ? 3 : 4)) void cpd_rowstats_kernel(Cloud fit, Cloud tgt, const double *__restrict__ sigma2,
                                                              const double *__restrict__ aux,
                                                              const double *__restrict__ inv_den,
                                                              const double *__restrict__ tgt_boxes,
                                                              const int32_t *__restrict__ tile_bad, ChunkPlan plan,
                                                              double *__restrict__ partial, int32_t *regime_out) {
    // one LDS block: [T | tile | tw] during the pair loop, the waves' accumulators [4 waves][4 planes][64 PT] in the epilogue
    constexpr int kRed = 4 * 4 * 64 * PT;
    constexpr int kLoop = kTabN + 4 * kTile + 4 * kTile;
    __shared__ double smem[kRed > kLoop ? kRed : kLoop];
    __shared__ double sfrac[64 * PT];  // see cpd_colsum_kernel
    double *T = smem;
    P4 *tile = reinterpret_cast<P4 *>(smem + kTabN);
    P4 *tw = reinterpret_cast<P4 *>(smem + kTabN + 4 * kTile);
    STAMP_DECL
    STAMP_BEGIN(1)
    FloorTableRegs trom;  // table, owned points, scalars and the first quarter are requested together: see cpd_colsum_kernel
    fastexp_floor_table_fetch256(trom);
    const double s2in = sigma2[0], aux0 = aux[0], aux1 = aux[1], cx = aux[2], cy = aux[3], cz = aux[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave's number, provably uniform: loop bounds and LDS bases stay scalar
    const int64_t ibase = (int64_t)blockIdx.x * (64 * PT) + lane;
    double x[PT], y[PT], z[PT], n[PT], a1[PT], ax[PT], ay[PT], az[PT];
    bool okv[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int64_t i = ibase + (int64_t)t * 64;
        const bool ok = i < fit.n;
        okv[t] = ok;
        x[t] = ok ? fit.x[i] : 0.0;
        y[t] = ok ? fit.y[i] : 0.0;
        z[t] = ok ? fit.z[i] : 0.0;
    }
    int64_t j0, j1;
    plan.range(blockIdx.y, tgt.n, &j0, &j1);
    const int q0 = q * 64;
    double lx = 0.0, ly = 0.0, lz = 0.0, linv = 0.0;
    const int64_t spec_jb = j0;
    {
        const int64_t e = min((j0 / kTile + 1) * kTile, j1), j = j0 + q0 + lane;
        if (j < e) {
            lx = tgt.x[j];
            ly = tgt.y[j];
            lz = tgt.z[j];
            linv = inv_den[j];
        }
    }
    const double c = fastexp_scale_for_variance<kTB>(2.0 * s2in);
    const double am = aux0 + aux1;
    if (regime_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        *regime_out = (tgt_boxes && 3.0 * am * am * (-c) > kFineCullRatio * GINGR_CULL_SCALED(kTabN)) ? 1 : 0;
        __threadfence_system();
    }
    const bool clamp = fastexp_needs_clamp(3.0 * am * am, c);               // wave-uniform
    const bool expand = use_expansion(fmax(aux0, aux1), c);                 // wave-uniform
    const double lim = fastexp_d2_limit<kTB>(c);
    const double m2c = -2.0 * c;
    fastexp_floor_table_park256(T, trom);
    const Box own = wave_bbox<PT>(x, y, z, okv);  // raw coordinates, before any centring; identical in every wave
    constexpr unsigned kAllSlots = (1u << PT) - 1u;
    Box sown[PT];  // per owned slot (64 consecutive points)
    if (FINE) slot_boxes<PT>(x, y, z, okv, sown);
    const double *tgt_sub = tgt_boxes ? tgt_boxes + ((tgt.n + kTile - 1) / kTile) * 6 : nullptr;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (expand) {
            x[t] -= cx;
            y[t] -= cy;
            z[t] -= cz;
        }
        double fr;
        n[t] = expand_owned_magic(c * __builtin_fma(z[t], z[t], __builtin_fma(y[t], y[t], x[t] * x[t])), &fr);
        if (q == 0) sfrac[t * 64 + lane] = fr;
        a1[t] = ax[t] = ay[t] = az[t] = 0.0;
    }
    STAMP(1)
    __syncthreads();       // the exponential table (filled by all four waves) and sfrac are complete
    STAMP(2)
    fastexp_round_down();  // see cpd_colsum_kernel
    // one box tile (or the part of it inside the chunk) per step, wave q on quarter q, next quarter's loads in flight: see
    // cpd_colsum_kernel
    struct Work {
        int64_t jb, je;
        unsigned mask;
        bool valid;
    };
    PartWalk<PT, FINE> walk;
    walk.init(j0, j1, q, tgt_boxes ? tgt_sub : nullptr, tgt_boxes ? tile_bad : nullptr, -c);
    auto find = [&]() {
        Work w{0, 0, kAllSlots, false};
        w.valid = walk.next(own, sown, &w.jb, &w.je, &w.mask);
        return w;
    };
    auto issue = [&](const Work &w) {
        const int64_t j = w.jb + q0 + lane;
        if (j < w.je) {
            lx = tgt.x[j];
            ly = tgt.y[j];
            lz = tgt.z[j];
            linv = inv_den[j];
        }
    };
    Work cur = find();
    if (cur.valid && cur.jb != spec_jb) issue(cur);  // (otherwise the prologue's request was the right one)
#ifdef GINGR_STAMPS
    bool first__ = true;
#endif
    while (cur.valid) {
        if (plan.fair) fair_priority(cur.jb - j0, j1 - j0);
        __builtin_amdgcn_wave_barrier();
        if (cur.jb + q0 + lane < cur.je) {
            const double inv = linv;
            if (expand) {
                const double tx = lx - cx, ty = ly - cy, tz = lz - cz;
                const double ax_ = m2c * tx, ay_ = m2c * ty, az_ = m2c * tz;
                tile[q0 + lane] = P4{ax_, ay_, az_, c * __builtin_fma(tz, tz, __builtin_fma(ty, ty, tx * tx))};
                tw[q0 + lane] = P4{ax_ * inv, ay_ * inv, az_ * inv, inv};
            } else {
                tile[q0 + lane] = P4{lx, ly, lz, inv};
                tw[q0 + lane] = P4{lx * inv, ly * inv, lz * inv, inv};
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef GINGR_STAMPS
        if (first__) { STAMP(3) first__ = false; }
#endif
        const Work nxt = find();
        if (nxt.valid) issue(nxt);
        const int q1 = min((int)(cur.je - cur.jb), q0 + 64);
        const unsigned mask = cur.mask;
        if (expand) {
            if (!FINE || mask == kAllSlots)
                rowstats_tile_expand<PT, false>(tile, tw, q0, q1, mask, x, y, z, n, a1, ax, ay, az, T);
            else
                rowstats_tile_expand<PT, true>(tile, tw, q0, q1, mask, x, y, z, n, a1, ax, ay, az, T);
        } else if (clamp) {
            if (!FINE || mask == kAllSlots)
                rowstats_tile<PT, true, false>(tile, tw, q0, q1, mask, x, y, z, a1, ax, ay, az, c, lim, T);
            else
                rowstats_tile<PT, true, true>(tile, tw, q0, q1, mask, x, y, z, a1, ax, ay, az, c, lim, T);
        } else {
            if (!FINE || mask == kAllSlots)
                rowstats_tile<PT, false, false>(tile, tw, q0, q1, mask, x, y, z, a1, ax, ay, az, c, lim, T);
            else
                rowstats_tile<PT, false, true>(tile, tw, q0, q1, mask, x, y, z, a1, ax, ay, az, c, lim, T);
        }
        cur = nxt;
    }
    fastexp_round_nearest();
    STAMP(4)
    // combine the four waves in a fixed order; the LDS block is reused, so everybody must be done with T / tile / tw first
    __syncthreads();
    constexpr int kPts = 64 * PT;
    double *sred = smem;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        sred[(q * 4 + 0) * kPts + t * 64 + lane] = a1[t];
        sred[(q * 4 + 1) * kPts + t * 64 + lane] = ax[t];
        sred[(q * 4 + 2) * kPts + t * 64 + lane] = ay[t];
        sred[(q * 4 + 3) * kPts + t * 64 + lane] = az[t];
    }
    __syncthreads();
    const int64_t M = fit.n;
    double *base = partial + (int64_t)blockIdx.y * 4 * M;
    const double back = expand ? -0.5 / c : 1.0;  // sum_j p a_j -> sum_j p x~_j
    for (int p = tid; p < kPts; p += kBlock) {
        double v[4];
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
            v[pl] = (sred[(0 * 4 + pl) * kPts + p] + sred[(1 * 4 + pl) * kPts + p]) + (sred[(2 * 4 + pl) * kPts + p] + sred[(3 * 4 + pl) * kPts + p]);
        if (expand) {
            const double e = expand_owned_scale(sfrac[p]);
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) v[pl] *= e;
        }
        const int64_t i = (int64_t)blockIdx.x * kPts + p;
        if (i < M) {
            base[i] = v[0];
            base[M + i] = expand ? __builtin_fma(cx, v[0], back * v[1]) : v[1];
            base[2 * M + i] = expand ? __builtin_fma(cy, v[0], back * v[2]) : v[2];
            base[3 * M + i] = expand ? __builtin_fma(cz, v[0], back * v[3]) : v[3];
        }
    }
    STAMP_END
}

// P1 / PX from chunk partials plus per-block partials of
// Np = sum P1, trPXY = sum_i y_i . PX_i, yPy = sum_i P1_i |y_i|^2 over the local rows.
// Always launched with kScalarBlocks workgroups; part[(1..3)*kScalarBlocks + block].
// A workgroup of 1024 threads takes 256 consecutive rows at a time: thread (g, row) adds the chunks of quarter g of the chunk range,
// [g nch / 4, (g + 1) nch / 4), in ascending order -- every load instruction reads a 512-byte run of one chunk plane, the four planes
// are independent chains, so a thread keeps 4 x 4 loads in flight -- and the four quarter sums of a row are combined as
// (q0 + q1) + (q2 + q3): a fixed order.  (Round 2 used one thread per row over all chunks: the 30 chunks of a 6250-row shard were 8
// dependent batches of loads, 9.5 us of pure latency; four threads per row need two.)
// obs.weight != nullptr: the CPD observations of the rows (CPDCorrespondence.estimate + getUncertainty, CPD.scala:36-46,120-128)
// are produced in the same pass -- weight_i = P1_i / (sigma2 lambda), e_i = weight_i (R^T (yhat_i - c - t) - (ref_i - c) - mean_i)
// with yhat_i = y_i + (PX_i / P1_i - y_i); rows overridden by a landmark get weight 0 (GingrAlgorithm.scala:289-292).
// ROWS (round 4): rows per workgroup step, 256 or 64 (4 ROWS threads).  With 256 a shard of 6 250 rows keeps only 25 of the 256
// workgroups -- 25 compute units -- busy reading its 6 MB of partials (9.2 us, a sixth of the rows in three quarters of the full
// cloud's time); 64 rows per workgroup spread the same rows over 98.  Same order of additions per row either way.
template <int ROWS>
__global__ __launch_bounds__(4 * ROWS) void rowstats_reduce_kernel(const double *__restrict__ partial, int nchunks, Cloud fit,
                                                                   double *__restrict__ P1, double *__restrict__ PX,
                                                                   double *__restrict__ part, CpdObsArgs obs) {
    __shared__ double sh[ROWS];
    __shared__ double quart[3][4][ROWS];  // [quarter 1..3][plane][row]
    const int64_t M = fit.n;
    const int row = threadIdx.x % ROWS, g = threadIdx.x / ROWS;
    const int c0 = (int)((int64_t)g * nchunks / 4), c1 = (int)((int64_t)(g + 1) * nchunks / 4);
    double np = 0.0, tr = 0.0, ypy = 0.0;
    for (int64_t i0 = (int64_t)blockIdx.x * ROWS; i0 < M; i0 += (int64_t)kScalarBlocks * ROWS) {
        const int64_t i = i0 + row;
        const bool ok = i < M;
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        double yx = 0.0, yy = 0.0, yz = 0.0, rf[3] = {0.0, 0.0, 0.0}, mn[3] = {0.0, 0.0, 0.0};
        int masked = 0;
        if (ok) {
            if (g == 0) {  // what the finishing thread of the row needs besides the sums: requested together with them
                yx = fit.x[i];
                yy = fit.y[i];
                yz = fit.z[i];
                if (obs.weight) {
                    masked = obs.lm_mask ? obs.lm_mask[i] : 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        rf[d] = obs.ref[d * M + i];
                        mn[d] = obs.mean[d * M + i];
                    }
                }
            }
            const double *b = partial + i;
#pragma unroll 4
            for (int c = c0; c < c1; ++c) {
                const double *bc = b + (int64_t)c * 4 * M;
                v[0] += bc[0];
                v[1] += bc[M];
                v[2] += bc[2 * M];
                v[3] += bc[3 * M];
            }
        }
        if (g > 0) {
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) quart[g - 1][pl][row] = v[pl];
        }
        __syncthreads();
        if (g == 0 && ok) {
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) v[pl] = (v[pl] + quart[0][pl][row]) + (quart[1][pl][row] + quart[2][pl][row]);
            P1[i] = v[0];
            PX[i] = v[1];
            PX[M + i] = v[2];
            PX[2 * M + i] = v[3];
            if (obs.weight) {
                if (masked) {
                    obs.weight[i] = 0.0;
                    obs.evec[i] = obs.evec[M + i] = obs.evec[2 * M + i] = 0.0;
                } else {
                    const double p1inv = 1.0 / v[0];                                                  // CPD.scala:37
                    const double ox = yx + (v[1] * p1inv - yx), oy = yy + (v[2] * p1inv - yy), oz = yz + (v[3] * p1inv - yz);
                    const double wgt = 1.0 / (obs.sigma2[0] * obs.lambda * p1inv);                     // CPD.scala:126
                    const double *R = obs.R;
                    const double dx = ox - obs.center[0] - obs.t[0], dy = oy - obs.center[1] - obs.t[1], dz = oz - obs.center[2] - obs.t[2];
                    const double ex = R[0] * dx + R[3] * dy + R[6] * dz - (rf[0] - obs.center[0]) - mn[0];
                    const double ey = R[1] * dx + R[4] * dy + R[7] * dz - (rf[1] - obs.center[1]) - mn[1];
                    const double ez = R[2] * dx + R[5] * dy + R[8] * dz - (rf[2] - obs.center[2]) - mn[2];
                    obs.weight[i] = wgt;
                    obs.evec[i] = wgt * ex;
                    obs.evec[M + i] = wgt * ey;
                    obs.evec[2 * M + i] = wgt * ez;
                }
            }
            np += v[0];
            tr += yx * v[1] + yy * v[2] + yz * v[3];
            ypy += v[0] * (yx * yx + yy * yy + yz * yz);
        }
        __syncthreads();  // quart is rewritten by the next group of rows
    }
    // the scalar partials of the block: only the ROWS finishing threads hold values
    double tot[3] = {0.0, 0.0, 0.0};
    const double vals[3] = {np, tr, ypy};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (g == 0) sh[row] = vals[q];
        __syncthreads();
#pragma unroll
        for (int st = ROWS / 2; st > 0; st >>= 1) {
            if (g == 0 && row < st) sh[row] += sh[row + st];
            __syncthreads();
        }
        tot[q] = sh[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[kScalarBlocks + blockIdx.x] = tot[0];
        part[2 * kScalarBlocks + blockIdx.x] = tot[1];
        part[3 * kScalarBlocks + blockIdx.x] = tot[2];
    }
}
__global__ __launch_bounds__(256) void cpd_scalars_finish_kernel(const double *__restrict__ part, double *__restrict__ scalars,
                                                                 double *__restrict__ xch8, int contribute_xpx) {
    __shared__ double sh[256];
    const int map[4] = {1, 0, 2, 3};  // part slot -> scalar index
    for (int q = 0; q < 4; ++q) {
        const double tot = block_sum<256>(part[q * kScalarBlocks + threadIdx.x], sh);
        if (threadIdx.x == 0) {
            scalars[map[q]] = tot;
            // xPx is a sum over ALL targets, computed on every shard: only one of them may contribute it
            if (xch8) xch8[map[q]] = (map[q] == 1 && !contribute_xpx) ? 0.0 : tot;
        }
        __syncthreads();
    }
    if (xch8 && threadIdx.x >= 4 && threadIdx.x < 8) xch8[threadIdx.x] = 0.0;
}

// ---------------------------------------------------------------- nearest neighbour (exact, lowest index on ties)
// Distances use separately rounded multiplies and adds (no FMA contraction) so that d2 is bit-identical to the
// reference expression dx*dx + dy*dy + dz*dz evaluated in float64 on a CPU; the argmin is then index-exact.
__device__ __forceinline__ double norm2_exact(double dx, double dy, double dz) {
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// A workgroup of four waves serves 64 consecutive queries (spatially compact when the fitter keeps clouds in k-d leaf order):
// every wave holds the same queries and scans ONE 64-point quarter of each staged target tile, visited only if that quarter's box
// is not farther from a lane's query than the lane's best so far.  Tiles are found with the lanes testing 64 tile boxes at a time
// against the queries' box; the tiles at the smallest gap come first, then the four waves share their best distances (the bound
// only) and the remaining tiles are visited under that bound -- exact pruning: the result is the same as the full scan including
// the lowest-original-index tie rule (a NaN box or query never prunes).
constexpr int kNNThreads = 64;   // queries per workgroup
constexpr int kNNBlock = 256;    // threads per workgroup

// COUNT (diagnostics, gingr_ctx_nn_counting): *tests += the distance tests the launch really executed (64 lanes x the entries of every
// scanned quarter), one integer atomic per wave at its end -- the denominator of the kernel's roofline figure after pruning.
template <bool COUNT>
__global__ __launch_bounds__(kNNBlock) void nn_kernel(Cloud q, Cloud tgt, const int32_t *__restrict__ orig,
                                                      const double *__restrict__ tgt_boxes, int64_t cols_per_chunk,
                                                      double *__restrict__ pd2, int32_t *__restrict__ pidx,
                                                      int32_t *__restrict__ porig, const int32_t *warm /* may alias idx_out */,
                                                      unsigned long long *tests, const uint8_t *__restrict__ mask,
                                                      const int32_t *__restrict__ nmask, int32_t *idx_out,
                                                      double *__restrict__ d2_out) {
    // masked launch (the queries the grid search of nn_grid.hip left over): nothing to do at all, or nothing for these 64 queries
    if (nmask && *nmask == 0) return;
    if (mask) {
        const int64_t iq = (int64_t)blockIdx.x * kNNThreads + (threadIdx.x & 63);
        if (!__any(iq < q.n && mask[iq] != 0)) return;  // the four waves hold the same queries: a workgroup-uniform exit
    }
    unsigned long long scanned = 0;  // wave-uniform
    __shared__ P4 tile[kTile];
    __shared__ double sbest[4][kNNThreads], sorig[4][kNNThreads];
    __shared__ int32_t sidx[4][kNNThreads];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kNNThreads + lane;
    const bool ok = i < q.n;
    const double qx = ok ? q.x[i] : 0.0, qy = ok ? q.y[i] : 0.0, qz = ok ? q.z[i] : 0.0;
    double best = __builtin_huge_val(), bo = __builtin_huge_val();  // best distance and the ORIGINAL index that holds it
    double bound = __builtin_huge_val();                            // best of all four waves after the first sweep
    int32_t bi = -1;
    // Warm start (nullable): the position of the target this query matched LAST time (an ICP iteration moves the queries a little).
    // Its distance, computed with the arithmetic of the scan, is a valid candidate and prunes from the first tile on; the result is
    // the same exact minimum with the same tie rule (a tile holding an equally close target has a gap <= the bound and is visited).
    if (warm && ok) {
        const int32_t p = warm[i];
        if (p >= 0 && p < tgt.n) {
            const double d2 = norm2_exact(tgt.x[p] - qx, tgt.y[p] - qy, tgt.z[p] - qz);
            if (d2 == d2) {  // a NaN query keeps the cold-start behaviour
                best = d2;
                bo = (double)(orig ? orig[p] : p);
                bi = p;
            }
        }
    }
    const int64_t j0 = (int64_t)blockIdx.y * cols_per_chunk;
    const int64_t j1 = min(tgt.n, j0 + cols_per_chunk);
    const int t0 = (int)(j0 / kTile), nt = (int)((j1 - j0 + kTile - 1) / kTile);
    const double *qboxes = tgt_boxes ? tgt_boxes + ((tgt.n + kTile - 1) / kTile) * 6 : nullptr;
    // the queries' bounding box (invalid lanes excluded)
    Box wb;
    {
        double lo[3] = {ok ? qx : __builtin_huge_val(), ok ? qy : __builtin_huge_val(), ok ? qz : __builtin_huge_val()};
        double hi[3] = {ok ? qx : -__builtin_huge_val(), ok ? qy : -__builtin_huge_val(), ok ? qz : -__builtin_huge_val()};
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                lo[d] = fmin(lo[d], __shfl_xor(lo[d], off));
                hi[d] = fmax(hi[d], __shfl_xor(hi[d], off));
            }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            wb.lo[d] = uniform_d(lo[d]);
            wb.hi[d] = uniform_d(hi[d]);
        }
    }
    double gmin = __builtin_huge_val();
    if (tgt_boxes) {
        for (int t = lane; t < nt; t += 64) gmin = fmin(gmin, box_gap2(wb, tgt_boxes + (int64_t)(t0 + t) * 6));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gmin = fmin(gmin, __shfl_xor(gmin, off));
        gmin = uniform_d(gmin);
    }
    for (int phase = 0; phase < 2; ++phase) {
        if (!tgt_boxes && phase == 1) break;  // no boxes: the first sweep visits everything
        double bmax = ok ? fmin(best, bound) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bmax = fmax(bmax, __shfl_xor(bmax, off));
        bmax = uniform_d(bmax) * (1.0 + 1e-12);  // identical in the four waves in the second sweep (bound is shared)
        for (int tc = 0; tc < nt; tc += 64) {
            const int tl = tc + lane;
            bool take = tl < nt;
            if (tgt_boxes && take) {
                const double g = box_gap2(wb, tgt_boxes + (int64_t)(t0 + tl) * 6);
                // a NaN box gives g = NaN: taken in the first sweep
                take = phase == 0 ? !(g > gmin) : (g > gmin && !(g > bmax));
            }
            unsigned long long cand = __ballot(take);  // workgroup-uniform: same queries, same gmin / bound in every wave
            while (cand) {
                const int t = tc + __builtin_ctzll(cand);
                cand &= cand - 1;
                const int64_t jb = j0 + (int64_t)t * kTile, q0 = jb + 64 * wave;
                bool need = ok && q0 < j1;
                if (qboxes) {
                    const double *bx = qboxes + ((int64_t)(t0 + t) * 4 + wave) * 6;
                    const double gx = fmax(fmax(bx[0] - qx, qx - bx[3]), 0.0), gy = fmax(fmax(bx[1] - qy, qy - bx[4]), 0.0),
                                 gz = fmax(fmax(bx[2] - qz, qz - bx[5]), 0.0);
                    const double pd = __builtin_fma(gz, gz, __builtin_fma(gy, gy, gx * gx));
                    need = need && !(pd > fmin(best, bound) * (1.0 + 1e-12));
                }
                const bool wave_needs = __any(need);
                if (!__syncthreads_or(wave_needs)) continue;
                {
                    const int64_t j = jb + threadIdx.x;
                    if (j < j1) tile[threadIdx.x] = P4{tgt.x[j], tgt.y[j], tgt.z[j], (double)(orig ? orig[j] : (int32_t)j)};
                }
                __syncthreads();
                if (wave_needs) {
                    const int cnt = (int)min((int64_t)64, j1 - q0);
                    if (COUNT) scanned += (unsigned long long)cnt * 64ull;
#pragma unroll 4
                    for (int jj = 0; jj < cnt; ++jj) {
                        const P4 p = tile[64 * wave + jj];
                        const double d2 = norm2_exact(p.x - qx, p.y - qy, p.z - qz);
                        // strictly closer, or exactly as close with a lower original index: "lowest index wins" independent of
                        // the (spatially sorted) device order and of the visiting order
                        if (d2 < best || (d2 == best && p.w < bo)) {
                            best = d2;
                            bo = p.w;
                            bi = (int32_t)(q0 + jj);
                        }
                    }
                }
                __syncthreads();  // the tile is restaged by the next visited tile
            }
        }
        if (phase == 0 && tgt_boxes) {  // share the distance bound of the first sweep
            sbest[wave][lane] = best;
            __syncthreads();
            bound = fmin(fmin(sbest[0][lane], sbest[1][lane]), fmin(sbest[2][lane], sbest[3][lane]));
            __syncthreads();
        }
    }
    // combine the four waves: smallest distance, ties -> lowest original index
    sbest[wave][lane] = best;
    sorig[wave][lane] = bo;
    sidx[wave][lane] = bi;
    __syncthreads();
    if (wave == 0 && ok) {
        int w = 0;
        for (int k = 1; k < 4; ++k)
            if (sbest[k][lane] < sbest[w][lane] || (sbest[k][lane] == sbest[w][lane] && sorig[k][lane] < sorig[w][lane])) w = k;
        const double wo = sorig[w][lane];
        if (idx_out) {  // masked launch over ONE chunk: this is the answer (no reduction kernel follows); unflagged queries stay as they are
            if (mask[i]) idx_out[i] = sidx[w][lane], d2_out[i] = sbest[w][lane];
        } else {
            pd2[(int64_t)blockIdx.y * q.n + i] = sbest[w][lane];
            pidx[(int64_t)blockIdx.y * q.n + i] = sidx[w][lane];
            porig[(int64_t)blockIdx.y * q.n + i] = (int32_t)(wo < 2147483648.0 ? wo : -1.0);
        }
    }
    if (COUNT && lane == 0 && scanned) atomicAdd(tests, scanned);
}

__global__ void nn_reduce_kernel(const double *__restrict__ pd2, const int32_t *__restrict__ pidx,
                                 const int32_t *__restrict__ porig, int nchunks, int64_t M, int32_t *__restrict__ idx,
                                 double *__restrict__ d2, const uint8_t *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (mask && !mask[i]) return;  // answered by the grid search; the scan's partials may not even exist
    double best = pd2[i];
    int32_t bi = pidx[i], bo = porig[i];
    for (int c = 1; c < nchunks; ++c) {
        const double v = pd2[(int64_t)c * M + i];
        const int32_t o = porig[(int64_t)c * M + i];
        if (v < best || (v == best && o >= 0 && (bo < 0 || o < bo))) {
            best = v;
            bi = pidx[(int64_t)c * M + i];
            bo = o;
        }
    }
    idx[i] = bi;
    d2[i] = best;
}

// ---- small problems (the stateless gingr_nn of a few thousand points each; BASELINE config 2): all pairs, no boxes, no tiles.
// Pass 1: a workgroup = 512 queries (two per lane) x one slice of the targets in the CALLER's order; the whole slice goes to LDS with
// every load in flight at once, then every lane walks it with wave-uniform (broadcast) LDS reads.  Per pair the separately rounded
// expression of norm2_exact (3 subtractions, 3 products, 2 sums) and ONE minimum -- 9 instructions; which target it was is not
// tracked (a compare, a select and a move per pair for an answer 1 / nslices of the slices contribute to).
// Pass 2: sixteen lanes per query pick the first slice that holds the overall minimum (ascending slices, strict <) and search THAT
// slice again for the first target at exactly this distance (same expression, same bits): the lowest index on ties, as the full scan.
// A NaN distance never wins (v_min returns the other operand); a query with no finite distance keeps index -1, distance +inf.
__global__ __launch_bounds__(256) void nn_small_kernel(Cloud q, const double *__restrict__ tx, const double *__restrict__ ty,
                                                       const double *__restrict__ tz, int32_t n_targets, int32_t slice_len,
                                                       double *__restrict__ pd2) {
    extern __shared__ double sh[];  // [3][slice_len]
    const int tid = threadIdx.x;
    const int32_t j0 = (int32_t)blockIdx.y * slice_len, n = min(slice_len, n_targets - j0);
    double *sx = sh, *sy = sh + slice_len, *sz = sh + 2 * slice_len;
    for (int32_t k = tid; k < n; k += 256) sx[k] = tx[j0 + k], sy[k] = ty[j0 + k], sz[k] = tz[j0 + k];
    const int64_t ia = (int64_t)blockIdx.x * 512 + tid, ib = ia + 256;
    const bool oka = ia < q.n, okb = ib < q.n;
    const double ax = oka ? q.x[ia] : 0.0, ay = oka ? q.y[ia] : 0.0, az = oka ? q.z[ia] : 0.0;
    const double bx = okb ? q.x[ib] : 0.0, by = okb ? q.y[ib] : 0.0, bz = okb ? q.z[ib] : 0.0;
    __syncthreads();
    double besta = __builtin_huge_val(), bestb = __builtin_huge_val();
    int32_t j = 0;
    for (; j + 4 <= n; j += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = sx[j + k], y = sy[j + k], z = sz[j + k];
            besta = fmin(besta, norm2_exact(x - ax, y - ay, z - az));
            bestb = fmin(bestb, norm2_exact(x - bx, y - by, z - bz));
        }
    }
    for (; j < n; ++j) {
        const double x = sx[j], y = sy[j], z = sz[j];
        besta = fmin(besta, norm2_exact(x - ax, y - ay, z - az));
        bestb = fmin(bestb, norm2_exact(x - bx, y - by, z - bz));
    }
    if (oka) pd2[(int64_t)blockIdx.y * q.n + ia] = besta;
    if (okb) pd2[(int64_t)blockIdx.y * q.n + ib] = bestb;
}

__global__ void nn_small_count_kernel(unsigned long long *tests, unsigned long long n) { *tests += n; }

__global__ __launch_bounds__(256) void nn_small_reduce_kernel(Cloud q, const double *__restrict__ tx, const double *__restrict__ ty,
                                                              const double *__restrict__ tz, int32_t n_targets, int32_t slice_len,
                                                              const double *__restrict__ pd2, int nslices, int32_t *__restrict__ idx,
                                                              double *__restrict__ d2) {
    const int l = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4), M = q.n;
    const bool ok = i < M;
    double best = __builtin_huge_val();
    int32_t bs = INT32_MAX;  // the first slice that holds `best`
    if (ok)
        for (int c = l; c < nslices; c += 16) {
            const double v = pd2[(int64_t)c * M + i];
            if (v < best) best = v, bs = c;
        }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const double v = __shfl_xor(best, off);
        const int32_t o = __shfl_xor(bs, off);
        if (v < best || (v == best && o < bs)) best = v, bs = o;
    }
    int32_t bi = INT32_MAX;
    if (ok && bs != INT32_MAX) {  // (best < +inf) the first target of slice bs at exactly this distance
        const double qx = q.x[i], qy = q.y[i], qz = q.z[i];
        const int32_t j0 = bs * slice_len, j1 = min(j0 + slice_len, n_targets);
        for (int32_t j = j0 + l; j < j1; j += 16)
            if (norm2_exact(tx[j] - qx, ty[j] - qy, tz[j] - qz) == best) {
                bi = j;
                break;
            }
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) bi = min(bi, __shfl_xor(bi, off));
    if (ok && l == 0) {
        idx[i] = bi == INT32_MAX ? -1 : bi;
        d2[i] = best;
    }
}

// ---------------------------------------------------------------- Gaussian kernel block
__global__ __launch_bounds__(kBlock) void gauss_block_kernel(Cloud A, Cloud B, double sigma, double scaling,
                                                             double *__restrict__ out) {
    __shared__ double T[GINGR_EXP_TABLE];
    fastexp_table_init(T);
    __syncthreads();
    const double c = fastexp_scale_for_variance(sigma * sigma);
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= B.n) return;
    const double dx = A.x[i] - B.x[j], dy = A.y[i] - B.y[j], dz = A.z[i] - B.z[j];
    const double d2 = fmin(__builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)), fastexp_d2_limit(c));
    out[i * B.n + j] = scaling * fastexp2_scaled(d2, c, T);
}

// ---------------------------------------------------------------- sum of squared pair distances (initial sigma2)
__global__ __launch_bounds__(kBlock) void sumsq_pairs_kernel(Cloud A, Cloud B, double *__restrict__ partial) {
    __shared__ P4 tile[kTile];
    __shared__ double sh[kBlock];
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
    const bool ok = i < A.n;
    const double ax = ok ? A.x[i] : 0.0, ay = ok ? A.y[i] : 0.0, az = ok ? A.z[i] : 0.0;
    double acc = 0.0;
    for (int64_t jb = 0; jb < B.n; jb += kTile) {
        __syncthreads();
        const int64_t j = jb + tid;
        if (j < B.n) tile[tid] = P4{B.x[j], B.y[j], B.z[j], 0.0};
        __syncthreads();
        const int cnt = (int)min((int64_t)kTile, B.n - jb);
        for (int jj = 0; jj < cnt; ++jj) {
            const P4 p = tile[jj];
            const double dx = p.x - ax, dy = p.y - ay, dz = p.z - az;
            acc += dx * dx + dy * dy + dz * dz;
        }
    }
    if (!ok) acc = 0.0;
    __syncthreads();
    const double tot = block_sum<kBlock>(acc, sh);
    if (tid == 0) partial[blockIdx.x] = tot;
}

// The same sum from moments: sum_ij |a_i - b_j|^2 = n_B sum |a_i - c|^2 + n_A sum |b_j - c|^2 - 2 (sum (a_i - c)) . (sum (b_j - c)) for any c
// (here a_0, so that the three terms are of the size of the result: no cancellation beyond a digit) -- O(n_A + n_B) instead of the
// pair loop's 1.3 ms at 50k x 50k; it differs from the reference's double loop (CPD.scala:81-90) by rounding only, as the pair
// loop's tree of partial sums did.  partial: [2][kMomentBlocks][4].
constexpr int kMomentBlocks = 64;
__global__ __launch_bounds__(kBlock) void cloud_moments_kernel(Cloud A, Cloud B, double *__restrict__ partial) {
    __shared__ double sh[kBlock];
    const Cloud C = blockIdx.y == 0 ? A : B;
    const double cx = A.x[0], cy = A.y[0], cz = A.z[0];
    double sx = 0.0, sy = 0.0, sz = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < C.n; i += (int64_t)kMomentBlocks * kBlock) {
        const double dx = C.x[i] - cx, dy = C.y[i] - cy, dz = C.z[i] - cz;
        sx += dx, sy += dy, sz += dz;
        s2 += dx * dx + dy * dy + dz * dz;
    }
    double *out = partial + ((int64_t)blockIdx.y * kMomentBlocks + blockIdx.x) * 4;
    const double v[4] = {sx, sy, sz, s2};
    for (int k = 0; k < 4; ++k) {
        __syncthreads();
        const double tot = block_sum<kBlock>(v[k], sh);
        if (threadIdx.x == 0) out[k] = tot;
    }
}
__global__ void sumsq_from_moments_kernel(const double *__restrict__ partial, int64_t nA, int64_t nB, double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double m[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < kMomentBlocks; ++b)
            for (int k = 0; k < 4; ++k) m[c][k] += partial[((int64_t)c * kMomentBlocks + b) * 4 + k];
    out[0] = ((double)nB * m[0][3] + (double)nA * m[1][3]) - 2.0 * (m[0][0] * m[1][0] + m[0][1] * m[1][1] + m[0][2] * m[1][2]);
}

__global__ __launch_bounds__(1024) void sum_vector_kernel(const double *__restrict__ v, int64_t n, double scale,
                                                          double *__restrict__ out) {
    __shared__ double sh[1024];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += v[i];
    const double tot = block_sum<1024>(acc, sh);
    if (threadIdx.x == 0) out[0] = tot * scale;
}

__global__ void aos_to_soa_kernel(const double *__restrict__ aos, int64_t n, const int32_t *__restrict__ perm,
                                  double *__restrict__ soa) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t o = perm ? perm[i] : i;
    soa[i] = aos[3 * o];
    soa[n + i] = aos[3 * o + 1];
    soa[2 * n + i] = aos[3 * o + 2];
}

__global__ void soa_to_aos_kernel(const double *__restrict__ soa, int64_t n, const int32_t *__restrict__ perm,
                                  double *__restrict__ aos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t o = perm ? perm[i] : i;
    aos[3 * o] = soa[i];
    aos[3 * o + 1] = soa[n + i];
    aos[3 * o + 2] = soa[2 * n + i];
}

__global__ void scatter_kernel(const double *__restrict__ in, int64_t n, const int32_t *__restrict__ perm,
                               double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[perm ? perm[i] : i] = in[i];
}

constexpr int kPT = GINGR_PT_DEFAULT;     // points per thread in both CPD passes
// Row statistics of a TINY shard: two points per thread halve the workgroup's rows (more workgroups along the row axis, a fourth
// workgroup per CU).  With round 2's work split (a workgroup owns 256 rows whatever PT is) four points per thread win from a few
// thousand rows on: 8-GPU shard of the 50k workload (6250 rows) 0.50 ms per iteration with PT = 2, 0.49 ms with PT = 4.
#ifndef GINGR_SMALL_COLSUM_COLS
#define GINGR_SMALL_COLSUM_COLS 16384
#endif
#ifndef GINGR_SMALL_SHARD_ROWS
#define GINGR_SMALL_SHARD_ROWS 2048
#endif
constexpr int64_t kSmallShardRows = GINGR_SMALL_SHARD_ROWS;
constexpr int64_t kSmallColsumCols = GINGR_SMALL_COLSUM_COLS;
constexpr int64_t kTinyCols = 2048;
// Build-time knobs of the chunk planner (the sweeps behind the defaults: tools/chunk_sweep.sh, tools/small_chunk_sweep.sh build the
// library with -DGINGR_...=v through tools/abn.sh; none of them is read from the environment):
//   GINGR_ROWSTATS_PT        2 or 4 points per thread in the row-statistics pass whatever the shard size (0: by shard size)
//   GINGR_COLSUM_QUARTERS / GINGR_ROWSTATS_QUARTERS   fixed chunk length in 64-point quarters (0: planner)
//   GINGR_COLSUM_CHUNKS / GINGR_ROWSTATS_CHUNKS       exactly n chunks balanced to a quarter (0: planner)
//   GINGR_FAIR_PRIORITY      0 never, 1 in one-round launches (default), 2 always: waves lower their issue priority as they advance
#ifndef GINGR_ROWSTATS_PT
#define GINGR_ROWSTATS_PT 0
#endif
#ifndef GINGR_COLSUM_QUARTERS
#define GINGR_COLSUM_QUARTERS 0
#endif
#ifndef GINGR_ROWSTATS_QUARTERS
#define GINGR_ROWSTATS_QUARTERS 0
#endif
#ifndef GINGR_COLSUM_CHUNKS
#define GINGR_COLSUM_CHUNKS 0
#endif
#ifndef GINGR_ROWSTATS_CHUNKS
#define GINGR_ROWSTATS_CHUNKS 0
#endif
#ifndef GINGR_FAIR_PRIORITY
#define GINGR_FAIR_PRIORITY 1
#endif
#ifndef GINGR_COLSUM_PT
#define GINGR_COLSUM_PT 0
#endif
// owned points (targets) per thread of the column-sum pass: small target clouds take 2, tiny ones 1 -- two / four times the
// workgroups of a launch that fills a fifth of the chip at femur size (GINGR_COLSUM_PT: build-time override for the sweep)
inline int colsum_pt(int64_t cols) {
    constexpr int forced = GINGR_COLSUM_PT;
    if (forced == 1 || forced == 2 || forced == kPT) return forced;
    if (kPT > 2 && cols <= kTinyCols) return 1;  // femur size: 0.0973 -> 0.0950 ms per iteration against 2
    return (kPT > 2 && cols <= kSmallColsumCols) ? 2 : kPT;
}
// owned points (rows) per thread of the row-statistics pass: 1 where both clouds are tiny (femur), 2 on short shards (few row blocks whatever the target count) and on
// problems that are small on BOTH sides (15k x 15k: 0.354 -> 0.348 ms; a 6 250-row shard of 50k targets is slower with 2: 0.421 -> 0.435)
inline int rowstats_pt(int64_t rows, int64_t cols) {
    constexpr int forced = GINGR_ROWSTATS_PT;
    if (forced == 1 || forced == 2 || forced == kPT) return forced;
    if (kPT > 2 && rows <= kTinyCols && cols <= kTinyCols) return 1;  // femur size: 0.0950 -> 0.0932 ms per iteration against 2
    return (kPT > 2 && (rows <= kSmallShardRows || (rows <= kSmallColsumCols && cols <= kSmallColsumCols))) ? 2 : kPT;
}
// Workgroups per all-pairs launch.  A CU holds 3-4 of them and one lives for (tiles per chunk) x ~30 us, so the launch ends with
// a tail of about one workgroup's life: many short workgroups beat few long ones until the per-chunk partials (written here,
// read by the reduce kernels) cost more than the tail.  Measured with GINGR_COLSUM_TILES / GINGR_ROWSTATS_TILES at 50k <-> 50k
// on shards of 1/1, 1/2, 1/4, 1/8 of the rows (profiles/r01_chunk_length_sweep.txt): two tiles per chunk (~4900 workgroups)
// is 5 % faster than five (~2000) on the whole cloud, one tile is best on the shards.
constexpr int kTargetBlocks = 5120;
#ifndef GINGR_MIN_CHUNK
#define GINGR_MIN_CHUNK 256
#endif
constexpr int kMinChunk = GINGR_MIN_CHUNK;  // shortest chunk the planner picks by itself (64, 128 or 256 points)

// split `stream_len` into chunks so that block_cols * nchunks ~ kTargetBlocks.  A chunk is a whole number of tiles, or -- when
// even one tile per chunk leaves too few workgroups (small shards) -- a half or a quarter of a tile: 64-point quarters are the
// unit of the culling boxes, and a chunk that divides a tile never straddles two tiles' boxes.
// `quarters_override` > 0 (build-time knob GINGR_COLSUM_QUARTERS / GINGR_ROWSTATS_QUARTERS) fixes the chunk length.
// Resident workgroups of the chip for one of the all-pairs kernels (compute units x workgroups per unit), queried once.  Launches
// come in rounds of that many workgroups and the last, partly filled round costs almost a full one (measured at 50k x 50k:
// 4.79 -> 4.98 rounds of the column-sum pass is 2.7 % FASTER, 6.38 -> 5.87 rounds of the row-statistics pass 3 %), so the chunk
// planner below makes the number of workgroups come out just under a whole number of rounds.
inline int resident_workgroups(int which /* 0 column sums, 1 row statistics PT = 2, 2 row statistics PT = kPT */) {
    static int cache[3] = {0, 0, 0};
    if (cache[which] > 0) return cache[which];
    int per_cu = 0, dev = 0;
    hipError_t e;
    if (which == 0)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cpd_colsum_kernel<GINGR_PT_DEFAULT, false>, kBlock, 0);
    else if (which == 1)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cpd_rowstats_kernel<2, false>, kBlock, 0);
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cpd_rowstats_kernel<GINGR_PT_DEFAULT, false>, kBlock, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = which == 2 ? 3 : 4;
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    (void)hipGetLastError();
    cache[which] = per_cu * cus;
    return cache[which];
}

// The quarter-tile x slot culling variant (FINE) pays for its longer prologue and per-part slot tests only when the streamed cloud
// is long: measured in the late regime (tools/fine_cull_sweep.sh, sigma2 = 4) it wins 3 % at 50k streamed points, 1 % at 30k and
// LOSES 4 % at 15k, 8 % at 1.6k (femur) -- bit-identical results either way.
constexpr int64_t kFineMinStream = 24576;

inline ChunkPlan plan_chunks(int64_t owned, int owned_per_block, int64_t stream_len, int *nchunks, int quarters_override = 0,
                             int forced_chunks = 0, int resident = 0) {
    const int64_t bx = ceil_div(owned, owned_per_block);
    int64_t want = ceil_div(kTargetBlocks, bx > 0 ? bx : 1);
    if (want < 1) want = 1;
    const int64_t n = stream_len > 0 ? stream_len : 1;
    int64_t len = round_up(ceil_div(n, want), 64);
    if (quarters_override > 0) len = (int64_t)quarters_override * 64;
    if (len >= kTile)
        len = round_up(len, kTile);
    else if (len > 128)
        len = kTile;
    else if (len > 64)
        len = 128;
    else
        len = 64;
    if (quarters_override <= 0 && len < kMinChunk) len = kMinChunk;
    ChunkPlan p{len, len, (int32_t)ceil_div(n, len), 0};
    // build-time knobs GINGR_COLSUM_CHUNKS / GINGR_ROWSTATS_CHUNKS=<n>: exactly n chunks balanced to a 64-point quarter (lengths
    // differ by at most 64; the kernels handle chunks that start inside a tile).  Default (no knob, `resident` known): the
    // largest chunk count whose workgroups fill k whole launch rounds, k = kTargetBlocks / resident rounded -- see
    // resident_workgroups -- with the same balanced lengths.
    int forced = forced_chunks;
    if (forced <= 0 && quarters_override <= 0 && resident > 0) {
        // rounds: as many as kTargetBlocks asks for (finer balancing when culling makes workgroup costs uneven), but not so many that
        // a chunk drops below ~1024 streamed points -- every workgroup pays ~3-4 us of prologue (table fill, owned points, boxes),
        // which short chunks do not amortise (8-GPU shard of the 50k workload: 0.52 -> 0.49 ms per iteration with one round)
        int64_t k = (kTargetBlocks + resident / 2) / resident;
        const int64_t kmax = (n / 1024) * (bx > 0 ? bx : 1) / resident;
        if (k > kmax) k = kmax;
        if (k < 1) k = 1;
        int64_t nc = k * resident / (bx > 0 ? bx : 1);
        const int64_t Qmax = ceil_div(n, 64);
        if (nc > Qmax) nc = Qmax;
        if (nc < 1) nc = 1;
        // Short chunks stay whole tiles: wave q works on quarter q of every tile part, so a chunk of 3 quarters leaves one wave
        // idle; only from ~16 quarters per chunk on is the unevenness (one quarter per wave at most) small against the round gain
        // (emulated 8-GPU shard: 0.517 ms with tile-aligned chunks, 0.530 ms with balanced 192/256-point chunks).
        if (Qmax / nc >= 16) forced = (int)nc;
    }
    if (forced > 0) {
        const int64_t Q = ceil_div(n, 64), q = Q / forced, rem = Q % forced;
        if (q >= 1) {
            p.len_big = (q + 1) * 64;
            p.len_tail = q * 64;
            p.n_big = (int32_t)rem;
            if (rem == 0) {
                p.len_big = p.len_tail;
                p.n_big = forced;
            }
        }
    }
    *nchunks = p.chunks(n);
    constexpr int fair_env = GINGR_FAIR_PRIORITY;
    p.fair = (fair_env == 2 || (fair_env == 1 && resident > 0 && (int64_t)*nchunks * bx <= resident)) ? 1 : 0;
    return p;
}
inline int colsum_tiles_override() { return GINGR_COLSUM_QUARTERS; }
inline int colsum_chunks_override() { return GINGR_COLSUM_CHUNKS; }
inline int rowstats_chunks_override() { return GINGR_ROWSTATS_CHUNKS; }
inline int rowstats_tiles_override() { return GINGR_ROWSTATS_QUARTERS; }

}  // namespace

int64_t cpd_colsum_ws_doubles(int64_t M, int64_t N) {
    int nch;
    plan_chunks(N, 64 * colsum_pt(N), M, &nch, colsum_tiles_override(), colsum_chunks_override(), resident_workgroups(0));
    return (int64_t)nch * N;
}
int cpd_colsum_chunks(int64_t M, int64_t N) {
    int nch;
    plan_chunks(N, 64 * colsum_pt(N), M, &nch, colsum_tiles_override(), colsum_chunks_override(), resident_workgroups(0));
    return nch;
}

int64_t cpd_rowstats_ws_doubles(int64_t M, int64_t N) {
    int nch;
    plan_chunks(M, 64 * rowstats_pt(M, N), N, &nch, rowstats_tiles_override(), rowstats_chunks_override(),
                resident_workgroups(rowstats_pt(M, N) == 2 ? 1 : 2));
    return (int64_t)nch * 4 * M;
}

static void plan_nn(int64_t nq, int64_t nt_points, bool pruned, int *nchunks, int64_t *chunk_len);

int64_t nn_ws_bytes(int64_t M, int64_t N) {
    int nch;
    int64_t len;
    plan_nn(M, N, false, &nch, &len);  // the unpruned plan has the most chunks
    return (int64_t)nch * M * (sizeof(double) + 2 * sizeof(int32_t));
}

void launch_cloud_centroid(gingr_ctx *ctx, Cloud c, double *out3) {
    hipLaunchKernelGGL(cloud_centroid_kernel, dim3(1), dim3(1024), 0, ctx->stream, c, out3);
}

void launch_cloud_absmax(gingr_ctx *ctx, Cloud c, const double *ctr, double *slot) {
    (void)hipMemsetAsync(slot, 0, sizeof(double), ctx->stream);
    const int nb = (int)(ceil_div(c.n, 256) < 64 ? ceil_div(c.n, 256) : 64);
    hipLaunchKernelGGL(cloud_absmax_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, ctx->stream, c, ctr, slot);
}

void launch_tile_bbox(gingr_ctx *ctx, Cloud c, double *boxes, const double *ctr, double *absmax_slot) {
    if (c.n <= 0) return;
    const int64_t ntiles = ceil_div(c.n, kTile);
    hipLaunchKernelGGL(tile_bbox_kernel, dim3((unsigned)ceil_div(ntiles, kBoxTilesPerBlock)), dim3(256), 0, ctx->stream, c, boxes, ctr,
                       absmax_slot, ntiles);
}

int launch_cpd_colsum(gingr_ctx *ctx, Cloud fit, Cloud target, const double *sigma2_dev, const double *aux,
                      const double *fit_boxes, double *ws, double *den_partial, int forced_chunks) {
    int nch;
    {
        TimerScope ts(ctx, 0);
        {
            const int pt = colsum_pt(target.n);
            const ChunkPlan len = plan_chunks(target.n, 64 * pt, fit.n, &nch, colsum_tiles_override(),
                                              forced_chunks > 0 ? forced_chunks : colsum_chunks_override(), resident_workgroups(0));
            dim3 grid((unsigned)ceil_div(target.n, 64 * pt), (unsigned)nch);
            const double *boxes = ctx->cull ? fit_boxes : (const double *)nullptr;
            // one variant, picked from the regime the device last reported (stale at worst: the results are the same)
            const bool fine = boxes && (ctx->fine_override >= 0 ? ctx->fine_override != 0
                                                                : (fit.n >= kFineMinStream && ctx->regime_host &&
                                                                   *(volatile int32_t *)ctx->regime_host != 0));
            if (pt == 1)
                hipLaunchKernelGGL((cpd_colsum_kernel<1, false>), grid, dim3(kBlock), 0, ctx->stream, fit, target, sigma2_dev, aux, boxes, len,
                                   ws, boxes ? ctx->regime_dev : (int32_t *)nullptr);
            else if (pt == 2)  // (small clouds never take the quarter-tile culling variant: kFineMinStream)
                hipLaunchKernelGGL((cpd_colsum_kernel<2, false>), grid, dim3(kBlock), 0, ctx->stream, fit, target, sigma2_dev, aux, boxes, len,
                                   ws, boxes ? ctx->regime_dev : (int32_t *)nullptr);
            else if (fine)
                hipLaunchKernelGGL((cpd_colsum_kernel<kPT, true>), grid, dim3(kBlock), 0, ctx->stream, fit, target, sigma2_dev, aux,
                                   boxes, len, ws, ctx->regime_dev);
            else
                hipLaunchKernelGGL((cpd_colsum_kernel<kPT, false>), grid, dim3(kBlock), 0, ctx->stream, fit, target, sigma2_dev, aux,
                                   boxes, len, ws, boxes ? ctx->regime_dev : (int32_t *)nullptr);
        }
    }
    if (den_partial)  // nullptr: the caller adds the chunk partials up itself (launch_cpd_den_finalize with the returned count)
        hipLaunchKernelGGL(chunk_reduce_kernel, dim3((unsigned)ceil_div(target.n, 256)), dim3(256), 0, ctx->stream, ws, nch,
                           target.n, den_partial);
    return nch;
}

void launch_cpd_den_finalize(gingr_ctx *ctx, Cloud target, const double *sigma2_dev, double w, int64_t M_total,
                             double *den, double *inv_den, double *Pt1, int32_t *tile_bad, double *part,
                             double *scalars_dev, const double *chunk_partial, int nchunks) {
    hipLaunchKernelGGL(cpd_den_finalize_kernel, dim3(kScalarBlocks), dim3(256), 0, ctx->stream, target, sigma2_dev, w,
                       (double)M_total / (double)target.n, den, inv_den, Pt1, tile_bad, part, scalars_dev, chunk_partial, nchunks);
}

void launch_cpd_rowstats(gingr_ctx *ctx, Cloud fit, Cloud target, const double *sigma2_dev, const double *aux,
                         const double *inv_den, const double *tgt_boxes, const int32_t *tile_bad, double *ws, double *P1,
                         double *PX_soa, double *part, double *scalars_dev, double *xch8, int contribute_xpx,
                         const CpdObsArgs *obs, bool finish_scalars) {
    int nch;
    {
        TimerScope ts(ctx, 1);
        {
            const int pt = rowstats_pt(fit.n, target.n);
            const ChunkPlan len = plan_chunks(fit.n, 64 * pt, target.n, &nch, rowstats_tiles_override(), rowstats_chunks_override(),
                                              resident_workgroups(pt == 2 ? 1 : 2));
            dim3 grid((unsigned)ceil_div(fit.n, 64 * pt), (unsigned)nch);
            const bool cull = ctx->cull && tgt_boxes && tile_bad;
            const double *boxes = cull ? tgt_boxes : (const double *)nullptr;
            const bool fine = boxes && (ctx->fine_override >= 0 ? ctx->fine_override != 0
                                                                : (target.n >= kFineMinStream && ctx->regime_host &&
                                                                   *(volatile int32_t *)ctx->regime_host != 0));
            int32_t *regime_out = boxes ? ctx->regime_dev : (int32_t *)nullptr;
            auto launch = [&](auto kern) {
                hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, ctx->stream, fit, target, sigma2_dev, aux, inv_den, boxes, tile_bad,
                                   len, ws, regime_out);
            };
            // one variant, picked from the regime the device last reported (stale at worst: the results are the same)
            if (pt == 1) {
                launch(cpd_rowstats_kernel<1, false>);
            } else if (pt == 2) {
                if (fine)
                    launch(cpd_rowstats_kernel<2, true>);
                else
                    launch(cpd_rowstats_kernel<2, false>);
            } else {
                if (fine)
                    launch(cpd_rowstats_kernel<kPT, true>);
                else
                    launch(cpd_rowstats_kernel<kPT, false>);
            }
        }
    }
    CpdObsArgs none;
    memset(&none, 0, sizeof(none));
    if (fit.n <= (int64_t)kScalarBlocks * 64)  // up to 16 384 rows: a quarter of the rows per workgroup, four times the workgroups
        hipLaunchKernelGGL(rowstats_reduce_kernel<64>, dim3(kScalarBlocks), dim3(256), 0, ctx->stream, ws, nch, fit, P1, PX_soa, part,
                           obs ? *obs : none);
    else
        hipLaunchKernelGGL(rowstats_reduce_kernel<256>, dim3(kScalarBlocks), dim3(1024), 0, ctx->stream, ws, nch, fit, P1, PX_soa, part,
                           obs ? *obs : none);
    if (finish_scalars)  // otherwise the caller's phase-1 finalize kernel sums the block partials (cpd_scalar_partials_layout)
        hipLaunchKernelGGL(cpd_scalars_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, part, scalars_dev, xch8, contribute_xpx);
}

static void plan_nn(int64_t nq, int64_t nt_points, bool pruned, int *nchunks, int64_t *chunk_len) {
    const int64_t bx = ceil_div(nq, kNNThreads);
    // Splitting the targets into chunks weakens the pruning (a chunk far from the queries has no near tile to shrink the
    // bound), so with pruning chunks are only used when there are too few query waves to occupy the chip; the full scan
    // wants ~8 waves per CU.
    int64_t want = pruned ? (bx >= 256 ? 1 : ceil_div(512, bx > 0 ? bx : 1)) : ceil_div(2048, bx > 0 ? bx : 1);
    const int64_t max_chunks = ceil_div(nt_points, kTile);
    if (want > max_chunks) want = max_chunks;
    if (want < 1) want = 1;
    int64_t len = round_up(ceil_div(nt_points, want), kTile);
    if (len < kTile) len = kTile;
    *chunk_len = len;
    *nchunks = (int)ceil_div(nt_points > 0 ? nt_points : 1, len);
}

void launch_nn(gingr_ctx *ctx, Cloud query, Cloud target, const int32_t *target_orig, const double *tgt_boxes, void *ws,
               int32_t *idx, double *d2, const int32_t *warm, const uint8_t *mask, const int32_t *nmask) {
    int nch;
    int64_t len;
    const bool pruned = ctx->cull && tgt_boxes != nullptr;
    plan_nn(query.n, target.n, pruned, &nch, &len);
    // masked (the leftovers of the grid search, normally none): one chunk from 4096 queries on, and then the scan kernel writes the
    // answers itself -- one launch that exits at once instead of two
    if (mask && pruned && query.n >= 4096) nch = 1, len = round_up(target.n, kTile);
    const bool direct = mask && nch == 1;
    double *pd2 = reinterpret_cast<double *>(ws);
    int32_t *pidx = reinterpret_cast<int32_t *>(pd2 + (int64_t)nch * query.n);
    int32_t *porig = pidx + (int64_t)nch * query.n;
    dim3 grid((unsigned)ceil_div(query.n, kNNThreads), (unsigned)nch);
    {
        TimerScope ts(ctx, 8);
        if (ctx->nn_tests)
            hipLaunchKernelGGL(nn_kernel<true>, grid, dim3(kNNBlock), 0, ctx->stream, query, target, target_orig,
                               pruned ? tgt_boxes : (const double *)nullptr, len, pd2, pidx, porig, warm, ctx->nn_tests, mask, nmask,
                               direct ? idx : (int32_t *)nullptr, direct ? d2 : (double *)nullptr);
        else
            hipLaunchKernelGGL(nn_kernel<false>, grid, dim3(kNNBlock), 0, ctx->stream, query, target, target_orig,
                               pruned ? tgt_boxes : (const double *)nullptr, len, pd2, pidx, porig, warm, (unsigned long long *)nullptr,
                               mask, nmask, direct ? idx : (int32_t *)nullptr, direct ? d2 : (double *)nullptr);
    }
    if (direct) return;
    hipLaunchKernelGGL(nn_reduce_kernel, dim3((unsigned)ceil_div(query.n, 256)), dim3(256), 0, ctx->stream, pd2, pidx, porig,
                       nch, query.n, idx, d2, mask);
}

// all pairs of two small clouds in the caller's order (see nn_small_kernel); ws: nn_small_ws_bytes(M, N)
static int nn_small_slices(int64_t M, int64_t N) {
    const int64_t groups = ceil_div(M, 512);
#ifndef GINGR_NN_SMALL_WGS
#define GINGR_NN_SMALL_WGS 768
#endif
    int64_t s = GINGR_NN_SMALL_WGS / (groups > 0 ? groups : 1);  // at most three workgroups (12 waves) per compute unit: no fourth round
    const int64_t max_s = ceil_div(N, 32);               // at least 32 targets per slice
    if (s > max_s) s = max_s;
    const int64_t min_s = ceil_div(N, 2048);             // at most 2 048 targets (48 KB of LDS) per slice
    if (s < min_s) s = min_s;
    return (int)(s < 1 ? 1 : s);
}
bool nn_small_applies(int64_t M, int64_t N) { return M >= 1 && N >= 1 && N <= INT32_MAX / 2 && M * N <= (int64_t)1 << 26; }
int64_t nn_small_ws_bytes(int64_t M, int64_t N) { return (int64_t)nn_small_slices(M, N) * M * sizeof(double); }
void launch_nn_small(gingr_ctx *ctx, Cloud query, Cloud target, void *ws, int32_t *idx, double *d2) {
    const int ns = nn_small_slices(query.n, target.n);
    const int32_t len = (int32_t)ceil_div(target.n, ns);
    const int nslices = (int)ceil_div(target.n, len);
    double *pd2 = reinterpret_cast<double *>(ws);
    TimerScope ts(ctx, 8);  // (both launches: the slices mean nothing before they are combined)
    hipLaunchKernelGGL(nn_small_kernel, dim3((unsigned)ceil_div(query.n, 512), (unsigned)nslices), dim3(256), (size_t)3 * len * sizeof(double),
                       ctx->stream, query, target.x, target.y, target.z, (int32_t)target.n, len, pd2);
    hipLaunchKernelGGL(nn_small_reduce_kernel, dim3((unsigned)ceil_div(query.n, 16)), dim3(256), 0, ctx->stream, query, target.x, target.y,
                       target.z, (int32_t)target.n, len, pd2, nslices, idx, d2);
    if (ctx->nn_tests)  // diagnostics (gingr_ctx_nn_counting): every lane of every wave tests every target
        hipLaunchKernelGGL(nn_small_count_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->nn_tests,
                           (unsigned long long)round_up(query.n, 64) * (unsigned long long)target.n);
}

void launch_gauss_block(gingr_ctx *ctx, Cloud A, Cloud B, double sigma, double scaling, double *out) {
    // gridDim.y is limited to 65535 rows per launch
    const int64_t max_rows = 65535;
    for (int64_t r0 = 0; r0 < A.n; r0 += max_rows) {
        const int64_t nr = A.n - r0 < max_rows ? A.n - r0 : max_rows;
        Cloud sub{A.x + r0, A.y + r0, A.z + r0, nr};
        dim3 grid((unsigned)ceil_div(B.n, kBlock), (unsigned)nr);
        hipLaunchKernelGGL(gauss_block_kernel, grid, dim3(kBlock), 0, ctx->stream, sub, B, sigma, scaling, out + r0 * B.n);
    }
}

int64_t sumsq_pairs_ws_doubles(int64_t nA) { return std::max<int64_t>(ceil_div(nA, kBlock), 2 * kMomentBlocks * 4); }

void launch_sumsq_pairs(gingr_ctx *ctx, Cloud A, Cloud B, double *ws, double *out_scalar) {
    if (A.n * B.n >= (int64_t)1 << 20) {  // (small problems keep the pair loop: nothing to gain, and its bits are what the tests of old pin)
        hipLaunchKernelGGL(cloud_moments_kernel, dim3(kMomentBlocks, 2), dim3(kBlock), 0, ctx->stream, A, B, ws);
        hipLaunchKernelGGL(sumsq_from_moments_kernel, dim3(1), dim3(64), 0, ctx->stream, ws, A.n, B.n, out_scalar);
        return;
    }
    const int64_t nb = ceil_div(A.n, kBlock);
    hipLaunchKernelGGL(sumsq_pairs_kernel, dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream, A, B, ws);
    hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(1024), 0, ctx->stream, ws, nb, 1.0, out_scalar);
}

void launch_aos_to_soa(gingr_ctx *ctx, const double *aos, int64_t n, double *soa, const int32_t *perm) {
    if (n <= 0) return;
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, aos, n, perm, soa);
}

void launch_soa_to_aos(gingr_ctx *ctx, const double *soa, int64_t n, double *aos, const int32_t *perm) {
    if (n <= 0) return;
    hipLaunchKernelGGL(soa_to_aos_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, soa, n, perm, aos);
}

void launch_scatter(gingr_ctx *ctx, const double *in, int64_t n, const int32_t *perm, double *out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, in, n, perm, out);
}

// Spatial order for the tile culling: recursive median split along the longest axis of the bounding box (a balanced k-d
// tree laid out in leaf order).  Every aligned run of 256 * 2^k points is one tree node, i.e. a compact box, which is what
// the per-tile / per-workgroup bounding boxes of the CPD kernels need (a Z-curve order has seams whose chunks span the
// whole domain).  Deterministic: ties are broken by the original index.
// The points travel with their index (32-byte records, permuted in place): every pass is a contiguous sweep, and the two halves of
// the upper levels go to separate threads.  The result does not depend on either -- each split is the unique median cut of the total
// order (coordinate, original index), and the quarters are sorted by index at the end.
struct KdPoint {
    double c[3];
    int64_t idx;
};

static void kd_split(KdPoint *p, int64_t n, int64_t leaf, int par_levels) {
    if (n <= leaf) return;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            const double v = p[i].c[d];
            if (v == v) {
                if (v < lo[d]) lo[d] = v;
                if (v > hi[d]) hi[d] = v;
            }
        }
    int ax = 0;
    for (int d = 1; d < 3; ++d)
        if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
    // left half gets a multiple of `leaf` points so that leaves stay aligned to 256-point tiles
    int64_t half = ((n / leaf + 1) / 2) * leaf;
    if (half >= n) half = n / 2;
    std::nth_element(p, p + half, p + n, [ax](const KdPoint &a, const KdPoint &b) {
        const double ka = a.c[ax] == a.c[ax] ? a.c[ax] : 1e300, kb = b.c[ax] == b.c[ax] ? b.c[ax] : 1e300;  // NaN coordinates sort last
        return ka < kb || (ka == kb && a.idx < b.idx);
    });
    if (par_levels > 0 && n >= 16384) {
        std::thread left;
        try {
            left = std::thread([=] { kd_split(p, half, leaf, par_levels - 1); });
        } catch (const std::system_error &) {  // no thread to be had: this half inline as well
            kd_split(p, half, leaf, 0);
        }
        kd_split(p + half, n - half, leaf, par_levels - 1);
        if (left.joinable()) left.join();
    } else {
        kd_split(p, half, leaf, 0);
        kd_split(p + half, n - half, leaf, 0);
    }
}

void morton_order(const double *xyz, int64_t n, std::vector<int32_t> &perm) {
    std::vector<KdPoint> pts((size_t)n);
    for (int64_t i = 0; i < n; ++i) pts[(size_t)i] = KdPoint{{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, i};
    kd_split(pts.data(), n, 256, 4);  // up to sixteen threads
    // every 256-point leaf is split further into four spatially compact 64-point quarters (the unit of the fine exact-zero
    // culling: one owned slot of a wave, one quarter of a streamed tile); original order inside a quarter (reproducible)
    perm.resize((size_t)n);
    auto leaves = [&](int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; b += 256) {
            const int64_t m = b + 256 < n ? 256 : n - b;
            kd_split(pts.data() + b, m, 64, 0);
            for (int64_t i = 0; i < m; ++i) perm[(size_t)(b + i)] = (int32_t)pts[(size_t)(b + i)].idx;
            for (int64_t q = 0; q < m; q += 64) std::sort(perm.begin() + b + q, perm.begin() + b + (q + 64 < m ? q + 64 : m));
        }
    };
    const int64_t nleaves = (n + 255) / 256;
    if (n >= 16384) {
        const int nt = 8;
        std::thread th[nt - 1];
        for (int t = 1; t < nt; ++t) {
            const int64_t b0 = nleaves * t / nt * 256, b1 = nleaves * (t + 1) / nt * 256;
            try {
                th[t - 1] = std::thread(leaves, b0, b1);
            } catch (const std::system_error &) {
                leaves(b0, b1);
            }
        }
        leaves(0, nleaves / nt * 256);
        for (int t = 1; t < nt; ++t)
            if (th[t - 1].joinable()) th[t - 1].join();
    } else {
        leaves(0, nleaves * 256);
    }
}

#ifdef GINGR_STAMPS
// diagnostic build only (not in include/gingr_hip.h): allocate / read back the stamp buffer of the two pair loops
extern "C" int gingr_debug_stamps_enable(gingr_ctx *ctx) {
    unsigned long long *buf = nullptr;
    const size_t bytes = (size_t)2 * kStampWaves * 8 * sizeof(unsigned long long);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&buf), bytes));
    HIP_TRY(ctx, hipMemset(buf, 0, bytes));
    HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)));
    return GINGR_OK;
}
extern "C" int gingr_debug_stamps_read(gingr_ctx *ctx, unsigned long long *host /* [2][32768][8] */) {
    unsigned long long *buf = nullptr;
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_stamp_buf), sizeof(buf)));
    if (!buf) return gingr_set_error(ctx, GINGR_ERR_STATE, "stamps not enabled");
    HIP_TRY(ctx, hipMemcpy(host, buf, (size_t)2 * kStampWaves * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return GINGR_OK;
}
#endif

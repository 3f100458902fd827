// Weighted Gram Q^T L Q for model ranks above 112 (rp = 128 .. 512, NT = rp / 16 = 8 .. 32 column tiles) on gfx950.
//
// What it replaces: the Q^T L Q of scalismo's DiscreteLowRankGaussianProcess.regression, reached from
// G/api/GingrAlgorithm.scala:297-301 -- the same product gram_tri_kernel (gp.hip) computes for rp <= 112, at the ranks the
// reference's untruncated demo models have (E/CreateBunnyGPMM.scala, E/DemoHelper/DemoDatasetLoader.scala:22;
// G/api/registration/utils/GPMMHelper.scala:39-69).
//
// gram_tri_kernel keeps the whole upper triangle of G in the registers of a PAIR of waves, which stops at NT = 7 (28 tiles).  Here
// the NT (NT + 1) / 2 tiles of the triangle are dealt, in row-major order, to the EIGHT waves of a workgroup (T tiles each: 17 at
// rp = 256, no tile computed twice, no padded tile multiplied) and, past 8 x 17 tiles, to several workgroups per slab ("parts").
// All waves walk the same rows: a step is 4 SUB rows; every wave fetches its share of the step's SUB x NT fragments (a fragment =
// the 4 x 16 block of the basis one v_mfma_f64_16x16x4 takes as an operand, one value per lane), scales it (A operand: w * fragment)
// and leaves both forms in LDS, from where each wave reads the two operands of each of its tiles.  One barrier per step, two LDS
// buffers; the global loads of step s + 1 are in flight during the MFMAs of step s.
//
// What bounds it (tools/ubench_mfma_f64_fill.hip, profiles/r03_ubench_mfma_f64_fill.txt): a float64 MFMA occupies its SIMD for 64
// cycles and no other VECTOR instruction issues beside it; LDS traffic, scalar instructions and the issue of global loads are free.
// So the loop issues as few vector instructions as possible:
//   * the weight and the right-hand-side value of a basis ROW come as one 16-byte load from a row-indexed {w, e} array that a tiny
//     launch in front of this kernel fills (row_expand_kernel) -- no row -> (point, coordinate) arithmetic in the loop;
//   * every address is a wave-uniform base (scalar registers, advanced by scalar adds) plus a per-lane constant;
//   * the LDS addresses of a wave's tiles are computed once (two registers per tile), the buffer / sub-step parts are immediates.
// The right-hand side Q0^T e rides along as in gram_tri_kernel (one FMA per fragment value).
// Output: the same [slab][rp x rp] partial layout, upper tiles only; reduced by phase1_finalize_kernel / gram_reduce_kernel (gp.hip).
#include "gp.h"

#include <algorithm>

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kWaves = 8;
constexpr int kRowPad = 48;  // {w, e} entries behind the last basis row (zeros): what the steps past the end read
static_assert(kBasisRowSlack >= 3 * kRowPad, "the three-rows-per-point view of the basis (launch_gram_rows) reads 48 of ITS rows past the end");

// we[row] = {weight of the row's point (1 without weights), right-hand-side value of the row (0 without evec)}; zeros behind 3 M
__global__ __launch_bounds__(256) void row_expand_kernel(const double *__restrict__ weight, const double *__restrict__ evec, int64_t M,
                                                         d2 *__restrict__ we) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= 3 * M + kRowPad) return;
    d2 v = d2{0.0, 0.0};
    if (row < 3 * M) {
        const int64_t pt = row / 3;
        const int c = (int)(row - 3 * pt);
        v[0] = weight ? weight[pt] : 1.0;
        v[1] = evec ? evec[c * M + pt] : 0.0;
    }
    we[row] = v;
}

// the unweighted product of plain rows: we[row] = {1, 0}, zeros behind `rows`
__global__ __launch_bounds__(256) void row_ones_kernel(int64_t rows, d2 *__restrict__ we) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows + kRowPad) return;
    we[row] = d2{row < rows ? 1.0 : 0.0, 0.0};
}

struct GramWideArgs {
    const double *Q0;
    int64_t rows;  // 3 M
    int32_t rp, NT;
    const d2 *we;  // [rows + kRowPad]
    int64_t rows_per_slab;  // a multiple of 4 SUB
    double *partial;        // [nslabs][rp * rp]
    double *rhs_partial;    // nullable: [nslabs][rp]
    int32_t nparts, tiles_per_part;
    ZeroGate gate;  // (surface ICP: this pass runs only when the downdate in front of it left at once, gp.h)
};

// T: tiles per wave; SUB: 4-row sub-steps per barrier; the LDS buffers are laid out for NTCAP = 32 / SUB column tiles
template <int T, int SUB>
__global__ __launch_bounds__(64 * kWaves) void gram_wide_kernel(GramWideArgs A) {
    constexpr int NTCAP = 32 / SUB;
    constexpr int kSub = NTCAP * 128;  // doubles of one sub-step: [NTCAP][plain, scaled][64]
    constexpr int kBuf = SUB * kSub;   // 4 096 doubles = 32 KB per buffer
    constexpr int FO = 4;              // fragments a wave fetches per step, at most: SUB * NTCAP / kWaves
    constexpr int kStepRows = 4 * SUB;
    extern __shared__ double xb[];     // [2][SUB][NTCAP][2][64]
    if (!gate_open(A.gate)) return;    // (workgroup-uniform)
    const int lane = threadIdx.x & 63, kq = lane >> 4, cl = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = A.NT, rp = A.rp;
    const int part = (int)(blockIdx.x % (unsigned)A.nparts);
    const int64_t slab = blockIdx.x / (unsigned)A.nparts;
    const int64_t r0 = slab * A.rows_per_slab;
    const int64_t r1 = r0 + A.rows_per_slab < A.rows ? r0 + A.rows_per_slab : A.rows;
    const int64_t nsteps = (r1 - r0 + kStepRows - 1) / kStepRows;

    // ---- this wave's tiles: row-major over the upper triangle, [g0, g0 + ntile)
    const int total = NT * (NT + 1) / 2;
    const int pend = min(total, (part + 1) * A.tiles_per_part);
    const int g0 = part * A.tiles_per_part + wave * T;
    const int ntile = max(0, min(T, pend - g0));
    int t0 = 0, u0 = 0;
    {
        int rem = min(g0, total - 1);
        while (rem >= NT - t0) rem -= NT - t0, ++t0;
        u0 = t0 + rem;
    }
    const double *pa[T], *pb[T];  // LDS addresses of the A (scaled, tile row t) and B (plain, tile column u) fragments, buffer 0, sub-step 0
    {
        int t = t0, u = u0;
#pragma unroll
        for (int q = 0; q < T; ++q) {
            // a slot past the wave's share multiplies tile (t0, u0) again: never written out (only the last wave of a part has any)
            const int tt = q < ntile ? t : t0, uu = q < ntile ? u : u0;
            pa[q] = xb + (2 * tt + 1) * 64 + lane;
            pb[q] = xb + (2 * uu) * 64 + lane;
            if (++u == NT) ++t, u = t;
        }
    }
    v4f64 acc[T];
#pragma unroll
    for (int q = 0; q < T; ++q) acc[q] = v4f64{0, 0, 0, 0};

    // ---- this wave's share of the step's fragments: j = wave + 8 i < SUB * NT  ->  (sub-step j / NT, column tile j % NT).  A slot
    // past the step's fragments repeats the wave's first one (the same values to the same LDS words; left out of the right-hand
    // side at the end): no branch in the loop.  Rows past the basis read the kBasisRowSlack zero rows behind it (gp.h) and the
    // zero {w, e} entries behind the last row: no clamping either.
    const int nfr_all = SUB * NT;
    const double *qbase[FO];  // wave-uniform: row r0 + 4 sb of column 16 f
    const d2 *wbase[FO];      // wave-uniform: row r0 + 4 sb
    double *lw[FO];           // LDS, buffer 0: plain form (scaled: + 64)
    bool fvalid[FO];
#pragma unroll
    for (int i = 0; i < FO; ++i) {
        const int j = wave + kWaves * i;
        fvalid[i] = j < nfr_all;
        const int jc = fvalid[i] ? j : wave;  // (SUB * NT >= 8: slot 0 is always a fragment of its own)
        const int sb = jc / NT, f = jc - sb * NT;
        qbase[i] = A.Q0 + (r0 + 4 * sb) * rp + 16 * f;
        wbase[i] = A.we + r0 + 4 * sb;
        lw[i] = xb + sb * kSub + f * 128 + lane;
    }
    const unsigned lane_q = (unsigned)(kq * rp + cl);
    const int64_t qstep = (int64_t)kStepRows * rp;
    double fn[FO], racc[FO];
    d2 wn[FO];
#pragma unroll
    for (int i = 0; i < FO; ++i) racc[i] = 0.0;

    auto fetch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FO; ++i) {
            fn[i] = qbase[i][lane_q];
            wn[i] = wbase[i][kq];
            qbase[i] += qstep;
            wbase[i] += kStepRows;
        }
    };
    auto hand_over = [&](auto buf) __attribute__((always_inline)) {
        constexpr int B = decltype(buf)::value;
#pragma unroll
        for (int i = 0; i < FO; ++i) {
            lw[i][B * kBuf] = fn[i];
            lw[i][B * kBuf + 64] = fn[i] * wn[i][0];
            racc[i] = __builtin_fma(fn[i], wn[i][1], racc[i]);
        }
    };
    // The tile x sub-step products of a step, in the order k = sb * T + q, as PAIRS: the four operands of pair p + 1 are requested from
    // LDS before the two MFMAs of pair p are issued, and nothing else is in flight (left to itself the scheduler requests every
    // operand of the step up front: 70 spilled registers at T = 17).
    constexpr int TS = T * SUB, NP = (TS + 1) / 2;
    double fa[2][2], fb[2][2];
    auto request = [&](auto buf, int p) __attribute__((always_inline)) {  // (p: a constant after unrolling)
        constexpr int B = decltype(buf)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * p + h;
            if (k < TS) {
                const int sb = k / T, q = k - sb * T;
                fa[p & 1][h] = pa[q][B * kBuf + sb * kSub];
                fb[p & 1][h] = pb[q][B * kBuf + sb * kSub];
            }
        }
    };
    auto multiply = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * p + h;
            if (k < TS) {
                const int q = k % T;
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[p & 1][h], fb[p & 1][h], acc[q], 0, 0, 0);
            }
        }
    };
    using std::integral_constant;
    constexpr integral_constant<int, 0> c0{};
    constexpr integral_constant<int, 1> c1{};
    // prologue: step 0 into buffer 0, step 1 requested
    fetch();
    hand_over(c0);
    fetch();
    __syncthreads();
    // one step: the MFMAs on buffer B with the hand-over of the next step (requested one step ago) in the middle, then the request
    // for the step after that.  The last step of the slab hands nothing over: the rows behind it belong to the next slab (their
    // right-hand-side values must not be counted here).
    auto step = [&](auto buf, bool more) __attribute__((always_inline)) {
        constexpr int B = decltype(buf)::value;
        request(buf, 0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p + 1 < NP) request(buf, p + 1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(p);
            __builtin_amdgcn_sched_barrier(0);
            if (p == NP / 2) {
                if (more) {  // wave-uniform
                    hand_over(integral_constant<int, 1 - B>{});
                    fetch();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    int64_t st = 0;
    for (; st + 1 < nsteps; st += 2) {
        step(c0, true);
        step(c1, st + 2 < nsteps);
    }
    if (st < nsteps) step(c0, false);

    // ---- right-hand side: the four row lanes of a column, then the sub-steps of a column tile in ascending order
    if (A.rhs_partial && part == 0) {  // workgroup-uniform
        double *rsh = xb;              // [SUB * NT][16] (the buffers are free: every wave is behind the last barrier)
#pragma unroll
        for (int i = 0; i < FO; ++i) {
            racc[i] += __shfl_xor(racc[i], 16);
            racc[i] += __shfl_xor(racc[i], 32);
            if (fvalid[i] && kq == 0) rsh[(wave + kWaves * i) * 16 + cl] = racc[i];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < rp; c += 64 * kWaves) {
            const int f = c >> 4;
            double s = rsh[f * 16 + (c & 15)];
            for (int sb = 1; sb < SUB; ++sb) s += rsh[(sb * NT + f) * 16 + (c & 15)];
            A.rhs_partial[slab * rp + c] = s;
        }
    }
    // ---- D[i][j] of tile (t, u): i = kq + 4 reg is the A-side index (column 16 t + i of Q0), j = cl the B-side index
    double *out = A.partial + slab * rp * rp;
    {
        int t = t0, u = u0;
#pragma unroll
        for (int q = 0; q < T; ++q) {
            if (q < ntile) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) out[(int64_t)(16 * t + kq + 4 * reg) * rp + 16 * u + cl] = acc[q][reg];
            }
            if (++u == NT) ++t, u = t;
        }
    }
}

struct WidePlan {
    int NT, SUB, T, nparts, tiles_per_part, nslabs;
    int64_t rows_per_slab;
};

// T <= 17 accumulator tiles per wave (136 registers): parts = workgroups per slab so that a part's tiles fit eight waves
WidePlan wide_plan_rows(int64_t rows, int32_t rp, int max_workgroups) {
    WidePlan p;
    p.NT = rp / 16;
    p.SUB = p.NT <= 8 ? 4 : (p.NT <= 16 ? 2 : 1);
    const int total = p.NT * (p.NT + 1) / 2;
    p.nparts = (total + kWaves * 17 - 1) / (kWaves * 17);
    p.tiles_per_part = (total + p.nparts - 1) / p.nparts;
    p.T = (p.tiles_per_part + kWaves - 1) / kWaves;
    // one workgroup per compute unit; small shards keep at least 64 rows per slab (never more slabs than gram_tri_kernel's plan:
    // the right-hand-side partials live in the sweep workspace)
    const int64_t want = std::min<int64_t>(std::max(1, max_workgroups / p.nparts), std::max<int64_t>(1, ceil_div(rows, 64)));
    p.rows_per_slab = round_up(ceil_div(rows, want), 4 * p.SUB);
    p.nslabs = (int)ceil_div(rows, p.rows_per_slab);
    return p;
}
WidePlan wide_plan(int64_t M, int32_t rp) { return wide_plan_rows(3 * M, rp, 256); }

}  // namespace

int64_t gram_wide_ws_doubles(int64_t M, int32_t rp) {
    const WidePlan p = wide_plan(M, rp);
    return (int64_t)p.nslabs * rp * rp + 2 * (3 * M + kRowPad);
}

// see gp.h; ws: gram_wide_ws_doubles(M, rp) doubles
static int gram_wide_go(gingr_ctx *ctx, const double *Q0, int64_t rows, int32_t rp, const WidePlan &p, const d2 *we, double *ws,
                        double *rhs_partial, const ZeroGate *gate) {
    GramWideArgs a;
    a.Q0 = Q0;
    a.rows = rows;
    a.rp = rp;
    a.NT = p.NT;
    a.we = we;
    a.rows_per_slab = p.rows_per_slab;
    a.partial = ws;
    a.rhs_partial = rhs_partial;
    a.nparts = p.nparts;
    a.tiles_per_part = p.tiles_per_part;
    a.gate = gate ? *gate : ZeroGate{};
    const size_t lds = (size_t)2 * 4096 * sizeof(double);
    const dim3 grid((unsigned)(p.nslabs * p.nparts)), block(64 * kWaves);
    auto go = [&](auto kern) {
        set_dynamic_lds(kern, (size_t)(lds));
        TimerScope ts(ctx, 2);  // (the Gram pass itself: row_expand_kernel stays outside, as in the rocprof statistics)
        hipLaunchKernelGGL(kern, grid, block, lds, ctx->stream, a);
    };
#define GINGR_WIDE(t, sub) \
    case t: go(gram_wide_kernel<t, sub>); break;
    if (p.SUB == 4) {
        go(gram_wide_kernel<5, 4>);  // NT = 8: 36 tiles
    } else if (p.SUB == 2) {
        switch (p.T) {  // NT = 9 .. 16
            GINGR_WIDE(6, 2) GINGR_WIDE(7, 2) GINGR_WIDE(9, 2) GINGR_WIDE(10, 2) GINGR_WIDE(12, 2) GINGR_WIDE(14, 2) GINGR_WIDE(15, 2)
            default: go(gram_wide_kernel<17, 2>); break;
        }
    } else {
        switch (p.T) {  // NT = 17 .. 32, two to four parts
            GINGR_WIDE(10, 1) GINGR_WIDE(11, 1) GINGR_WIDE(12, 1) GINGR_WIDE(13, 1) GINGR_WIDE(14, 1) GINGR_WIDE(15, 1) GINGR_WIDE(16, 1)
            default: go(gram_wide_kernel<17, 1>); break;
        }
    }
#undef GINGR_WIDE
    return p.nslabs;
}

int launch_gram_wide(gingr_ctx *ctx, const double *Q0, int64_t M, int32_t rp, const double *weight, double *ws, const double *evec,
                     double *rhs_partial, const ZeroGate *gate) {
    const WidePlan p = wide_plan(M, rp);
    d2 *we = reinterpret_cast<d2 *>(ws + (int64_t)p.nslabs * rp * rp);
    hipLaunchKernelGGL(row_expand_kernel, dim3((unsigned)ceil_div(3 * M + kRowPad, 256)), dim3(256), 0, ctx->stream, weight, evec, M, we);
    return gram_wide_go(ctx, Q0, 3 * M, rp, p, we, ws, rhs_partial, gate);
}

// Z^T Z of `rows` plain rows of width rp (128 .. 512, a multiple of 16; 48 zero rows behind them), unweighted: slab partials of the
// upper tiles in ws, the slab count returned.  One-off products (the model's moment blocks: fitter.hip): at most 96 workgroups, so
// that the partials stay small.
constexpr int kRowsWorkgroups = 96;
int64_t gram_rows_ws_doubles(int64_t rows, int32_t rp) {
    const WidePlan p = wide_plan_rows(rows, rp, kRowsWorkgroups);
    return (int64_t)p.nslabs * rp * rp + 2 * (rows + kRowPad);
}
int launch_gram_rows(gingr_ctx *ctx, const double *Z, int64_t rows, int32_t rp, double *ws) {
    const WidePlan p = wide_plan_rows(rows, rp, kRowsWorkgroups);
    d2 *we = reinterpret_cast<d2 *>(ws + (int64_t)p.nslabs * rp * rp);
    hipLaunchKernelGGL(row_ones_kernel, dim3((unsigned)ceil_div(rows + kRowPad, 256)), dim3(256), 0, ctx->stream, rows, we);
    return gram_wide_go(ctx, Z, rows, rp, p, we, ws, nullptr, nullptr);
}

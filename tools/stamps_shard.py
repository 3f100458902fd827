#!/usr/bin/env python3
"""Per-wave time stamps of the two CPD pair loops on a row shard (diagnostic build, see affinity.hip: GINGR_STAMPS).

usage (GPU box):  make -C gingr_amd/csrc variant NAME=stamps DEFS=-DGINGR_STAMPS        (here, cross-compiled)
                  GINGR_HIP_LIB=gingr_amd/libgingr_hip_stamps.so python3 tools/stamps_shard.py [world] [points] > profile.txt

Runs the bench workload (synthetic 50k <-> 50k, rank 100) as rank 0 of `world` ranks for a few iterations, reads the stamps of
the LAST launch of cpd_colsum_kernel / cpd_rowstats_kernel and prints where a one-round launch spends its time: dispatch skew,
prologue stages, pair loop, epilogue, how the waves of one SIMD finish, workgroups per compute unit, the clock the waves saw."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (first: one HIP runtime per process)
import gingr_amd as ga  # noqa: E402
from gingr_amd.sharded import ShardedFitter  # noqa: E402
from bench import synth_clouds  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
points = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
y, x = synth_clouds(points)
ctx = ga.Context(0)
lib = ctx._lib
lib.gingr_debug_stamps_enable.argtypes = [ctypes.c_void_p]
lib.gingr_debug_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
SINGLE = len(sys.argv) > 3 and sys.argv[3] == "single"
if SINGLE:
    # the same number of rows as rank 0 of `world` ranks, but as a registration of its own on one shard: the same pair loops and the same
    # narrow tail, no exchange kernels between them
    y = y[: (points + world - 1) // world]
model = ga.GPMMTriangleMesh3D(ctx, y, relativeTolerance=0.0, maxRank=100).Gaussian(70.0, 50.0)
if SINGLE:
    fitter = ShardedFitter(ctx, model, x)
else:
    uid = ctx.rccl_unique_id()
    ctx.rccl_init(uid, 1, 0)
    fitter = ShardedFitter(ctx, model, x, rank=0, world=world, all_reduce=None, rccl=True)
s2 = ctx.cpd_initial_sigma2(y, x)
fitter.set_state(np.zeros(100), s2)
fitter.update_cpd(0.1, 1.0, 5)
ctx.synchronize()
assert lib.gingr_debug_stamps_enable(ctx.handle) == 0
if len(sys.argv) > 3 and sys.argv[3] == "colsum_back_to_back":
    # the column-sum pass alone, N launches back to back (phase 0 of the phase API): what clock do its waves see when nothing short and
    # narrow runs between the launches?  (the full iteration has ~60 us of single-workgroup kernels per 0.4 ms)
    from gingr_amd import _native as nat
    p = nat.CpdParams(0.1, 1.0)
    for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 200):
        assert lib.gingr_fitter_cpd_phase_async(fitter.handle, ctypes.byref(p), 0) == 0
elif len(sys.argv) > 3 and sys.argv[3] in ("phases01", "phases012"):
    # phases 0 and 1 (both pair loops, the reductions, the Gram pass) without / with phase 2 (the replicated narrow tail), no exchange
    from gingr_amd import _native as nat
    p = nat.CpdParams(0.1, 1.0)
    for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 100):
        for ph in ((0, 1) if sys.argv[3] == "phases01" else (0, 1, 2)):
            assert lib.gingr_fitter_cpd_phase_async(fitter.handle, ctypes.byref(p), ph) == 0
else:
    fitter.update_cpd(0.1, 1.0, 3)
ctx.synchronize()
W = 1 << 15
buf = np.zeros((2, W, 8), dtype=np.uint64)
assert lib.gingr_debug_stamps_read(ctx.handle, buf.ctypes.data) == 0
m_loc = fitter.end - fitter.begin
print(f"# row shard: rank 0 of {world}, {m_loc} local rows x {points} targets, rank 100; stamps of the last launch of each pair loop")
print("# wall clock = 100 MHz (10 ns); all times in us relative to the first wave's entry")

for k, name in enumerate(("cpd_colsum_kernel", "cpd_rowstats_kernel")):
    b = buf[k]
    used = b[:, 0] != 0
    nw = int(used.sum())
    if nw == 0:
        print(f"\n== {name}: no stamps")
        continue
    b = b[used].astype(np.int64)
    t0 = b[:, 0].min()
    T = (b[:, :6] - t0) / 100.0            # us
    cyc = b[:, 6]
    hw = b[:, 7] & 0xFFFFFFFF
    xcc = (b[:, 7] >> 32) & 0xF
    # gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se 3 bits)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    dur = T[:, 5] - T[:, 0]
    q = lambda v, p: float(np.percentile(v, p))
    print(f"\n== {name}: {nw} waves = {nw // 4} workgroups; launch spans {T[:, 5].max():.1f} us (first entry to last exit)")
    print(f"entry skew (dispatch):           median {q(T[:, 0], 50):6.2f}  p90 {q(T[:, 0], 90):6.2f}  max {T[:, 0].max():6.2f}")
    st = [("owned points + boxes ready", 1, 0), ("table barrier passed", 2, 1), ("first quarter staged", 3, 2), ("pair loop", 4, 3),
          ("epilogue (combine + store)", 5, 4)]
    for label, a, bb in st:
        d = T[:, a] - T[:, bb]
        print(f"{label:32s} median {q(d, 50):6.2f}  p10 {q(d, 10):6.2f}  p90 {q(d, 90):6.2f}  max {d.max():6.2f}")
    pro = T[:, 3] - T[:, 0]
    print(f"whole prologue (entry -> first pairs) median {q(pro, 50):6.2f}  p90 {q(pro, 90):6.2f}  max {pro.max():6.2f}")
    print(f"pair loop END times:             median {q(T[:, 4], 50):6.2f}  p10 {q(T[:, 4], 10):6.2f}  p90 {q(T[:, 4], 90):6.2f}  max {T[:, 4].max():6.2f}")
    ghz = cyc / (dur * 1e3)
    print(f"core clock seen by the waves (cycles / wall): median {q(ghz, 50):.3f} GHz  p10 {q(ghz, 10):.3f}  p90 {q(ghz, 90):.3f}")
    # waves per SIMD and their finishing order
    order = np.argsort(simd_key, kind="stable")
    keys, starts, counts = np.unique(simd_key[order], return_index=True, return_counts=True)
    hist = np.bincount(counts)
    print("waves per SIMD: " + ", ".join(f"{c} on {n}" for c, n in enumerate(hist) if n))
    cuk, cuc = np.unique(cu_key, return_counts=True)
    h2 = np.bincount(cuc // 4)
    print(f"compute units used: {len(cuk)}; workgroups per unit: " + ", ".join(f"{c} on {n}" for c, n in enumerate(h2) if n))
    fin = []
    full = counts.max()
    for s, c in zip(starts, counts):
        if c == full:
            fin.append(np.sort(T[order[s:s + c], 4]))
    if fin:
        fin = np.array(fin)
        print(f"SIMDs with {full} waves: mean pair-loop end of the 1st..{full}th finisher: " + " / ".join(f"{v:.1f}" for v in fin.mean(0)))
    # where the late finishers sit: per XCD (each has its own clock domain / L2) and the latest waves one by one
    print("per XCD: waves, median core clock GHz, median / max pair-loop end, median pair-loop duration")
    for xc in sorted(set(xcc.tolist())):
        m = xcc == xc
        print(f"  xcd {xc}: {int(m.sum()):5d}  {np.median(ghz[m]):.3f}  {np.median(T[m, 4]):7.1f} / {T[m, 4].max():7.1f}  {np.median((T[:, 4] - T[:, 3])[m]):7.1f}")
    late = np.argsort(-T[:, 4])[:12]
    wgid = np.flatnonzero(used)[late] // 4
    print("latest pair-loop ends: " + ", ".join(f"{T[i, 4]:.1f}us(xcd{xcc[i]} se{se[i]} cu{cu[i]} simd{simd[i]} wg{w} start{T[i, 3]:.1f} clk{ghz[i]:.2f})" for i, w in zip(late, wgid)))
    # per SIMD: total pair-loop time of its waves vs its last end (is a late SIMD one with more work, or a slower one?)
    tot = np.array([(T[order[s:s + c], 4] - T[order[s:s + c], 3]).sum() for s, c in zip(starts, counts)])
    last = np.array([T[order[s:s + c], 4].max() for s, c in zip(starts, counts)])
    fullm = counts == full
    if fullm.sum() > 4:
        cc = np.corrcoef(tot[fullm], last[fullm])[0, 1]
        print(f"SIMDs with {full} waves: last end p10 {np.percentile(last[fullm], 10):.1f} median {np.median(last[fullm]):.1f} p90 {np.percentile(last[fullm], 90):.1f} "
              f"max {last[fullm].max():.1f}; correlation(sum of loop times, last end) {cc:.2f}")
    # issue-slot use: the pair loop's VALU issue cycles against what the SIMD had
    loop = T[:, 4] - T[:, 3]
    print(f"sum over a SIMD's waves of pair-loop time / (waves x launch span): {loop.sum() / (nw * T[:, 5].max()):.3f}")
ctx_close = getattr(ctx, "close", None)
fitter.close()
if ctx_close:
    ctx_close()

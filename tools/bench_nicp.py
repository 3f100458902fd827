#!/usr/bin/env python3
"""Optimal-step non-rigid ICP (the reference's other/ N-ICP-T / N-ICP-A baselines) on the femur pair: time per iteration and the
surface distance reached.   python tools/bench_nicp.py [stages=3] [inner=3]"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gingr_amd as ga  # noqa: E402
from gingr_amd import classic  # noqa: E402

stages = int(sys.argv[1]) if len(sys.argv) > 1 else 3
inner = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = np.load(os.path.join(HERE, "..", "tests", "golden", "inputs.npz"))
m = np.load(os.path.join(HERE, "..", "tests", "golden", "femur_mesh.npz"))
tv, gv = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
tc, gc = m["femur_cells"], m["femur_target_cells"]
lm_t = {f"L{i}": p for i, p in enumerate(d["femur_lm"].astype(np.float64))}
lm_g = {f"L{i}": p for i, p in enumerate(d["femur_target_lm"].astype(np.float64))}
ctx = ga.Context(0)
out = {"what": "optimal-step non-rigid ICP on the femur pair (1 622 vertices, unique edges of its triangulation; landmarks L0-L5)", "stages": stages, "inner": inner}
alphas = [10.0, 5.0, 2.0, 1.0, 0.5][:stages]
for kind in ("T", "A"):
    task = classic.NonRigidOptimalStepICP(ctx, (tv, tc), (gv, gc), lm_t, lm_g, kind=kind)
    task.Iteration(tv, 10.0, 10.0)   # warm-up (allocations, code load)
    ctx.synchronize()
    t0 = time.perf_counter()
    fit = task.Registration(inner, tolerance=1e-3, alpha=alphas, beta=alphas)
    dt = time.perf_counter() - t0
    before = ctx.mesh_distance_stats(tv, gv, gc)
    after = ctx.mesh_distance_stats(fit, gv, gc)
    out[kind] = {"iterations": task.iterations, "ms_per_iteration": 1e3 * dt / task.iterations, "unknowns": int(tv.shape[0] * (1 if kind == "T" else 4)),
                 "avg_surface_distance_before": float(before[0]) / tv.shape[0], "avg_surface_distance_after": float(after[0]) / tv.shape[0]}
    task.close()
print(json.dumps(out))

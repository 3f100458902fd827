#!/usr/bin/env python3
"""Diagnosis: device-resident CPD iterations on the real femur pair -- time per iteration as a function of how many are enqueued at
once, for w = 0 / 0.1 and NoTransforms / Rigid (config 1 measured 0.32 ms per iteration against 0.11 ms of kernel time)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import gingr_amd as ga
from gingr_amd.sharded import ShardedFitter
d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "inputs.npz"))
ref, target = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
ctx = ga.Context(0)
model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).Gaussian(sigma=70.0, scaling=50.0)
rank = int(model.rank)
s2 = ctx.cpd_initial_sigma2(ref, target)
for w in (0.0, 0.1):
    for tr in (0, 1):
        f = ShardedFitter(ctx, model, target, global_transform=tr, step_length=1.0)
        for n in (5, 20, 50, 100, 300):
            f.set_state(np.zeros(rank), s2)
            f.update_cpd(w, 1.0, 3)
            ctx.synchronize()
            f.set_state(np.zeros(rank), s2)
            ctx.synchronize()
            t0 = time.perf_counter()
            f.update_cpd(w, 1.0, n)
            t1 = time.perf_counter()
            ctx.synchronize()
            t2 = time.perf_counter()
            _, sc, _ = f.get_state()
            print(json.dumps({"w": w, "transform": tr, "n": n, "enqueue_ms_per_it": (t1 - t0) / n * 1e3, "ms_per_it": (t2 - t0) / n * 1e3,
                              "sigma2": sc.sigma2, "status": sc.status}), flush=True)
        f.close()

#!/usr/bin/env python3
"""Rate of the per-iteration host-boundary call at the metric size: every iteration pushes the state (alpha, pose), runs one CPD
update on the device and pulls the new state and the 1.2 MB fit back over PCIe -- what a host that drives `update` call by call
sees (the bench's `value` keeps the iterations on the device).   python tools/bench_host_boundary.py [points=50000] [steps=40]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (synthetic workload of the metric)
import gingr_amd as ga  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
y, x = bench.synth_clouds(n)
U, lam = bench.synth_gpmm(y, 100)
ctx = ga.Context(0)
model = ga.PointDistributionModel(y, np.zeros_like(y), U, lam)
algo = ga.CpdRegistration(ctx)
s = algo.createInitialState(model, x, ga.CpdConfiguration(maxIterations=10000, w=0.1))
for _ in range(3):
    s = algo.update(s)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    s = algo.update(s)
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"what": "host-boundary update() incl. state push + state / fit pull over PCIe (pinned staging)", "points": n,
                  "iterations_per_s": 1.0 / dt, "ms_per_iteration": 1e3 * dt, "fit_bytes_per_iteration": 24 * n,
                  "status": int(s.general.status)}))

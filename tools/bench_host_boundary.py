#!/usr/bin/env python3
"""PCIe-inclusive rate of the per-iteration host boundary: one Python `update()` call per iteration pushes the state
(alpha, pose, sigma2), runs the update and pulls the state + the full fit (3M doubles) back.  Reported next to the bench
JSON; never the benchmark `value`."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: see INTEGRATION.md section 3)
import gingr_amd as ga
from bench import synth_clouds, synth_gpmm

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
y, x = synth_clouds(M)
basis, lam = synth_gpmm(y, 100)
ctx = ga.Context(0)
algo = ga.CpdRegistration(ctx)
model = ga.PointDistributionModel(y, np.zeros_like(y), basis, lam)
state = algo.createInitialState(model, x, ga.CpdConfiguration(maxIterations=100, w=0.1))
for _ in range(3):
    state = algo.update(state)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    state = algo.update(state)
dt = time.perf_counter() - t0
print(json.dumps({"what": "host-boundary update() incl. state push + fit pull over PCIe", "points": M,
                  "iterations_per_s": n / dt, "ms_per_iteration": dt / n * 1e3, "fit_bytes_per_iteration": 24 * M}))

#!/usr/bin/env python3
"""ICP update rate (point-cloud closest point) on the synthetic 50k <-> 50k workload: one step = nearest-neighbour search +
GP update.  usage: bench_icp.py [points] [cull=0|1] [nn_grid=0|1] [count=1]  (cull=0: full brute-force scan; nn_grid=0: tile scan only)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import gingr_amd as ga
from gingr_amd.sharded import ShardedFitter
from bench import synth_clouds, synth_gpmm

OPTS = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
POS = [a for a in sys.argv[1:] if "=" not in a]
M = int(POS[0]) if POS else 50000
y, x = synth_clouds(M)
basis, lam = synth_gpmm(y, 100)
ctx = ga.Context(0)
from gingr_amd import _native as nat
ctx.set_option(nat.OPT_CULL, int(OPTS.get("cull", 1)))
ctx.set_option(nat.OPT_NN_GRID, int(OPTS.get("nn_grid", 1)))
COUNT = OPTS.get("count", "0") == "1"
f = ShardedFitter(ctx, ga.PointDistributionModel(y, np.zeros_like(y), basis, lam), x)
f.set_state(np.zeros(100), 100.0)
f.update_icp(100.0, 1.0, 100, 3)
ctx.synchronize()
t0 = time.perf_counter()
n = 20
f.update_icp(100.0, 1.0, 100, n)
ctx.synchronize()
dt = time.perf_counter() - t0
a, sc, fit = f.get_state()
tests = None
if COUNT:  # distance tests per closest-point search (a separate, untimed iteration: the counter serialises atomics)
    ctx.nn_counting(True)
    f.update_icp(100.0, 1.0, 100, 1)
    ctx.synchronize()
    tests = ctx.nn_tests()
    ctx.nn_counting(False)
print(json.dumps({"what": "ICP update (nn + GP), synthetic clouds", "points": M, "iterations_per_s": n / dt,
                  "ms_per_iteration": dt / n * 1e3, "cull": OPTS.get("cull", "1"),
                  "nn_grid": OPTS.get("nn_grid", "1"), "distance_tests_per_search": tests, "status": sc.status,
                  "fit_checksum": float(np.abs(fit).sum())}))

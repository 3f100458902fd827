"""Timing of the on-device GPMM construction (one-off model set-up, not the benchmark metric)."""
import json
import sys
import time

import numpy as np
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch  # noqa: F401  (first: one HIP runtime per process)

import gingr_amd as ga

ctx = ga.Context(0)
out = []
for M, tol, max_rank in [(50000, 0.0, 100), (50000, 0.01, 0), (100000, 0.0, 512)]:
    ref = np.random.default_rng(1234).normal(0, 100, (M, 3))
    for rep in range(2):
        t0 = time.perf_counter()
        dm = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=tol, maxRank=max_rank).Gaussian(70.0, 50.0)
        r = dm.rank
        ctx.synchronize()
        dt = time.perf_counter() - t0
        dm.device().close()
    t0 = time.perf_counter()
    h = ga.PointSetHelper(ctx, ref)
    mx = h.maximumPointDistance()
    dt2 = time.perf_counter() - t0
    out.append({"points": M, "relative_tolerance": tol, "max_rank": max_rank, "rank": r, "build_s": dt,
                "distance_extrema_s": dt2, "max_distance": mx})
    print(out[-1], file=sys.stderr)
print(json.dumps(out))

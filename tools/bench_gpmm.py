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
    for rep in range(2):                      # (the first call of a size pays its allocations)
        t0 = time.perf_counter()
        h = ga.PointSetHelper(ctx, ref)
        mx = h.maximumPointDistance()
        dt2 = time.perf_counter() - t0
    out.append({"points": M, "relative_tolerance": tol, "max_rank": max_rank, "rank": r, "build_s": dt,
                "distance_extrema_s": dt2, "max_distance": mx})
    print(out[-1], file=sys.stderr)
# the other kernels of GPMMTriangleMesh3D on the femur reference (1 622 vertices; the Laplacian pseudo-inverse is host LAPACK,
# one-off as in the reference) and the mirrored kernel (two interleaved factorisations) at 50k
d = np.load(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests", "golden", "inputs.npz"))
m = np.load(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests", "golden", "femur_mesh.npz"))
fem, cells = d["femur"].astype(np.float64), m["femur_cells"]
g = ga.GPMMTriangleMesh3D(ctx, fem, relativeTolerance=0.01, maxRank=300, cells=cells)
big = ga.GPMMTriangleMesh3D(ctx, np.random.default_rng(1234).normal(0, 100, (50000, 3)), relativeTolerance=0.0, maxRank=100)
for name, make in [("femur Gaussian(70, 50)", lambda: g.Gaussian(70.0, 50.0)), ("femur GaussianDot", lambda: g.GaussianDot(70.0, 0.05)),
                   ("femur GaussianSymmetry(70, 50)", lambda: g.GaussianSymmetry(70.0, 50.0)),
                   ("femur InverseLaplacian(30) incl. host pinv", lambda: g.InverseLaplacian(30.0)),
                   ("50k GaussianSymmetry(70, 50) rank 100", lambda: big.GaussianSymmetry(70.0, 50.0))]:
    for rep in range(2):
        t0 = time.perf_counter()
        dm = make()
        r = dm.rank
        ctx.synchronize()
        dt = time.perf_counter() - t0
        dm.device().close()
    out.append({"model": name, "rank": r, "build_s": dt})
    print(out[-1], file=sys.stderr)
print(json.dumps(out))

#!/usr/bin/env python3
"""ICP update rate with the SURFACE correspondence (the reference's default ICP method) on a synthetic closed mesh:
icosphere of subdivision level L (L=6: 40 962 vertices, 81 920 triangles), bumpy and posed copy as the target; one step =
cell/vertex normals + closest point on the target surface + nearest target vertex + the three rejection tests + GP update.
    PYTHONPATH=. python tools/bench_icp_surface.py [level] [tri_grid=0|1] [method=TriangularClosestPoint|AlongNormalClosestPoint]
    (tri_grid=0: the tile scan alone, for same-box comparisons; method: ICP.scala:40-44)
"""
import json
import sys
import time

import numpy as np
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch  # noqa: F401  (first: one HIP runtime per process)

import gingr_amd as ga


def icosphere(level):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v), np.asarray(f, dtype=np.int32)


OPTS = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
POS = [a for a in sys.argv[1:] if "=" not in a]
level = int(POS[0]) if POS else 6
METHOD = OPTS.get("method", "TriangularClosestPoint")
verts, cells = icosphere(level)
ref = verts * 80.0
bump = 1.0 + 0.08 * np.sin(3 * verts[:, 0]) * np.cos(2 * verts[:, 1]) + 0.05 * np.sin(5 * verts[:, 2])
c, s = np.cos(0.05), np.sin(0.05)
R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
target = (ref * bump[:, None]) @ R.T + np.array([1.5, -1.0, 0.5])
ctx = ga.Context(0)
from gingr_amd import _native as nat
ctx.set_option(nat.OPT_TRI_GRID, int(OPTS.get("tri_grid", 1)))
model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.0, maxRank=100).Gaussian(40.0, 10.0)
model.cells = cells
algo = ga.IcpRegistration(ctx)
cfg = ga.IcpConfiguration(maxIterations=100, initialSigma=10.0, endSigma=1.0, correspondenceMethod=METHOD)
state = algo.createInitialState(model, target, cfg, targetCells=cells)
state = algo.update(state)          # binds, uploads the meshes, first launch
cp, w = algo.surfaceCorrespondence(state)
ctx.synchronize()
n = int(OPTS.get("n", 10))
t0 = time.perf_counter()
for _ in range(n):
    state = algo.update(state)      # host-boundary call per iteration (push state, update, pull the fit)
ctx.synchronize()
dt = time.perf_counter() - t0
# the same iterations the way GingrAlgorithm.run takes them when nobody watches the intermediate states (api.py: _run_resident): the
# state stays on the device, an iteration reads back its scalars only, coefficients and fit come back once at the end
nres = int(OPTS.get("resident", 50))
cfg_run = ga.IcpConfiguration(maxIterations=nres + 1, initialSigma=10.0, endSigma=1.0, correspondenceMethod=METHOD)
s0 = algo.createInitialState(model, target, cfg_run, targetCells=cells)
algo.run(s0)                         # warm (the first run also uploads the state)
ctx.synchronize()
t0 = time.perf_counter()
end = algo.run(s0)
ctx.synchronize()
dt_run = time.perf_counter() - t0
# ... and n updates enqueued by ONE native call (what bench.py times for the CPD metric)
from gingr_amd.sharded import ShardedFitter
f = ShardedFitter(ctx, model, target)
f.set_meshes(cells, cells)
if METHOD == "AlongNormalClosestPoint":
    assert f._lib.gingr_fitter_set_surface_method(f.handle, 1) == 0
f.set_state(np.zeros(model.rank), 10.0)
import ctypes
ip = nat.IcpParams(10.0, 1.0, 100)


def fused(k):
    rc = f._lib.gingr_fitter_update_icp_surface_async(f.handle, ctypes.byref(ip), k)
    assert rc == 0, rc


fused(3)
ctx.synchronize()
t0 = time.perf_counter()
fused(nres)
ctx.synchronize()
dt_fused = time.perf_counter() - t0
print(json.dumps({"what": "ICP update, surface correspondence (closest point on triangles + 3 rejection tests + GP)",
                  "vertices": int(ref.shape[0]), "triangles": int(cells.shape[0]), "rank": model.rank,
                  "ms_per_iteration": dt_fused / nres * 1e3, "iterations_per_s": nres / dt_fused,
                  "ms_per_iteration_run_resident": dt_run / nres * 1e3,
                  "ms_per_iteration_host_boundary": dt / n * 1e3, "iterations_per_s_host_boundary": n / dt,
                  "timing": "ms_per_iteration: %d updates in one native call, state resident; run_resident: GingrAlgorithm.run without a "
                            "call-back (the whole ICP loop is one native call, the state comes back once); host_boundary: update(state) per iteration, the fit (%d x 3 "
                            "doubles) pulled every time" % (nres, ref.shape[0]),
                  "accepted_fraction_first_iteration": float(w.mean()), "status": int(state.general.status),
                  "run_status": int(end.general.status), "run_iterations": int(end.general.iteration),
                  "sigma2": float(state.general.sigma2), "tri_grid": int(OPTS.get("tri_grid", 1)), "method": METHOD,
                  "fit_checksum": float(np.abs(np.asarray(state.general.fit)).sum())}))

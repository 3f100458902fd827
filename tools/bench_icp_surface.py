#!/usr/bin/env python3
"""ICP update rate with the SURFACE correspondence (the reference's default ICP method) on a synthetic closed mesh:
icosphere of subdivision level L (L=6: 40 962 vertices, 81 920 triangles), bumpy and posed copy as the target; one step =
cell/vertex normals + closest point on the target surface + nearest target vertex + the three rejection tests + GP update.
    PYTHONPATH=. python tools/bench_icp_surface.py [level] [tri_grid=0|1]      (tri_grid=0: the tile scan alone, for same-box comparisons)
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
cfg = ga.IcpConfiguration(maxIterations=100, initialSigma=10.0, endSigma=1.0, correspondenceMethod="TriangularClosestPoint")
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
print(json.dumps({"what": "ICP update, surface correspondence (closest point on triangles + 3 rejection tests + GP)",
                  "vertices": int(ref.shape[0]), "triangles": int(cells.shape[0]), "rank": model.rank,
                  "iterations_per_s_host_boundary": n / dt, "ms_per_iteration": dt / n * 1e3,
                  "accepted_fraction_first_iteration": float(w.mean()), "status": int(state.general.status),
                  "sigma2": float(state.general.sigma2), "tri_grid": int(OPTS.get("tri_grid", 1)),
                  "fit_checksum": float(np.abs(np.asarray(state.general.fit)).sum())}))

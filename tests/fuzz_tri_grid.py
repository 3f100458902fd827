#!/usr/bin/env python3
"""One-off randomized sweep of the triangle grid (development aid, not collected by pytest): random open / closed target meshes, random
template poses and sizes, four surface-ICP updates each with the grid forced (GINGR_OPT_TRI_GRID = 2) against the tile scan alone (0):
fits, closest points and weights must agree bit for bit.   PYTHONPATH=. python tests/fuzz_tri_grid.py [n] [seed]"""
import sys

import numpy as np
import torch  # noqa: F401

import gingr_amd as ga
from gingr_amd import _native as nat
from oracle import gingr_oracle as go

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def sheet(n, m, size, amp, seed):
    r = np.random.default_rng(seed)
    X, Y = np.meshgrid(np.linspace(-size, size, n), np.linspace(-size * m / n, size * m / n, m), indexing="ij")
    Z = amp * np.sin(X / size * 2.3) * np.cos(Y / size * 1.7) + r.normal(0, 0.02 * amp + 1e-3, X.shape)
    v = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    idx = np.arange(n * m).reshape(n, m)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    return v, np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.int32)


def torus(n, m, R, r):
    th, ph = np.meshgrid(np.linspace(0, 2 * np.pi, n, endpoint=False), np.linspace(0, 2 * np.pi, m, endpoint=False), indexing="ij")
    v = np.stack([(R + r * np.cos(ph)) * np.cos(th), (R + r * np.cos(ph)) * np.sin(th), r * np.sin(ph)], -1).reshape(-1, 3)
    idx = np.arange(n * m).reshape(n, m)
    a, b = idx, np.roll(idx, -1, 0)
    c, d = np.roll(idx, -1, 1), np.roll(np.roll(idx, -1, 0), -1, 1)
    return v, np.concatenate([np.stack([a.ravel(), b.ravel(), c.ravel()], 1), np.stack([b.ravel(), d.ravel(), c.ravel()], 1)]).astype(np.int32)


for case in range(n_cases):
    closed = bool(rng.integers(0, 2))
    if closed:
        tv, tt = torus(int(rng.integers(12, 70)), int(rng.integers(8, 40)), 30.0, float(rng.uniform(4, 12)))
        rv, rt = torus(int(rng.integers(10, 40)), int(rng.integers(8, 24)), 30.0 * float(rng.uniform(0.9, 1.1)), float(rng.uniform(4, 12)))
    else:
        tv, tt = sheet(int(rng.integers(8, 80)), int(rng.integers(8, 80)), 40.0, float(rng.uniform(0, 8)), int(rng.integers(1 << 30)))
        rv, rt = sheet(int(rng.integers(6, 40)), int(rng.integers(6, 40)), 40.0 * float(rng.uniform(0.6, 1.3)), float(rng.uniform(0, 8)),
                       int(rng.integers(1 << 30)))
    if rng.integers(0, 4) == 0:      # a few triangles far wider than a cell
        n0 = tv.shape[0]
        tv = np.concatenate([tv, rng.normal(0, 60, (3, 3))])
        tt = np.concatenate([tt, np.array([[n0, n0 + 1, n0 + 2]], dtype=np.int32)])
    pose = (tuple(rng.normal(0, 0.05, 3)), tuple(rng.normal(0, float(rng.choice([0.3, 3.0, 30.0])), 3)))
    rank = int(rng.integers(2, 16))
    mo = go.build_gaussian_gpmm(rv, 40.0, float(rng.uniform(1, 10)), rel_tol=1e-9, max_rank=rank)
    out = []
    for tri_grid in (0, 2):
        ctx = ga.Context(0)
        ctx.set_option(nat.OPT_TRI_GRID, tri_grid)
        model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam, cells=rt)
        algo = ga.IcpRegistration(ctx)
        cfg = ga.IcpConfiguration(maxIterations=30, initialSigma=float(rng.choice([1.0, 20.0])) if tri_grid == 0 else cfg.initialSigma, endSigma=1.0,
                                  correspondenceMethod="TriangularClosestPoint")
        state = algo.createInitialState(model, tv, cfg, targetCells=tt, initial_pose=pose)
        for _ in range(4):
            state = algo.update(state)
        cp, w = algo.surfaceCorrespondence(state)
        out.append((np.array(state.general.fit), cp.copy(), w.copy(), state.general.status))
        algo.close()
        ctx.close()
    same = np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1], equal_nan=True) and \
        np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3]
    print(f"case {case:3d} closed={int(closed)} M={rv.shape[0]:5d} T={tt.shape[0]:5d} accepted {int(out[0][2].sum()):5d} status {out[0][3]} "
          f"{'ok' if same else 'MISMATCH'}", flush=True)
    if not same:
        sys.exit(1)
print("all cases bit-identical")

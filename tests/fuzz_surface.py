#!/usr/bin/env python3
"""One-off randomized sweep of the surface kernels (development aid, not collected by pytest; lives under tests/ because it uses
the oracle): random open / closed meshes of random sizes and poses, closest point on the surface + weights of the three
rejection rules + distance statistics (both directions, boundary-aware) + nearest neighbour + the along-normal flavour's intersections,
HIP path vs the oracle.
    PYTHONPATH=. python tests/fuzz_surface.py [n] [seed]"""
import sys

import numpy as np
import torch  # noqa: F401

import gingr_amd as ga
from oracle import gingr_oracle as go

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = ga.Context(0)
ctx_grid = ga.Context(0)
from gingr_amd import _native as _nat  # noqa: E402
ctx_grid.set_option(_nat.OPT_TRI_GRID, 2)


def grid(n, m, size, amp, closed):
    u, v = np.linspace(0, 1, n), np.linspace(0, 1, m)
    U, V = np.meshgrid(u, v, indexing="ij")
    if closed:                                              # torus-like tube: no boundary
        th, ph = 2 * np.pi * U[:-1, :-1], 2 * np.pi * V[:-1, :-1]
        R, r = size, size * 0.35 * (1 + amp * np.sin(3 * th) * np.cos(2 * ph))
        P = np.stack([(R + r * np.cos(ph)) * np.cos(th), (R + r * np.cos(ph)) * np.sin(th), r * np.sin(ph)], -1)
        a, b = P.shape[:2]
        idx = np.arange(a * b).reshape(a, b)
        i0, i1, j0, j1 = idx, np.roll(idx, -1, 0), idx, np.roll(idx, -1, 1)
        q00, q10, q01, q11 = idx, np.roll(idx, -1, 0), np.roll(idx, -1, 1), np.roll(np.roll(idx, -1, 0), -1, 1)
        tris = np.concatenate([np.stack([q00, q10, q01], -1).reshape(-1, 3), np.stack([q10, q11, q01], -1).reshape(-1, 3)])
        return P.reshape(-1, 3), tris.astype(np.int32)
    X, Y = (U - 0.5) * 2 * size, (V - 0.5) * 2 * size
    Z = amp * size * np.sin(3 * U + 1) * np.cos(2 * V)
    P = np.stack([X, Y, Z], -1)
    idx = np.arange(n * m).reshape(n, m)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    return P.reshape(-1, 3), np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.int32)


worst = worst_along = 0.0
for case in range(n_cases):
    closed = bool(rng.integers(0, 2))
    n, m = int(rng.integers(8, 40)), int(rng.integers(8, 40))
    size, amp = float(rng.uniform(5, 60)), float(rng.uniform(0.0, 0.3))
    v1, t1 = grid(n, m, size, amp, closed)
    v2, t2 = grid(int(rng.integers(8, 40)), int(rng.integers(8, 40)), size * float(rng.uniform(0.9, 1.1)), float(rng.uniform(0.0, 0.3)), closed)
    R = go.euler_to_rot(*rng.normal(0, 0.05, 3))
    v2 = v2 @ R.T + rng.normal(0, 0.02 * size, 3) + rng.normal(0, 1e-3 * size, v2.shape)
    if rng.integers(0, 4) == 0:
        t2 = t2[:, [0, 2, 1]]                               # flipped orientation: the normal rule fires everywhere
    mo = go.build_gaussian_gpmm(v1, size, size / 4, rel_tol=1e-6, max_rank=8)
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam, cells=t1)
    algo = ga.IcpRegistration(ctx)
    state = algo.createInitialState(model, v2, ga.IcpConfiguration(maxIterations=5, initialSigma=2.0, endSigma=1.0, correspondenceMethod="TriangularClosestPoint"), targetCells=t2)
    cp, w = algo.surfaceCorrespondence(state)
    fit = np.asarray(state.general.fit)
    ocp, ow, _ = go.surface_correspondence(fit, t1, v2, t2)
    e = float(np.abs(cp - ocp).max() / size)
    bad_w = int((w != ow).sum())
    s0 = algo.surfaceDistanceStats(state, 0, boundary_aware=True, sdev=1.0)
    o0 = go.surface_distance_stats(fit, v2, t2, True, 1.0)
    s1 = algo.surfaceDistanceStats(state, 1, boundary_aware=True, sdev=2.0)
    o1 = go.surface_distance_stats(v2, fit, t1, True, 2.0)
    idx, d2, _ = ctx.nn(fit, v2)
    oidx, _, _ = go.icp_closest_point(fit, v2)
    ok = e < 1e-10 and bad_w == 0 and s0[2] == o0[2] and s1[2] == o1[2] and abs(s0[0] - o0[0]) <= 1e-9 * max(1, abs(o0[0])) and \
        abs(s1[3] - o1[3]) <= 1e-9 * max(1, abs(o1[3])) and np.array_equal(idx, oidx)
    worst = max(worst, e)
    print(f"case {case:3d} closed={int(closed)} M={v1.shape[0]:5d} T={t1.shape[0]:5d} N={v2.shape[0]:5d} cp err {e:.1e} weights differing {bad_w} "
          f"counted {s0[2]}/{o0[2]} {s1[2]}/{o1[2]} {'ok' if ok else 'MISMATCH'}", flush=True)
    algo.close()
    if not ok:
        sys.exit(1)
    # the along-normal flavour on the same pair (ClosestPointRegistrator.scala:102-131): nearest intersection of the normal line
    ocp = None
    for c_along in (ctx, ctx_grid):   # the tile scan (meshes below the grid's size threshold) and the walk over the triangle grid
        algo = ga.IcpRegistration(c_along)
        state = algo.createInitialState(model, v2, ga.IcpConfiguration(maxIterations=5, initialSigma=2.0, endSigma=1.0, correspondenceMethod="AlongNormalClosestPoint"), targetCells=t2)
        cp, w = algo.surfaceCorrespondence(state)
        if ocp is None:
            ocp, ow, _ = go.along_normal_correspondence(np.asarray(state.general.fit), t1, v2, t2)
        ea, bad_w = float(np.abs(cp - ocp).max() / size), int((w != ow).sum())
        algo.close()
        worst_along = max(worst_along, ea)
        if not (ea < 1e-9 and bad_w == 0):
            print(f"case {case:3d} along the normal (grid walk: {c_along is ctx_grid}): cp err {ea:.1e} weights differing {bad_w} MISMATCH", flush=True)
            sys.exit(1)
print("worst closest-point error / size:", worst, " along the normal:", worst_along)

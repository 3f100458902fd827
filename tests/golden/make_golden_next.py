#!/usr/bin/env python3
"""Generates tests/golden/expected_next.npz: golden vectors for the SURVEY 8f rows (surface correspondence, GPMM construction,
surface distance statistics, classic CPD, Metropolis-Hastings chain).  Run from the repo root:

    python tests/golden/make_golden_next.py

Inputs are the committed fixtures (tests/golden/inputs.npz, femur_mesh.npz -- the reference's own demo data, see make_golden.py and
make_mesh_fixture.py); outputs come from the CPU oracle (oracle/gingr_oracle.py), NOT from the reference, which cannot run here:
they pin the oracle and the HIP path against regressions and against each other ("parity unpinned")."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gingr_oracle as go  # noqa: E402


def grid_mesh(n, size, height, seed):
    rng = np.random.default_rng(seed)
    xs = np.linspace(-size, size, n)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = height * np.sin(X / size * 2.0) * np.cos(Y / size * 1.5) + rng.normal(0, 0.05, X.shape)
    v = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    return v, np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.int32)


def main():
    d = np.load(os.path.join(HERE, "inputs.npz"))
    m = np.load(os.path.join(HERE, "femur_mesh.npz"))
    femur, target = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
    cells, tcells = m["femur_cells"].astype(np.int32), m["femur_target_cells"].astype(np.int32)
    out = {}
    # (1) surface correspondence of the posed femur against the target (ClosestPointRegistrator.scala:75-100)
    R = go.euler_to_rot(0.02, -0.03, 0.01)
    posed = femur @ R.T + np.array([1.0, -2.0, 0.5])
    cp, w, dist = go.surface_correspondence(posed, cells, target, tcells)
    out.update(surf_pose=np.array([0.02, -0.03, 0.01, 1.0, -2.0, 0.5]), surf_cp=cp, surf_w=w, surf_mean_distance=np.array(dist))
    # (2) surface distance statistics, both directions, sdev 5 (IndependentPointDistanceEvaluator.scala:54-70)
    out["stats_m2t"] = np.array(go.surface_distance_stats(posed, target, tcells, False, 5.0))
    out["stats_t2m"] = np.array(go.surface_distance_stats(target, posed, cells, False, 5.0))
    # (3) GPMM over every third femur vertex: Gaussian(60, 30), tolerance 0.01 (GPMMHelper.scala:96-101)
    sub = femur[::3]
    mo = go.build_gpmm_mixture(sub, [60.0], [30.0], 0.01)
    out.update(gpmm_rank=np.array(mo.rank), gpmm_variance=mo.lam)
    # (4) classic CPD, three iterations each from the initial state, 150 points
    rng = np.random.default_rng(3)
    Y = rng.normal(0, 10, (150, 3))
    X = 1.1 * Y[rng.permutation(150)] @ go.euler_to_rot(0.1, -0.2, 0.15).T + np.array([1.0, 2.0, -1.0]) + rng.normal(0, 0.05, (150, 3))
    out.update(ccpd_Y=Y, ccpd_X=X)
    G = go.cpd_g_block(Y, Y, 8.0)
    for kind in ("rigid", "affine", "nonrigid"):
        TY, s2 = Y, go.classic_cpd_initial_sigma2(Y, X)
        for _ in range(3):
            P = go.classic_cpd_expectation(X, TY, s2, 0.05)
            if kind == "rigid":
                TY, s2, _ = go.classic_cpd_maximization_rigid(X, TY, P)
            elif kind == "affine":
                TY, s2, _ = go.classic_cpd_maximization_affine(X, TY, P)
            else:
                TY, s2, _ = go.classic_cpd_maximization_nonrigid(X, TY, P, s2, G, 2.0)
        out[f"ccpd_{kind}_TY"], out[f"ccpd_{kind}_sigma2"] = TY, np.array(s2)
    # (5) Metropolis-Hastings chain (GingrAlgorithm.scala:115-175): CPD informed proposals + stock random walks, 26 states, seed 42
    v1, t1 = grid_mesh(16, 20.0, 3.0, 7)
    v2, t2 = grid_mesh(18, 22.0, 3.5, 8)
    v2 = v2 @ go.euler_to_rot(0.02, -0.015, 0.03).T + np.array([0.4, -0.3, 0.6])
    mo = go.build_gaussian_gpmm(v1, 25.0, 4.0, rel_tol=1e-9, max_rank=12)
    st0 = go.initial_state(mo, go.cpd_initial_sigma2(mo.ref + mo.mean, v2))
    upd = lambda st, z: go.cpd_update(mo, v2, st, w=0.05, z=z)

    def logq(f, t):
        try:
            pids, pts, var = go.cpd_observations(mo, v2, f, 0.05)
            return go.posterior_logpdf_of_mesh(mo, f, pids, pts, var, f.fit)
        except np.linalg.LinAlgError:
            return -np.inf
    logv = lambda st: go.model_evaluator_logvalue(st.alpha) + go.independent_point_distance_logvalue(st.fit, t1, v2, t2, 1.0)
    best, states, flags = go.mh_run(mo, st0, 26, upd, logq, logv, 0.5, go.ChainRandom(42))
    out.update(chain_accept=np.array(flags), chain_alpha=np.stack([s.alpha for s in states]), chain_best_alpha=best.alpha,
               chain_sigma2=np.array([s.sigma2 for s in states]))
    np.savez_compressed(os.path.join(HERE, "expected_next.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()

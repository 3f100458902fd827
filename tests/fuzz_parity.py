#!/usr/bin/env python3
"""One-off randomized parity sweep (development aid, not part of the test suite): random sizes / ranks / transforms / outlier
weights / landmarks, three CPD or ICP updates each, HIP path vs the oracle.   PYTHONPATH=. python tests/fuzz_parity.py [n] [seed]  (lives under tests/: it uses the oracle)"""
import sys

import numpy as np
import torch  # noqa: F401

import gingr_amd as ga
from oracle import gingr_oracle as go

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = ga.Context(0)
worst = 0.0
for case in range(n_cases):
    M = int(rng.integers(5, 700))
    N = int(rng.integers(5, 700))
    rank = int(rng.integers(1, min(3 * M, 60)))
    transform = int(rng.integers(0, 3))
    step = float(rng.choice([1.0, 1.0, 0.5, 0.8]))
    w = float(rng.choice([0.0, 0.05, 0.3]))
    ref = rng.normal(0, 40, (M, 3))
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, rank)))
    lam = np.sort(rng.uniform(0.5, 200.0, rank))[::-1].copy()
    mo = go.PDM(ref=ref, mean=rng.normal(0, 0.3, (M, 3)), U=np.ascontiguousarray(U), lam=lam)
    base = mo.instance(rng.normal(0, 1, rank)) @ go.euler_to_rot(*rng.normal(0, 0.1, 3)).T + rng.normal(0, 2, 3)
    target = base[rng.integers(0, M, N)] + rng.normal(0, 0.5, (N, 3))
    n_lm = int(rng.integers(0, 4))
    lms = lmo = None
    if n_lm:
        pids = rng.choice(M, n_lm, replace=False).astype(np.int32)
        pts = target[rng.integers(0, N, n_lm)]
        covs = np.stack([np.eye(3) * float(rng.uniform(0.5, 4.0)) for _ in range(n_lm)])
        lms = ga.LandmarkCorrespondences(pids, pts, covs)
        lmo = go.Landmarks(pids=pids, points=pts, covs=covs)
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
    icp = bool(rng.integers(0, 2))
    if icp:
        algo, cfg = ga.IcpRegistration(ctx), ga.IcpConfiguration(maxIterations=20, initialSigma=30.0, endSigma=2.0, correspondenceMethod="PointcloudClosestPoint")
    else:
        algo, cfg = ga.CpdRegistration(ctx), ga.CpdConfiguration(maxIterations=20, w=w)
    state = algo.createInitialState(model, target, cfg, transform=transform, stepLength=step, landmarks=lms)
    st = go.initial_state(mo, state.general.sigma2, global_transformation=transform, step_length=step)
    ok = True
    for it in range(3):
        state = algo.update(state)
        if icp:
            st, _ = go.icp_update(mo, target, st, 30.0, 2.0, 20, lmo)
        else:
            st = go.cpd_update(mo, target, st, w=w, landmarks=lmo)
        if state.general.status != st.status:
            ok = False
            break
        if st.status == 0:
            err = float(np.linalg.norm(state.general.fit - st.fit) / max(np.linalg.norm(st.fit), 1e-300))
            worst = max(worst, err)
            if err > 1e-5:
                ok = False
                break
    algo.close()
    print(f"case {case:3d} {'ICP' if icp else 'CPD'} M={M:4d} N={N:4d} r={rank:3d} T={transform} step={step} w={w} lm={n_lm} "
          f"status={st.status} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
print("worst relative error on the fit:", worst)

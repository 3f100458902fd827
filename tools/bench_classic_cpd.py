#!/usr/bin/env python3
"""Iteration time of the classic CPD family on the device (G/other/algorithms/cpd): femur pair (1 622 vertices) and synthetic
clouds, rigid / affine / non-rigid.  For the non-rigid kind the host-LAPACK time of the same M x M solve (numpy.linalg.solve on
this box's cores -- the reference's `A \\ B` is the same LAPACK call through Breeze) is printed beside it.
    PYTHONPATH=. python tools/bench_classic_cpd.py [M ...]"""
import json
import os
import sys
import time

import numpy as np
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch  # noqa: F401  (first: one HIP runtime per process)

import gingr_amd as ga
from gingr_amd import classic as cl

ctx = ga.Context(0)
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
d = np.load(os.path.join(root, "inputs.npz"))
cases = [("femur", d["femur"].astype(np.float64), d["femur_target"].astype(np.float64), 30.0)]
for M in [int(a) for a in sys.argv[1:]] or [5000]:
    rng = np.random.default_rng(M)
    Y = rng.normal(0, 50, (M, 3))
    X = Y[rng.permutation(M)] @ np.array([[0.995, -0.0998, 0], [0.0998, 0.995, 0], [0, 0, 1.0]]) + rng.normal(0, 2, (M, 3)) + 1.0
    cases.append((f"synthetic {M}", Y, X, 15.0))
out = []
for name, Y, X, beta in cases:
    f = cl.CPDFactory(ctx, Y, lambda_=2.0, beta=beta, w=0.0)
    row = {"case": name, "M": int(Y.shape[0]), "N": int(X.shape[0])}
    for kind, mk in (("rigid", f.registerRigidly), ("affine", f.registerAffine), ("nonrigid", f.registerNonRigidly)):
        reg = mk(X)
        reg.Iteration()
        ctx.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            reg.cpd.ctx._lib.gingr_classic_cpd_iterate(reg._h, 1)
        ctx.synchronize()
        row[f"{kind}_ms_per_iteration"] = (time.perf_counter() - t0) / n * 1e3
        row[f"{kind}_sigma2"] = reg.sigma2()
        reg.close()
    M = Y.shape[0]
    A = np.exp(-((Y[:, None, :] - Y[None, :, :]) ** 2).sum(-1) / (2 * beta * beta)) + np.eye(M) if M <= 6000 else None
    if A is not None:
        B = np.random.default_rng(0).normal(0, 1, (M, 3))
        t0 = time.perf_counter()
        np.linalg.solve(A, B)
        row["host_lapack_solve_ms"] = (time.perf_counter() - t0) * 1e3
        row["host_cores"] = os.cpu_count()
    out.append(row)
print(json.dumps(out))

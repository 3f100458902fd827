#!/usr/bin/env python3
"""Stateless operators at sizes far beyond the benchmark (default 1 000 000 x 1 000 000 points): index arithmetic, workspace sizes
and chunk plans must hold; results are checked on sampled rows / columns against float64 numpy.
    python tools/large_size_check.py [points=1000000]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import gingr_amd.api as ga  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(3)
tgt = (rng.normal(0.0, 50.0, (n, 3))).astype(np.float32).astype(np.float64)
fit = tgt[rng.permutation(n)] + rng.normal(0.0, 0.5, (n, 3))
ctx = ga.Context(0)
out = {"points": n}

t0 = time.perf_counter()
idx, d2, mean = ctx.nn(fit, tgt)
out["nn_s"] = time.perf_counter() - t0
rows = rng.choice(n, 24, replace=False)
for r in rows:
    d = np.sum((tgt - fit[r]) ** 2, axis=1)
    j = int(np.argmin(d))
    assert j == idx[r] or d[j] == d[idx[r]], (r, j, idx[r])
out["nn_rows_checked"] = len(rows)

sigma2 = 1.0   # late regime: the exact-zero culling leaves a few tiles per block
t0 = time.perf_counter()
st = ctx.cpd_stats(fit, tgt, sigma2, 0.1)
out["cpd_stats_s"] = time.perf_counter() - t0
M = N = n
c = 0.1 / 0.9 * (2 * np.pi * sigma2) ** 1.5 * (M / N)
cols = rng.choice(n, 12, replace=False)
for j in cols:
    k = np.exp(-np.sum((fit - tgt[j]) ** 2, axis=1) / (2 * sigma2))
    den = k.sum() + c
    assert abs(st["den"][j] - den) <= 1e-10 * den, (j, st["den"][j], den)
den = np.asarray(st["den"])
for r in rows[:12]:
    k = np.exp(-np.sum((tgt - fit[r]) ** 2, axis=1) / (2 * sigma2)) / den
    p1 = k.sum()
    assert abs(st["P1"][r] - p1) <= 1e-9 * max(p1, 1e-300), (r, st["P1"][r], p1)
    px = k @ tgt
    assert np.allclose(np.asarray(st["PX"]).reshape(-1, 3)[r], px, rtol=1e-9, atol=1e-9 * np.abs(px).max())
out["cpd_rows_cols_checked"] = [int(len(rows[:12])), int(len(cols))]
out["sigma2_next"] = float(st["sigma2_next"])
print(json.dumps(out))

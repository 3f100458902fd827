#!/usr/bin/env python3
"""One-off randomized sweep of the two CPD all-pairs passes against the strict C oracle (development aid, not collected by pytest):
odd sizes across chunk / tile / launch-round boundaries, sigma2 from the dense regime down to heavy exact-zero culling, clustered
and uniform clouds, outlier weights.   python tests/fuzz_cpd_stats.py [n] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import gingr_amd as ga  # noqa: E402
from oracle import c_oracle as co  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = ga.Context(0)
worst = {"den": 0.0, "P1": 0.0, "PX": 0.0, "s2": 0.0}
sizes = [1, 2, 63, 64, 65, 255, 256, 257, 511, 513, 1023, 1025, 2047, 2049, 4095, 4097, 6250, 12500, 20011]
for case in range(n_cases):
    M = int(rng.choice(sizes)) if rng.random() < 0.7 else int(rng.integers(1, 20000))
    N = int(rng.choice(sizes)) if rng.random() < 0.7 else int(rng.integers(1, 20000))
    spread = float(rng.choice([1.0, 50.0, 1000.0]))
    x = rng.normal(0, spread, (N, 3))
    if rng.random() < 0.3:                                   # clustered: several far-apart blobs (culling with uneven costs)
        x += rng.integers(-3, 4, (N, 3)) * spread * 20
    y = x[rng.integers(0, N, M)] + rng.normal(0, spread * 0.05, (M, 3))
    s2 = float(spread * spread * 10.0 ** rng.uniform(-5, 2))
    w = float(rng.choice([0.0, 0.1, 0.5]))
    got = ctx.cpd_stats(y, x, s2, w)
    want = co.cpd_stats(y, x, s2, w)
    with np.errstate(all="ignore"):
        fin = np.isfinite(want.den) & (want.den > 1e-290)
        assert np.array_equal(np.isnan(got["P1"]), np.isnan(want.P1)), "NaN pattern of P1"
        e_den = float(np.max(np.abs(got["den"][fin] - want.den[fin]) / want.den[fin])) if fin.any() else 0.0
        ok = np.isfinite(want.P1) & (want.P1 > 1e-290)
        e_p1 = float(np.max(np.abs(got["P1"][ok] - want.P1[ok]) / want.P1[ok])) if ok.any() else 0.0
        scale = np.maximum(np.abs(want.PX[ok]).max(initial=0.0), 1e-300)
        e_px = float(np.max(np.abs(got["PX"][ok] - want.PX[ok])) / scale) if ok.any() else 0.0
        e_s2 = abs(got["sigma2_next"] - want.sigma2_next) / abs(want.sigma2_next) if np.isfinite(want.sigma2_next) and want.sigma2_next != 0 else 0.0
    for k, v in (("den", e_den), ("P1", e_p1), ("PX", e_px), ("s2", e_s2)):
        worst[k] = max(worst[k], v)
    good = e_den < 1e-10 and e_p1 < 1e-9 and e_px < 1e-9 and e_s2 < 1e-6
    print(f"case {case:3d} M={M:6d} N={N:6d} spread={spread:7.1f} sigma2={s2:11.4g} w={w} den {e_den:.1e} P1 {e_p1:.1e} PX {e_px:.1e} "
          f"s2' {e_s2:.1e} nan_rows={int(np.isnan(want.P1).sum())} {'ok' if good else 'MISMATCH'}", flush=True)
    if not good:
        sys.exit(1)
print("worst relative errors:", worst)

"""Run-to-run bit stability of the posterior mean (weighted Gram + right-hand side + solve + posed instance) at ranks on every solve
path: 2 000 calls per rank on the same inputs must give the same bits (a race inside a one-workgroup kernel would not)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gingr_amd as ga
ctx = ga.Context(0)
rng = np.random.default_rng(3)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for rank in (100, 130, 150, 200, 240, 256, 300):
    M = 700
    ref = rng.normal(0, 30, (M, 3))
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, rank)))
    lam = np.sort(rng.uniform(0.5, 400.0, rank))[::-1].copy()
    dm = ga.DeviceModel(ctx, ga.PointDistributionModel(ref, np.zeros((M, 3)), U, lam))
    obs = ref + rng.normal(0, 1.0, (M, 3))
    w = 10.0 ** rng.uniform(-3, 1, M)
    w[rng.random(M) < 0.3] = 0.0
    first = None
    bad = 0
    for k in range(reps):
        mean, a = dm.posterior_mean(obs, w)
        if first is None:
            first = (mean.copy(), a.copy())
        elif not (np.array_equal(mean, first[0]) and np.array_equal(a, first[1])):
            bad += 1
            if bad <= 3:
                print("  rank", rank, "call", k, "differs: max |da|", np.abs(a - first[1]).max(), "max |a|", np.abs(first[1]).max())
    print("rank", rank, "calls", reps, "differing", bad, flush=True)
    dm.close()

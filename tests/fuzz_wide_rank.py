"""Random sizes and ranks against numpy for the wide-rank paths (gp_wide.hip, the super-panel solve): not collected by pytest
(`python tests/fuzz_wide_rank.py [cases] [seed]` on a GPU box).  Each case: a random model (M points, rank r in 113 .. 512, 3 M >= r),
random weights over a wide range with zeros, a pose; the stateless posterior mean (weighted Gram + right-hand side + solve + posed
instance) against the normal equations solved in numpy; then two fused CPD updates against the oracle for every fifth case."""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gingr_amd as ga  # noqa: E402
from oracle import gingr_oracle as go  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = ga.Context(0)
    worst = 0.0
    for c in range(cases):
        rank = int(rng.integers(113, 513))
        M = int(rng.integers((rank + 2) // 3 + 1, 2500))
        ref = rng.normal(0, 30, (M, 3))
        U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, rank)))
        lam = np.sort(rng.uniform(0.5, 400.0, rank))[::-1].copy()
        mean = rng.normal(0, 0.1, (M, 3))
        dm = ga.DeviceModel(ctx, ga.PointDistributionModel(ref, mean, U, lam))
        euler, t = tuple(rng.normal(0, 0.05, 3)), rng.normal(0, 2.0, 3)
        R = go.euler_to_rot(*euler)
        obs = (ref + mean) @ R.T + t + rng.normal(0, 1.0, (M, 3))
        w = 10.0 ** rng.uniform(-3, 1, M)
        w[rng.random(M) < rng.uniform(0, 0.5)] = 0.0
        got_mean, got_a = dm.posterior_mean(obs, w, euler=euler, translation=tuple(t))
        Q = U * np.sqrt(lam)[None, :]
        e = ((obs - t) @ R - ref - mean).reshape(-1)          # model-frame residual
        W3 = np.repeat(w, 3)
        A = np.eye(rank) + (Q * W3[:, None]).T @ Q
        a = np.linalg.solve(A, Q.T @ (W3 * e))
        want_mean = (ref + mean + (Q @ a).reshape(M, 3)) @ R.T + t
        ea, em = rel(got_a, a), rel(got_mean, want_mean)
        worst = max(worst, ea)
        assert ea < 1e-7 and em < 1e-9, (c, M, rank, ea, em)
        dm.close()
        if c % 5 == 0:
            N = int(rng.integers(50, 1500))
            mo = go.PDM(ref=ref, mean=mean, U=U, lam=lam)
            target = rng.normal(0, 30, (N, 3))
            algo = ga.CpdRegistration(ctx)
            state = algo.createInitialState(ga.PointDistributionModel(ref, mean, U, lam), target, ga.CpdConfiguration(maxIterations=10, w=0.2, initialSigma=400.0))
            st = go.initial_state(mo, 400.0)
            for _ in range(2):
                state = algo.update(state)
                st = go.cpd_update(mo, target, st, w=0.2)
                assert state.general.status == st.status == 0 and rel(state.general.fit, st.fit) < 1e-5, (c, M, N, rank)
            algo.close()
        if c % 10 == 9:
            print(f"{c + 1} cases, worst coefficient error {worst:.2e}", flush=True)
    print(f"ok: {cases} cases, worst coefficient error {worst:.2e}")


if __name__ == "__main__":
    main()

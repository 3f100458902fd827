"""Random models through the set-up path (round 6: eig.hip, dense_spd_inverse, the moment blocks as one symmetric product, the r x r
products on the matrix pipe, the threaded k-d order): not collected by pytest (`python tests/fuzz_model_setup.py [cases] [seed]` on a
GPU box).  Each case: a random cloud (40 .. 1 500 points), a random kernel (one Gaussian, a mixture of two, or the mirrored Gaussian
= two kernels), a random tolerance (rank capped at the 512 of the device model, as in the library) or rank limit -- the model
built on the device against the oracle's restatement of scalismo's route (rank, eigenvalues, orthonormal basis, covariance); then, for every third case, one CPD update and one point-cloud
ICP update (the posterior through the moment eigenbasis up to rank 192) against the oracle with the ORACLE's model uploaded, which
exercises Binv, the moment blocks and the constant products whatever the device build's eigenvector signs are."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gingr_amd as ga  # noqa: E402
from oracle import gingr_oracle as go  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def check_model(host, mo, case):
    assert host.variance.shape[0] == mo.rank, (case, host.variance.shape[0], mo.rank)
    e_lam = rel(host.variance, mo.lam)
    U = np.asarray(host.basis)
    e_orth = float(np.abs(U.T @ U - np.eye(U.shape[1])).max())
    rows = np.random.default_rng(0).permutation(U.shape[0])[:240]
    Cd = (U[rows] * host.variance[None, :]) @ U[rows].T
    Co = (mo.U[rows] * mo.lam[None, :]) @ mo.U[rows].T
    e_cov = float(np.abs(Cd - Co).max() / np.abs(Co).max())
    # U = L V / sqrt(lambda): the orthonormality of the columns with the smallest eigenvalues carries eps * lambda_max / lambda_min
    # whatever computes V (rank close to the number of degrees of freedom of a smooth kernel: ratios of 1e5 .. 1e7)
    tol_orth = max(1e-9, 1e-14 * float(mo.lam[0] / max(mo.lam[-1], 1e-300)))
    assert e_lam < 1e-9 and e_orth < tol_orth and e_cov < 1e-9, (case, e_lam, e_orth, tol_orth, e_cov)
    return max(e_lam, e_cov)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = ga.Context(0)
    worst = 0.0
    for c in range(cases):
        M = int(rng.integers(40, 1500))
        ref = rng.normal(0, rng.uniform(10, 60), (M, 3))
        sigma, scaling = float(rng.uniform(8, 70)), float(rng.uniform(1, 40))
        kind = int(rng.integers(0, 3))
        if rng.random() < 0.5:
            tol, max_rank = float(10.0 ** rng.uniform(-4, -1)), 0
        else:
            tol, max_rank = 0.0, int(rng.integers(1, min(512, 3 * M) + 1))
        g = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=tol, maxRank=max_rank)
        if kind == 0:
            mo = go.build_gpmm_mixture(ref, [sigma], [scaling], tol, max_rank or 512)
            dm = g.Gaussian(sigma, scaling)
        elif kind == 1:
            s2, c2 = sigma * float(rng.uniform(0.3, 0.8)), scaling * float(rng.uniform(0.2, 2.0))
            mo = go.build_gpmm_mixture(ref, [sigma, s2], [scaling, c2], tol, max_rank or 512)
            dm = g.GaussianMixture([ga.GaussianKernelParameters(sigma, scaling), ga.GaussianKernelParameters(s2, c2)])
        else:
            mo = go.build_gpmm_diagonal(ref, go.symmetric_gauss_kernel_fun(ref, sigma, scaling), tol, max_rank or 512)
            dm = g.GaussianSymmetry(sigma, scaling)
        case = (c, M, kind, round(sigma, 2), round(scaling, 2), tol, max_rank, mo.rank)
        assert dm.rank == mo.rank, case
        worst = max(worst, check_model(dm.to_host(), mo, case))
        dm.device().close()
        if c % 3 == 0 and mo.rank >= 2:
            N = int(rng.integers(50, 1200))
            target = mo.instance(rng.normal(0, 0.7, mo.rank))[rng.permutation(M)[: min(M, N)]] + rng.normal(0, 0.5, (min(M, N), 3))
            model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
            algo = ga.CpdRegistration(ctx)
            cfg = ga.CpdConfiguration(maxIterations=10, w=0.1)
            state = algo.createInitialState(model, target, cfg)
            st = go.initial_state(mo, state.general.sigma2)
            state = algo.update(state)
            st = go.cpd_update(mo, target, st, w=0.1)
            assert state.general.status == st.status == 0 and rel(state.general.fit, st.fit) < 1e-6, (case, "cpd", rel(state.general.fit, st.fit))
            algo.close()
            algo = ga.IcpRegistration(ctx)
            cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=10.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
            state = algo.createInitialState(model, target, cfg)
            st = go.initial_state(mo, state.general.sigma2)
            state = algo.update(state)
            st, _ = go.icp_update(mo, target, st, 10.0, 1.0, 10)
            assert state.general.status == st.status == 0 and rel(state.general.fit, st.fit) < 1e-6, (case, "icp", rel(state.general.fit, st.fit))
            algo.close()
        if c % 10 == 9:
            print(f"case {c + 1}/{cases}: worst relative error so far {worst:.2e}", flush=True)
    print(f"{cases} cases, worst relative error {worst:.3e}")
    ctx.close()


if __name__ == "__main__":
    main()

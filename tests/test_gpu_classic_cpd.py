"""GPU tests of the classic CPD family (G/other/algorithms/cpd/*.scala) against the oracle's dense
restatement: single Iterations from the same state, whole Registrations, the non-rigid M x M solve across panel boundaries."""
import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def pair(M, N, seed, noise=0.05, scale=1.1):
    rng = np.random.default_rng(seed)
    Y = rng.normal(0, 10, (M, 3))
    R = go.euler_to_rot(0.1, -0.2, 0.15)
    X = scale * Y[rng.permutation(M)[:N] if N <= M else rng.integers(0, M, N)] @ R.T + np.array([1.0, 2.0, -1.0]) + rng.normal(0, noise, (N, 3))
    return Y, X


@pytest.mark.parametrize("M,N,w", [(60, 60, 0.0), (257, 300, 0.1), (1000, 700, 0.05)])
def test_rigid_and_affine_iterations(ctx, M, N, w):
    from gingr_amd import classic as cl
    Y, X = pair(M, N, M)
    s0 = go.classic_cpd_initial_sigma2(Y, X)
    f = cl.CPDFactory(ctx, Y, w=w)
    for reg, omax in ((f.registerRigidly(X), go.classic_cpd_maximization_rigid), (f.registerAffine(X), go.classic_cpd_maximization_affine)):
        assert abs(reg.sigma2() - s0) < 1e-12 * s0
        TY, s2 = Y, s0
        for it in range(3):                                   # every iteration from the oracle's state
            P = go.classic_cpd_expectation(X, TY, s2, w)
            oTY, os2, otr = omax(X, TY, P)
            gTY, gs2 = reg.Iteration(TY, s2)
            assert rel(gTY, oTY) < 1e-10, (it, rel(gTY, oTY))
            assert abs(gs2 - os2) < 1e-9 * abs(os2) + 1e-12
            s, L, t = reg.transform()
            if len(otr) == 3:
                assert abs(s - otr[0]) < 1e-10 and np.abs(L - otr[1]).max() < 1e-10 and np.abs(t - otr[2]).max() < 1e-8
            else:
                assert np.abs(L - otr[0]).max() < 1e-9 and np.abs(t - otr[1]).max() < 1e-8
            TY, s2 = oTY, os2
        reg.close()


@pytest.mark.parametrize("M,N,beta,lam", [(64, 64, 8.0, 2.0), (130, 150, 6.0, 2.0), (500, 420, 5.0, 3.0)])
def test_nonrigid_iterations(ctx, M, N, beta, lam):
    """M = 64 is exactly one panel, 130 crosses two panel boundaries with a ragged tail, 500 has 8 panels."""
    from gingr_amd import classic as cl
    Y, X = pair(M, N, 7 + M, noise=0.3)
    G = go.cpd_g_block(Y, Y, beta)
    reg = cl.CPDFactory(ctx, Y, lambda_=lam, beta=beta, w=0.05).registerNonRigidly(X)
    TY, s2 = Y, go.classic_cpd_initial_sigma2(Y, X)
    for it in range(3):
        P = go.classic_cpd_expectation(X, TY, s2, 0.05)
        oTY, os2, oW = go.classic_cpd_maximization_nonrigid(X, TY, P, s2, G, lam)
        gTY, gs2 = reg.Iteration(TY, s2)
        assert rel(reg.W(), oW) < 1e-7, (it, rel(reg.W(), oW))
        assert rel(gTY, oTY) < 1e-9, (it, rel(gTY, oTY))
        assert abs(gs2 - os2) < 1e-8 * abs(os2)
        TY, s2 = oTY, os2
    reg.close()


@pytest.mark.parametrize("kind", ["rigid", "affine", "nonrigid"])
def test_registration_loop(ctx, kind):
    from gingr_amd import classic as cl
    Y, X = pair(150, 150, 3)
    oTY, os2, oit, oconv = go.classic_cpd_registration(Y, X, kind, beta=8.0, max_iteration=60)
    f = cl.CPDFactory(ctx, Y, beta=8.0)
    reg = {"rigid": f.registerRigidly, "affine": f.registerAffine, "nonrigid": f.registerNonRigidly}[kind](X)
    gTY = reg.Registration(60)
    assert (reg.iterations, reg.converged) == (oit, oconv)
    assert rel(gTY, oTY) < 1e-6, rel(gTY, oTY)
    # the reference's Registration always restarts from the template and the initial variance (RigidCPD.scala:59-62): a second call,
    # or one after an Iteration has moved the device state, gives the same result
    reg.Iteration()
    again = reg.Registration(60)
    assert (reg.iterations, reg.converged) == (oit, oconv) and np.array_equal(again, gTY)
    reg.close()
    wrapper = {"rigid": cl.RigidCPDRegistration, "affine": cl.AffineCPDRegistration, "nonrigid": cl.NonRigidCPDRegistration}[kind]
    assert rel(wrapper(ctx, Y, beta=8.0, max_iterations=60).register(X), oTY) < 1e-6


def test_classic_cpd_errors(ctx):
    import gingr_amd as ga
    from gingr_amd import classic as cl
    Y, X = pair(40, 40, 1)
    with pytest.raises(ValueError):
        cl.CPDFactory(ctx, Y, beta=0.0)
    with pytest.raises(ga.GingrNativeError):
        cl.CPDFactory(ctx, Y, w=1.0).registerRigidly(X)
    # duplicated template points make G singular, lambda sigma2 / P1 keeps the system positive definite
    Yd = np.concatenate([Y, Y[:5]])
    reg = cl.CPDFactory(ctx, Yd, beta=5.0).registerNonRigidly(X)
    TY, s2 = reg.Iteration()
    assert np.all(np.isfinite(TY)) and np.isfinite(s2)
    reg.close()


def test_nonrigid_system_residual_at_2000_points(ctx):
    """32 panels of the blocked Cholesky: W must satisfy (G + lambda sigma2 diag(1/P1)) W = diag(1/P1) P X - Y, and TY = Y + G W,
    with P1 and P X taken from the stateless statistics call (no M x N matrix on the host either)."""
    from gingr_amd import classic as cl
    M = N = 2000
    Y, X = pair(M, N, 11, noise=0.4)
    beta, lam, w = 4.0, 2.0, 0.1
    reg = cl.CPDFactory(ctx, Y, lambda_=lam, beta=beta, w=w).registerNonRigidly(X)
    s2 = reg.sigma2()
    st = ctx.cpd_stats(Y, X, s2, w)
    TY, s2n = reg.Iteration()
    W = reg.W()
    d = Y[:, None, :] - Y[None, :, :]
    G = np.exp(-(d * d).sum(-1) / (2 * beta * beta))
    P1, PX = st["P1"], st["PX"]
    rhs = PX / P1[:, None] - Y
    lhs = G @ W + lam * s2 * W / P1[:, None]
    assert rel(lhs, rhs) < 1e-9, rel(lhs, rhs)
    assert rel(TY, Y + G @ W) < 1e-12
    xPx = float(st["Pt1"] @ (X * X).sum(1))
    want = (xPx - 2 * float((TY * PX).sum()) + float(P1 @ (TY * TY).sum(1))) / (P1.sum() * 3)
    assert abs(s2n - want) < 1e-9 * abs(want)
    reg.close()

"""CPU tests pinning the oracle's restatement of the classic CPD family (G/other/algorithms/cpd) to closed forms: with a sharp,
one-to-one soft assignment the rigid / affine Maximizations must return the generating transform; the non-rigid Maximization
must satisfy its own linear system; the Registration loop must stop on the reference's tolerance rule."""
import numpy as np

from oracle import gingr_oracle as go


def data(M=40, seed=0):
    rng = np.random.default_rng(seed)
    Y = rng.normal(0, 10, (M, 3))
    R = go.euler_to_rot(0.3, -0.2, 0.1)
    return Y, R, np.array([1.0, -2.0, 0.5])


def test_rigid_maximization_recovers_a_similarity_transform():
    Y, R, t = data()
    X = 1.3 * Y @ R.T + t
    P = np.eye(Y.shape[0])                                  # perfect one-to-one assignment
    TY, s2, (s, Rr, tt) = go.classic_cpd_maximization_rigid(X, Y, P)
    assert abs(s - 1.3) < 1e-12 and np.abs(Rr - R).max() < 1e-12 and np.abs(tt - t).max() < 1e-11
    assert np.abs(TY - X).max() < 1e-11 and abs(s2) < 1e-10
    # a reflection in the data must not produce a reflection in R (C = diag(1, 1, det(U V^T)), RigidCPD.scala:125-127)
    Xm = X * np.array([1.0, 1.0, -1.0])
    _, _, (_, Rm, _) = go.classic_cpd_maximization_rigid(Xm, Y, P)
    assert abs(np.linalg.det(Rm) - 1.0) < 1e-12


def test_affine_maximization_recovers_an_affine_map():
    Y, R, t = data(seed=1)
    B = R @ np.diag([1.2, 0.8, 1.1]) + 0.05
    X = Y @ B.T + t
    TY, s2, (Bb, tt) = go.classic_cpd_maximization_affine(X, Y, np.eye(Y.shape[0]))
    assert np.abs(Bb - B).max() < 1e-11 and np.abs(tt - t).max() < 1e-10 and np.abs(TY - X).max() < 1e-10 and abs(s2) < 1e-9


def test_nonrigid_maximization_solves_its_system():
    Y, R, t = data(M=60, seed=2)
    X = Y + np.random.default_rng(3).normal(0, 0.5, Y.shape)
    s2, lam, beta = 2.0, 2.0, 6.0
    P = go.classic_cpd_expectation(X, Y, s2, 0.1)
    G = go.cpd_g_block(Y, Y, beta)
    TY, ns2, W = go.classic_cpd_maximization_nonrigid(X, Y, P, s2, G, lam)
    P1 = P.sum(1)
    lhs = G @ W + lam * s2 * W / P1[:, None]
    assert np.abs(lhs - ((P @ X) / P1[:, None] - Y)).max() < 1e-9
    assert np.abs(TY - (Y + G @ W)).max() < 1e-12
    assert np.abs(G - G.T).max() == 0.0 and G[3, 3] == 1.0 and abs(G[0, 1] - np.exp(-((Y[0] - Y[1]) ** 2).sum() / (2 * beta ** 2))) < 1e-15


def test_registration_loop_counts_only_non_converged_iterations():
    Y, R, t = data(M=50, seed=4)
    X = Y @ R.T + t + np.random.default_rng(5).normal(0, 0.05, Y.shape)
    assert abs(go.classic_cpd_initial_sigma2(Y, X) - ((Y[:, None] - X[None]) ** 2).sum() / (3 * 50 * 50)) < 1e-9
    TY, s2, it, conv = go.classic_cpd_registration(Y, X, "rigid", max_iteration=100)
    assert conv and 0 < it < 100 and np.abs(TY - X).max() < 0.5
    _, _, it2, conv2 = go.classic_cpd_registration(Y, X, "rigid", max_iteration=3)
    assert (it2, conv2) == (3, False)

"""CPU tests of the oracle itself (no GPU): closed-form known answers, numpy-vs-C agreement, golden fixtures.

The reference has no tests or golden vectors for this path ("parity unpinned"), so the oracle is anchored on
 (1) closed forms that follow from the reference's formulas, (2) two independent restatements (dense numpy, streaming C)
agreeing to ~1e-13, (3) the committed fixtures produced from the reference's own demo data files.
"""
import math
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import gingr_oracle as go

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def clouds(M, N, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 50, (N, 3))
    y = x[rng.permutation(N)[:M]] + rng.normal(0, 2, (M, 3)) if M <= N else rng.normal(0, 50, (M, 3))
    return y, x


# ---------------------------------------------------------------------------------------- CPD closed forms
def test_w0_columns_sum_to_one_and_Np_is_N():
    y, x = clouds(120, 90, 1)
    st = go.cpd_stats_dense(y, x, 300.0, 0.0)
    assert np.allclose(st.Pt1, 1.0, atol=1e-13)
    assert abs(st.Np - 90) < 1e-10


def test_single_pair():
    y, x = np.array([[1.0, 2.0, 3.0]]), np.array([[1.5, 2.0, 3.0]])
    st = go.cpd_stats_dense(y, x, 2.0, 0.0)
    assert st.den[0] == math.exp(-0.25 / 4.0)
    assert st.P1[0] == 1.0 and np.allclose(st.PX[0], x[0])
    assert abs(st.sigma2_next - 0.25 / 3.0) < 1e-15


def test_sigma_to_infinity_gives_uniform_assignment():
    y, x = clouds(40, 50, 2)
    P = go.cpd_P(y, x, 1e14, 0.0)
    assert np.allclose(P, 1.0 / 40, rtol=1e-9)
    P1, yhat = go.cpd_correspondence(P, y, x)
    assert np.allclose(yhat, x.mean(0)[None, :], atol=1e-6)


def test_identical_clouds_small_sigma_maps_to_self():
    rng = np.random.default_rng(3)
    x = rng.normal(0, 50, (60, 3))
    P = go.cpd_P(x, x, 1e-3, 0.0)
    _, yhat = go.cpd_correspondence(P, x, x)
    assert np.allclose(yhat, x, atol=1e-9)


def test_outlier_constant_formula():
    assert go.cpd_outlier_constant(100, 50, 2.0, 0.0) == 0.0
    c = go.cpd_outlier_constant(100, 50, 2.0, 0.25)
    assert abs(c - (0.25 / 0.75) * (4 * math.pi) ** 1.5 * 2.0) < 1e-12
    assert abs(co.outlier_constant(100, 50, 2.0, 0.25) - c) < 1e-13


def test_sigma2_update_equals_weighted_mean_square_distance():
    y, x = clouds(70, 80, 4)
    P = go.cpd_P(y, x, 50.0, 0.2)
    d2 = ((x[None] - y[:, None]) ** 2).sum(-1)
    direct = (P * d2).sum() / (3 * P.sum())
    assert abs(go.cpd_update_sigma2(P, x, y) - direct) < 1e-10 * direct


def test_underflow_gives_nan_not_a_silent_fix():
    y = np.zeros((3, 3))
    x = np.array([[0.1, 0, 0], [500.0, 0, 0]])
    st = go.cpd_stats_dense(y, x, 1.0, 0.0)
    assert st.den[1] == 0.0 and np.all(np.isnan(st.P1))
    st_c = co.cpd_stats(y, x, 1.0, 0.0)
    assert st_c.den[1] == 0.0 and np.all(np.isnan(st_c.P1))


@pytest.mark.parametrize("M,N,s2,w", [(1, 1, 1.0, 0.0), (33, 65, 800.0, 0.0), (200, 150, 9.0, 0.3), (64, 64, 4.0, 0.1)])
def test_streaming_c_matches_dense_numpy(M, N, s2, w):
    y, x = clouds(M, N, M + N)
    a, b = go.cpd_stats_dense(y, x, s2, w), co.cpd_stats(y, x, s2, w)
    for k in ("den", "P1", "PX", "Pt1"):
        assert np.allclose(getattr(a, k), getattr(b, k), rtol=1e-12, atol=0)
    assert abs(a.sigma2_next - b.sigma2_next) < 1e-11 * abs(a.sigma2_next)
    # row-shard partials add up to the full statistics (the multi-GPU decomposition)
    h = M // 2
    part = co.cpd_colsum_partial(y, x, s2, 0, h) + co.cpd_colsum_partial(y, x, s2, h, M)
    assert np.allclose(part + co.outlier_constant(M, N, s2, w), b.den, rtol=1e-13)
    P1b, PXb = co.cpd_rowstats_partial(y, x, s2, b.den, h, M)
    assert np.allclose(P1b, b.P1[h:], rtol=1e-13) and np.allclose(PXb, b.PX[h:], rtol=1e-13)


def test_initial_sigma2_and_gauss_block_c_vs_numpy():
    y, x = clouds(90, 110, 9)
    assert abs(go.cpd_initial_sigma2(y, x) - co.initial_sigma2(y, x)) < 1e-10
    assert np.allclose(go.gauss_block(y, x, 70.0, 50.0), co.gauss_block(y, x, 70.0, 50.0), rtol=1e-14)
    assert np.allclose(go.cpd_g_block(y, x, 3.0), np.exp(-((y[:, None] - x[None]) ** 2).sum(-1) / 18.0))


# ---------------------------------------------------------------------------------------- nearest neighbour
def test_nn_matches_kdtree_and_ties_pick_lowest_index():
    from scipy.spatial import cKDTree
    y, x = clouds(300, 500, 5)
    idx, d2, md = go.icp_closest_point(y, x)
    dist, ref = cKDTree(x).query(y)
    assert np.array_equal(idx, ref) and np.allclose(np.sqrt(d2), dist)
    cidx, cd2, cmd = co.nn(y, x)
    assert np.array_equal(cidx, idx) and np.array_equal(cd2, d2) and abs(cmd - md) < 1e-13
    xx = np.array([[0.0, 0, 0], [2.0, 0, 0], [0.0, 0, 0], [2.0, 0, 0]])
    i2, _, _ = co.nn(np.array([[1.0, 0, 0], [0.1, 0, 0], [1.9, 0, 0]]), xx)
    assert list(i2) == [0, 0, 1]


def test_icp_sigma_schedule():
    s = 100.0
    for _ in range(150):
        s = go.icp_update_sigma2(s, 100.0, 1.0, 100)
    assert s == 1.0
    assert go.icp_update_sigma2(100.0, 100.0, 1.0, 100) == 100.0 - 0.99


# ---------------------------------------------------------------------------------------- rotations / Umeyama
def test_euler_round_trip_and_convention():
    for e in [(0.3, -0.2, 0.1), (-2.0, 1.0, 2.5), (0.0, 0.0, 0.0)]:
        R = go.euler_to_rot(*e)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-15) and abs(np.linalg.det(R) - 1) < 1e-14
        assert np.allclose(go.rot_to_euler(R), e, atol=1e-12)
    # Rz(phi) Ry(theta) Rx(psi)
    cz, sz = math.cos(0.3), math.sin(0.3)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    assert np.allclose(go.euler_to_rot(0.3, 0, 0), Rz)
    # gimbal lock branch
    Rg = go.euler_to_rot(0.4, math.pi / 2, 0.1)
    assert np.allclose(go.euler_to_rot(*go.rot_to_euler(Rg)), Rg, atol=1e-8)


@pytest.mark.parametrize("similarity", [False, True])
def test_umeyama_recovers_planted_transform(similarity):
    rng = np.random.default_rng(6)
    X = rng.normal(0, 30, (200, 3))
    R0, t0, s0 = go.euler_to_rot(0.4, -0.3, 0.2), np.array([5.0, -2.0, 1.0]), (1.3 if similarity else 1.0)
    Y = s0 * (X @ R0.T) + t0
    R, t, s = go.umeyama(X, Y, similarity)
    assert np.allclose(R, R0, atol=1e-12) and np.allclose(t, t0, atol=1e-10) and abs(s - s0) < 1e-12
    # reflection guard: a mirrored target still yields a proper rotation
    Rm, _, _ = go.umeyama(X, Y * np.array([1, 1, -1.0]), similarity)
    assert abs(np.linalg.det(Rm) - 1) < 1e-12


# ---------------------------------------------------------------------------------------- GP regression
def small_model(M=80, rank=9, seed=7):
    rng = np.random.default_rng(seed)
    ref = rng.normal(0, 30, (M, 3))
    mo = go.build_gaussian_gpmm(ref, 50.0, 20.0, rel_tol=1e-9, max_rank=rank)
    return mo, rng


def test_gpmm_basis_is_orthonormal_and_reproduces_kernel():
    mo, _ = small_model(M=60, rank=180)
    assert np.allclose(mo.U.T @ mo.U, np.eye(mo.rank), atol=1e-8)
    K = go.gauss_block(mo.ref, mo.ref, 50.0, 20.0)
    Ux = mo.U[0::3, 0::3]
    assert np.allclose((Ux * mo.lam[0::3]) @ Ux.T, K, atol=1e-6)


def test_posterior_single_observation_matches_scalar_gp_formula():
    mo, _ = small_model()
    pid, v = 5, 0.7
    obs = mo.ref[pid] + np.array([1.0, -2.0, 0.5])
    mesh, a = mo.posterior_mean([pid], [obs], [v * np.eye(3)])
    Q = mo.U * np.sqrt(mo.lam)
    Kxx = Q @ Q.T                                   # prior covariance of the low-rank GP
    rows = [3 * pid, 3 * pid + 1, 3 * pid + 2]
    want = (Kxx[:, rows] @ np.linalg.solve(Kxx[np.ix_(rows, rows)] + v * np.eye(3), obs - mo.ref[pid])).reshape(-1, 3)
    assert np.allclose(mesh - mo.ref, want, atol=1e-9)


def test_posterior_noise_limits():
    mo, rng = small_model()
    alpha = rng.normal(0, 1, mo.rank)
    target = mo.instance(alpha)
    pids = np.arange(mo.M)
    tight, _ = mo.posterior_mean(pids, target, 1e-10 * np.tile(np.eye(3), (mo.M, 1, 1)))
    loose, _ = mo.posterior_mean(pids, target, 1e12 * np.tile(np.eye(3), (mo.M, 1, 1)))
    assert np.allclose(tight, target, atol=1e-6)
    assert np.allclose(loose, mo.mean_mesh(), atol=1e-6)


def test_coefficients_inverts_instance_and_transform_commutes():
    mo, rng = small_model()
    alpha = rng.normal(0, 1, mo.rank)
    assert np.allclose(mo.coefficients(mo.instance(alpha)), alpha, atol=1e-5)
    R, t = go.euler_to_rot(0.2, 0.1, -0.3), np.array([3.0, 1.0, -2.0])
    posed = mo.transform(R, t)
    assert np.allclose(posed.instance(alpha), mo.instance(alpha) @ R.T + t, atol=1e-10)
    assert np.allclose(posed.coefficients(posed.instance(alpha)), alpha, atol=1e-5)


def test_update_failure_semantics():
    mo, _ = small_model(M=40, rank=6)
    target = np.concatenate([mo.ref, [[9000.0, 0, 0]]])
    st = go.initial_state(mo, 1.0)
    s1 = go.cpd_update(mo, target, st)
    assert s1.status == go.STATUS_NONE and s1.iteration == 1 and np.array_equal(s1.alpha, st.alpha)
    s2 = go.cpd_update(mo, target, s1)
    assert s2.status == go.STATUS_MODEL_FLEXIBILITY_ERROR


def test_probabilistic_retry_counter_semantics():
    """GingrAlgorithm.scala:69-70,194-210: a sampled proposal whose posterior fails returns the state unchanged while the
    instance's retry counter lasts (10), then ModelFlexibilityError; successes give retries back; iteration 0 never fails."""
    mo, rng = small_model(M=40, rank=6)
    bad_target = np.concatenate([mo.ref, [[9000.0, 0, 0]]])        # one target far from every point at sigma2 = 1: den = 0, P = 0/0
    good_target = mo.ref + rng.normal(0, 0.1, mo.ref.shape)
    retry = go.RetryCounter()
    st = go.initial_state(mo, 1.0)
    st.iteration = 3
    z = rng.standard_normal(mo.rank)
    for k in range(10):
        st2 = go.cpd_update(mo, bad_target, st, z=z, retry=retry)
        assert st2.status == go.STATUS_NONE and st2.iteration == st.iteration + 1 and np.array_equal(st2.alpha, st.alpha)
        assert retry.value == 9 - k
        st = st2
    st11 = go.cpd_update(mo, bad_target, st, z=z, retry=retry)
    assert st11.status == go.STATUS_MODEL_FLEXIBILITY_ERROR and retry.value == 0
    # a success replenishes ONE retry (math.min(retryCounterInitialize, retryCounter + 1))
    ok = go.cpd_update(mo, good_target, st, z=z, retry=retry)
    assert ok.status == go.STATUS_NONE and retry.value == 1 and not np.array_equal(ok.alpha, st.alpha)
    # deterministic updates never retry
    retry = go.RetryCounter()
    assert go.cpd_update(mo, bad_target, st, retry=retry).status == go.STATUS_MODEL_FLEXIBILITY_ERROR and retry.value == 10
    # iteration 0: unchanged, no error, counter untouched
    st0 = go.initial_state(mo, 1.0)
    s1 = go.cpd_update(mo, bad_target, st0, z=z, retry=retry)
    assert s1.status == go.STATUS_NONE and retry.value == 10


def test_failed_projection_is_an_error_even_at_iteration_zero():
    """GingrAlgorithm.scala:248-251: only the POSTERIOR failure is forgiven at iteration 0; a failed coefficients() gives
    ModelFlexibilityError at any iteration.  An infinite step length makes the blended coefficients non-finite."""
    mo, rng = small_model(M=40, rank=6)
    target = mo.ref + rng.normal(0, 0.1, mo.ref.shape)
    st = go.initial_state(mo, 1.0, step_length=float("inf"))
    with np.errstate(all="ignore"):
        s1 = go.cpd_update(mo, target, st)
    assert st.iteration == 0 and s1.status == go.STATUS_MODEL_FLEXIBILITY_ERROR and np.array_equal(s1.alpha, st.alpha)


def test_fit_scale_is_applied_after_the_rigid_transform():
    mo, rng = small_model(M=30, rank=6)
    st = go.State(alpha=rng.normal(0, 1, 6), euler=(0.1, 0.2, 0.3), center=np.zeros(3), translation=np.array([1.0, 2, 3]),
                  scale=1.5, sigma2=1.0, fit=np.zeros((30, 3)))
    want = 1.5 * (mo.instance(st.alpha) @ go.euler_to_rot(0.1, 0.2, 0.3).T + st.translation)
    assert np.allclose(go.model_instance_shape_pose_scale(mo, st), want, atol=1e-12)


# ---------------------------------------------------------------------------------------- golden fixtures
@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "inputs.npz")), np.load(os.path.join(GOLD, "expected.npz"))


def test_golden_femur_stats_numpy_and_c(gold):
    inp, exp = gold
    y, x = inp["femur"].astype(np.float64), inp["femur_target"].astype(np.float64)
    assert y.shape == (1622, 3) and x.shape == (1622, 3)
    assert abs(co.initial_sigma2(y, x) - float(exp["femur_sigma2_init"])) < 1e-9
    for tag in ("s1_w01", "sinit_w0", "s25_w0"):
        s2, w = exp[f"cpd_{tag}_args"]
        st = co.cpd_stats(y, x, float(s2), float(w))
        assert np.allclose(st.den, exp[f"cpd_{tag}_den"], rtol=1e-12)
        assert np.allclose(st.P1, exp[f"cpd_{tag}_P1"], rtol=1e-11)
        assert np.allclose(st.PX, exp[f"cpd_{tag}_PX"], rtol=1e-10, atol=1e-13)
        assert abs(st.sigma2_next - exp[f"cpd_{tag}_scalars"][1]) < 1e-9 * abs(st.sigma2_next)


def test_golden_bunny_nn(gold):
    inp, exp = gold
    idx, _, md = co.nn(exp["nn_query"].astype(np.float64), inp["bunny5k"].astype(np.float64))
    assert np.array_equal(idx, exp["nn_idx"]) and abs(md - float(exp["nn_mean_distance"])) < 1e-12


def test_golden_update_trajectory_one_step(gold):
    inp, exp = gold
    y, x = inp["femur"].astype(np.float64), inp["femur_target"].astype(np.float64)
    mo = go.PDM(ref=y, mean=np.zeros_like(y), U=exp["gpmm_basis"], lam=exp["gpmm_variance"])
    st = go.initial_state(mo, float(exp["femur_sigma2_init"]))
    st = go.cpd_update(mo, x, st, w=0.0, stats=co.cpd_stats(st.fit, x, st.sigma2, 0.0))
    assert np.allclose(st.alpha, exp["cpd_rigid_it1_alpha"], rtol=1e-7, atol=1e-10)
    pose = exp["cpd_rigid_it1_pose"]
    assert np.allclose([*st.euler, *st.translation, st.scale, st.sigma2], pose[:8], rtol=1e-8, atol=1e-11)
    assert all(exp[f"{t}_it5_pose"][8] == 0 for t in ("cpd_rigid", "cpd_rigid_lm_w"))


# ---------------------------------------------------------------------------------------- posterior sample / logpdf (8f rank 1)
def test_posterior_model_covariance_and_sampling_square_roots():
    """scalismo samples the posterior in its SVD basis (U innerU, lambda_p); the HIP path samples a + L^-T z.  Both are square
    roots of the same coefficient covariance D Minv D, and the coefficient norm used by logpdf is basis independent."""
    mo, rng = small_model(M=60, rank=12)
    pids = np.arange(mo.M)
    obs = mo.instance(rng.normal(0, 1, mo.rank)) + rng.normal(0, 0.3, (mo.M, 3))
    var = rng.uniform(0.5, 3.0, mo.M)
    covs = var[:, None, None] * np.eye(3)[None]
    post = mo.posterior_model(pids, obs, covs)
    Q = mo.U * np.sqrt(mo.lam)
    Mm = np.eye(mo.rank) + Q.T @ (Q / np.repeat(var, 3)[:, None])
    Minv = np.linalg.inv(Mm)
    cov_scalismo = (post.U * post.lam) @ post.U.T                    # U_p Lambda_p U_p^T
    assert np.allclose(cov_scalismo, Q @ Minv @ Q.T, atol=1e-9)
    L = np.linalg.cholesky(Mm)
    F = np.linalg.solve(L.T, np.eye(mo.rank))                         # L^-T: Cov(a + L^-T z) = Minv
    assert np.allclose(F @ F.T, Minv, atol=1e-12)
    # logpdf of a mesh: scalismo's route vs the closed form c = L^T (S + eps Mm)^-1 (Q^T d - S a)
    mesh = post.instance(rng.normal(0, 1, mo.rank))
    c_ref = post.coefficients(mesh)
    S = Q.T @ Q
    a = Minv @ (Q.T @ ((obs - mo.ref - mo.mean).reshape(-1) / np.repeat(var, 3)))
    b = Q.T @ (mesh - mo.ref - mo.mean).reshape(-1) - S @ a
    c_closed = L.T @ np.linalg.solve(S + 1e-5 * Mm, b)
    assert abs(c_ref @ c_ref - c_closed @ c_closed) < 1e-6 * (c_ref @ c_ref)
    assert abs(go.gp_logpdf(c_ref) - (-0.5 * c_closed @ c_closed - 0.5 * mo.rank * math.log(2 * math.pi))) < 1e-6 * abs(go.gp_logpdf(c_ref))


# ------------------------------------------------------------------------------------------- GPMM construction (8f rank 3)
def test_matrix_valued_pivoted_cholesky_is_three_interleaved_scalar_factorisations():
    """DiagonalKernel: pivots come as (P,x),(P,y),(P,z); the generic factor restricted to one coordinate is the scalar
    factor; the residual trace obeys the stopping rule."""
    rng = np.random.default_rng(8)
    P = rng.normal(0, 30, (120, 3))
    sig, sc, tol = [35.0, 12.0], [9.0, 2.0], 0.03
    L = go.pivoted_cholesky_matrix_valued(P, sig, sc, tol)
    n = L.shape[1]
    K = go.gaussian_mixture_kernel(P, P, sig, sc)
    Kf = np.zeros((360, 360))
    for d in range(3):
        Kf[d::3, d::3] = K
    resid = np.trace(Kf - L @ L.T)
    assert resid < tol * np.trace(Kf)
    Lprev = L[:, :n - 1]
    assert np.trace(Kf - Lprev @ Lprev.T) >= tol * np.trace(Kf)       # one column fewer would not have stopped
    _, piv = go.pivoted_cholesky_matrix_valued(P, sig, sc, tol, return_pivots=True)
    assert [q % 3 for q in piv] == [k % 3 for k in range(n)]
    assert all(piv[3 * i] // 3 == piv[3 * i + 1] // 3 == piv[3 * i + 2] // 3 for i in range(n // 3))
    # scalar factor of the same kernel: columns of coordinate x
    Ls = L[0::3][:, 0::3]
    ks = Ls.shape[1]
    assert np.abs(Ls @ Ls.T - (L @ L.T)[0::3, 0::3]).max() < 1e-12
    assert ks == (n + 2) // 3


def test_approximate_eig_is_the_eigendecomposition_of_the_low_rank_kernel():
    rng = np.random.default_rng(9)
    P = rng.normal(0, 20, (90, 3))
    m = go.build_gpmm_mixture(P, [25.0], [4.0], 0.02)
    L = go.pivoted_cholesky_matrix_valued(P, [25.0], [4.0], 0.02)
    assert np.abs(m.U.T @ m.U - np.eye(m.rank)).max() < 1e-10
    assert np.all(np.diff(m.lam) <= 1e-9 * m.lam[0])
    assert np.abs((m.U * m.lam) @ m.U.T - L @ L.T).max() < 1e-10 * m.lam[0]
    ev = np.linalg.eigvalsh(L @ L.T)[::-1][:m.rank]
    assert np.abs(ev - m.lam).max() < 1e-9 * m.lam[0]


def test_pointset_distance_extrema_against_scipy():
    from scipy.spatial.distance import pdist
    P = np.random.default_rng(10).normal(0, 50, (200, 3))
    mx, mn = go.pointset_distance_extrema(P)
    d = pdist(P)
    assert abs(mx - d.max()) < 1e-12 * mx and abs(mn - d.min()) < 1e-12 * mx
    sig, sc = go.automatic_gaussian_parameters(P)
    assert sig == [mx / 4.0, mx / 8.0] and sc == [mx / 8.0, mx / 16.0]


# ------------------------------------------------------------------------------------------- surface ICP (8f rank 2)
def test_closest_point_on_triangle_regions():
    A, B, C = np.array([[0.0, 0, 0]]), np.array([[4.0, 0, 0]]), np.array([[0.0, 3, 0]])
    cases = {(1.0, 1.0, 2.0): (1.0, 1.0, 0.0),        # interior: orthogonal projection
             (-1.0, -1.0, 1.0): (0.0, 0.0, 0.0),      # vertex A
             (6.0, -1.0, 0.0): (4.0, 0.0, 0.0),       # vertex B
             (-1.0, 5.0, 0.0): (0.0, 3.0, 0.0),       # vertex C
             (2.0, -3.0, 1.0): (2.0, 0.0, 0.0),       # edge AB
             (-2.0, 1.5, 0.0): (0.0, 1.5, 0.0)}       # edge AC
    for p, q in cases.items():
        assert np.allclose(go.closest_point_on_triangles(np.array(p), A, B, C)[0], q, atol=1e-15), p
    q = go.closest_point_on_triangles(np.array([4.0, 3.0, 0.0]), A, B, C)[0]      # edge BC: foot of the perpendicular
    assert abs((q - B[0]) @ (C[0] - B[0]) / 25.0 - 0.36) < 1e-15 and abs(np.cross(q - B[0], C[0] - B[0])).max() < 1e-12
    # brute force: no sampled surface point is closer
    rng = np.random.default_rng(2)
    for _ in range(50):
        T = rng.normal(0, 1, (3, 3))
        p = rng.normal(0, 2, 3)
        c = go.closest_point_on_triangles(p, T[0:1], T[1:2], T[2:3])[0]
        u = rng.dirichlet([1, 1, 1], 4000)
        samples = u @ T
        assert ((samples - p) ** 2).sum(1).min() >= ((c - p) ** 2).sum() - 1e-12


def test_mesh_topology_helpers():
    # 3 x 3 grid: the 8 outer vertices are boundary, the centre is not; normals of a flat CCW sheet point to +z
    idx = np.arange(9).reshape(3, 3)
    v = np.array([[i, j, 0.0] for i in range(3) for j in range(3)])
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
    bnd = go.boundary_vertices(9, tris)
    assert bnd.sum() == 8 and not bnd[4]
    n = go.vertex_normals(v, tris)
    assert np.allclose(n, [0, 0, 1])
    # a line through the sheet hits it once; a line through a corner vertex returns exactly that vertex
    ip = go.line_mesh_intersections(np.array([0.7, 0.6, 2.0]), np.array([0.0, 0.0, -1.0]), v, tris)
    assert ip.shape[0] >= 1 and np.allclose(ip, [0.7, 0.6, 0.0])
    p = v[4]
    ips = go.line_mesh_intersections(p, np.array([0.3, -0.2, 1.0]), v, tris)
    assert ips.shape[0] >= 1 and all(np.array_equal(q, p) for q in ips)


def _octahedron():
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64) * 10.0
    t = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    return v, t


def test_nicp_edges_matrix_and_defaults():
    """NonRigidOptimalStepICP.scala:63-87: unique sorted edges, M = +1 / -1 per edge, M^T M = graph Laplacian, eleven times alpha 10."""
    v, t = _octahedron()
    e = go.nicp_edges(t)
    assert e.shape == (12, 2) and np.all(e[:, 0] < e[:, 1])
    M = go.nicp_matrix_m(e, 6)
    assert np.array_equal(M.T @ M, go.graph_laplacian(6, t))
    assert go.NICP_DEFAULT_ALPHA == [10.0] * 11


def test_nicp_known_answers():
    """Closed-form cases of the two least-squares steps: a target that is the translated template is reached exactly by N-ICP-T
    whatever the stiffness (a constant displacement costs nothing), and a target that is an affine image of the template is reached
    by N-ICP-A (one common affine map costs nothing) -- both with all weights 1 (closed convex meshes, same orientation)."""
    v, t = _octahedron()
    e = go.nicp_edges(t)
    none = np.zeros(0, dtype=np.int64)
    shift = np.array([0.3, -0.2, 0.1])
    cp, w, dist = go.surface_correspondence(v, t, v + shift, t)
    assert np.all(w == 1.0)
    got, d = go.nicp_iteration_t(v, t, v + shift, t, e, none, np.zeros((0, 3)), 50.0, 1.0)
    # the closest surface points are not the translated vertices, but the stiff limit moves the template rigidly by their mean
    assert np.allclose(got - v, (got - v).mean(0), atol=2e-3) and abs(d - dist) < 1e-15
    A = np.array([[1.02, 0.01, 0.0], [0.0, 0.98, 0.02], [0.01, 0.0, 1.01]])
    tgt = v @ A.T + shift
    cp, w, _ = go.surface_correspondence(v, t, tgt, t)
    gotA, _, lm = go.nicp_iteration_a(v, t, tgt, t, e, none, np.zeros((0, 3)), 1e4, 1.0)
    # stiff limit of N-ICP-A: ONE affine map for all vertices, the least-squares affine fit of the correspondences
    Q = np.concatenate([v, np.ones((6, 1))], axis=1)
    X = np.linalg.lstsq(Q, cp, rcond=None)[0]
    assert np.allclose(gotA, Q @ X, atol=1e-4) and lm.shape == (0, 3)
    # landmarks: N-ICP-A zeroes the weight of the landmark vertex and pulls it to the landmark target
    ids = np.array([4])
    ul = np.array([[0.0, 0.0, 12.0]])
    gotL, _, lmL = go.nicp_iteration_a(v, t, tgt, t, e, ids, ul, 0.01, 100.0)
    assert np.allclose(lmL[0], ul[0], atol=1e-3) and np.allclose(gotL[4], lmL[0])


def test_c_mesh_closest_point_equals_the_numpy_restatement():
    """oracle/cpd_oracle.c:oracle_mesh_closest_point (brute force, used to check the device scan on full-size meshes) against
    gingr_oracle.mesh_closest_point: same points, distances and tie rule, bit for bit."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(12)
    v = rng.normal(size=(80, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v *= 10.0
    t = ConvexHull(v).simplices
    P = np.concatenate([rng.normal(size=(150, 3)) * 12.0, v[:20], (v[t[:10, 0]] + v[t[:10, 1]]) / 2.0])   # off-surface, vertices, edge midpoints
    cp, d2, tid = co.mesh_closest_point(P, v, t)
    ocp, od2 = go.mesh_closest_point(P, v, t)
    assert np.array_equal(cp, ocp) and np.array_equal(d2, od2)
    A, B, C = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    for i in range(P.shape[0]):                                    # the reported triangle is the first minimum
        q = go.closest_point_on_triangles(P[i], A, B, C)
        dd = q - P[i]
        dist = dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1] + dd[:, 2] * dd[:, 2]
        assert tid[i] == int(np.argmin(dist))

"""The scalismo-side pieces of the oracle (rows S1-S8, restated from the published algorithms because
scalismo's source is not under /root/reference) checked against INDEPENDENT third-party implementations that are in this image:
LAPACK's pivoted Cholesky (dpstrf), scipy's rotations and Procrustes solution, the textbook full-covariance Gaussian-process
regression, scipy's k-d tree and a brute-force sampled surface.  None of these checks can replace a run of the reference (the
oracle stays "parity unpinned"), but each one rules out a private misreading of the formula in question.  CPU only."""
import math

import numpy as np
import pytest

from oracle import gingr_oracle as go


def _cloud(n, seed, scale=30.0):
    return np.random.default_rng(seed).normal(size=(n, 3)) * scale


# ---- rotations -------------------------------------------------------------------------------------------------------------------

def test_euler_matrix_is_scipys_extrinsic_xyz():
    """Rz(phi) Ry(theta) Rx(psi) (RotationSpace3D, 'x-convention') = scipy's extrinsic 'xyz' rotation by (psi, theta, phi)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)
    for _ in range(50):
        phi, psi = rng.uniform(-math.pi, math.pi, 2)
        theta = rng.uniform(-math.pi / 2 + 0.05, math.pi / 2 - 0.05)
        R = go.euler_to_rot(phi, theta, psi)
        assert np.allclose(R, Rotation.from_euler("xyz", [psi, theta, phi]).as_matrix(), atol=1e-14)
        assert np.allclose(R, Rotation.from_euler("ZYX", [phi, theta, psi]).as_matrix(), atol=1e-14)   # the intrinsic reading
        back = Rotation.from_matrix(R).as_euler("xyz")
        assert np.allclose(back, [psi, theta, phi], atol=1e-10)
        assert np.allclose(go.rot_to_euler(R), [phi, theta, psi], atol=1e-10)


@pytest.mark.parametrize("similarity", [False, True])
def test_umeyama_against_scipy_procrustes(similarity):
    """The rotation is the orthogonal Procrustes solution of the centred clouds (scipy: SVD based, written independently); the
    scale is Umeyama's tr(D S) / var(src); the translation maps the centroids."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(17)
    for trial in range(10):
        X = _cloud(200, 100 + trial)
        Rt = Rotation.random(random_state=trial).as_matrix()
        s = 1.0 + 0.3 * rng.uniform(-1, 1) if similarity else 1.0
        Y = s * X @ Rt.T + rng.normal(size=3) * 10 + rng.normal(size=X.shape) * 0.5
        R, t, c = go.umeyama(X, Y, similarity)
        Xc, Yc = X - X.mean(0), Y - Y.mean(0)
        rot, rssd = Rotation.align_vectors(Yc, Xc)             # minimises sum |Yc_i - R Xc_i|^2 (Kabsch)
        assert np.allclose(R, rot.as_matrix(), atol=1e-9)
        if similarity:
            c_ref = float(np.sum(Yc * (Xc @ rot.as_matrix().T)) / np.sum(Xc * Xc))     # least-squares scale given R
            assert abs(c - c_ref) < 1e-10
        else:
            assert c == 1.0
        assert np.allclose(t, Y.mean(0) - c * R @ X.mean(0), atol=1e-10)
        # optimality: no nearby transform does better
        base = np.sum((c * X @ R.T + t - Y) ** 2)
        for k in range(5):
            dR = Rotation.from_rotvec(rng.normal(size=3) * 1e-3).as_matrix()
            assert np.sum((c * X @ (dR @ R).T + t - Y) ** 2) >= base - 1e-9


# ---- pivoted Cholesky ------------------------------------------------------------------------------------------------------------

def test_pivoted_cholesky_against_lapack_dpstrf():
    """LAPACK dpstrf: the same greedy diagonal pivoting (largest remaining diagonal, first maximum).  Run to a fixed rank on a
    generic cloud (no ties) the pivots and the factor agree; the trace stop rule is then checked on the residual."""
    from scipy.linalg.lapack import dpstrf
    P = _cloud(120, 3, 25.0)
    sigma, scaling = 40.0, 12.0
    K = go.gauss_block(P, P, sigma, scaling)
    L = go.pivoted_cholesky_scalar(P, sigma, scaling, rel_tol=1e-3)
    k = L.shape[1]
    c, piv, rank, info = dpstrf(K, lower=1, tol=-1.0)
    assert rank >= k
    Lp = np.tril(c)[:, :k]                                      # factor of the PERMUTED matrix
    perm = piv - 1
    Lfull = np.zeros_like(Lp)
    Lfull[perm] = Lp
    assert np.allclose(L, Lfull, atol=1e-9)
    # pivots of the oracle = rows where column j has its defining entry (diagonal of the permuted factor)
    for j in range(k):
        assert abs(L[perm[j], j] - Lp[j, j]) < 1e-9
    resid = np.trace(K - L @ L.T)
    assert resid <= 1e-3 * np.trace(K)
    assert np.trace(K - L[:, :k - 1] @ L[:, :k - 1].T) > 1e-3 * np.trace(K)        # one column fewer would not have stopped


def test_generic_pivoted_cholesky_against_lapack_on_the_full_3m_matrix():
    """The 3M x 3M matrix of a DiagonalKernel with three different scalar kernels, factorised by LAPACK, against the generic
    restatement (point-major (point, coordinate) order)."""
    from scipy.linalg.lapack import dpstrf
    P = _cloud(40, 9, 20.0)
    M = P.shape[0]
    sig = [30.0, 45.0, 60.0]
    sc = [5.0, 7.0, 11.0]
    blocks = [go.gauss_block(P, P, sig[d], sc[d]) for d in range(3)]
    K = np.zeros((3 * M, 3 * M))
    for d in range(3):
        K[d::3, d::3] = blocks[d]

    def kfun(d, rows, j):
        return blocks[d][rows, j]

    L, pivots = go.pivoted_cholesky_diagonal_kernel(M, kfun, rel_tol=1e-2, return_pivots=True)
    k = L.shape[1]
    c, piv, rank, info = dpstrf(K, lower=1, tol=-1.0)
    assert list(piv[:k] - 1) == pivots
    Lfull = np.zeros((3 * M, k))
    Lfull[piv - 1] = np.tril(c)[:, :k]
    assert np.allclose(L, Lfull, atol=1e-9)


def test_approximate_eig_against_dense_eigendecomposition():
    """computeApproximateEig: eigenpairs of L L^T through the small Gram matrix L^T L = those of the dense matrix."""
    P = _cloud(60, 4, 20.0)
    L = go.pivoted_cholesky_scalar(P, 35.0, 9.0, rel_tol=1e-2)
    U, lam = go.approximate_eig(L)
    w, V = np.linalg.eigh(L @ L.T)
    order = np.argsort(w)[::-1][:L.shape[1]]
    assert np.allclose(lam, w[order], rtol=1e-9, atol=1e-9)
    for j in range(L.shape[1]):
        assert abs(abs(U[:, j] @ V[:, order[j]]) - 1.0) < 1e-7


# ---- Gaussian-process regression -------------------------------------------------------------------------------------------------

def _model(M=50, seed=2):
    ref = _cloud(M, seed, 25.0)
    pdm = go.build_gaussian_gpmm(ref, 40.0, 10.0, rel_tol=1e-2)
    return ref, pdm


def test_posterior_mean_is_textbook_gp_regression_with_the_low_rank_covariance():
    """mean_p = mu + K_{.,obs} (K_obs,obs + Sigma)^-1 (y - mu_obs) with K = U diag(lam) U^T: the weight-space form the oracle
    restates (genericRegressionComputations: M = Q^T Sigma^-1 Q + I) and the function-space form of any GP textbook are
    the same posterior; written here with dense 3K x 3K solves and full 3x3 noise blocks."""
    ref, pdm = _model()
    rng = np.random.default_rng(8)
    M = ref.shape[0]
    pids = rng.choice(M, 20, replace=False)
    covs = np.empty((20, 3, 3))
    for k in range(20):
        A = rng.normal(size=(3, 3))
        covs[k] = A @ A.T + 0.5 * np.eye(3)
    pdm.mean[:] = rng.normal(size=pdm.mean.shape)
    pts = ref[pids] + rng.normal(size=(20, 3)) * 3.0
    mean_mesh, a = pdm.posterior_mean(pids, pts, covs)
    Kfull = (pdm.U * pdm.lam[None, :]) @ pdm.U.T
    rows = (3 * pids[:, None] + np.arange(3)[None, :]).reshape(-1)
    Sigma = np.zeros((60, 60))
    for k in range(20):
        Sigma[3 * k:3 * k + 3, 3 * k:3 * k + 3] = covs[k]
    y = (pts - ref[pids]).reshape(-1)
    mu = pdm.mean.reshape(-1)
    post = mu + Kfull[:, rows] @ np.linalg.solve(Kfull[np.ix_(rows, rows)] + Sigma, y - mu[rows])
    assert np.allclose(mean_mesh, ref + post.reshape(M, 3), atol=1e-9)
    # and the covariance of the posterior model (posterior.sample / logpdf rows): K - K_{.,obs} (..)^-1 K_{obs,.}
    pm = pdm.posterior_model(pids, pts, covs)
    Kp = (pm.U * pm.lam[None, :]) @ pm.U.T
    Kp_ref = Kfull - Kfull[:, rows] @ np.linalg.solve(Kfull[np.ix_(rows, rows)] + Sigma, Kfull[rows, :])
    assert np.allclose(Kp, Kp_ref, atol=1e-8)


def test_coefficients_is_ridge_regression_with_noise_1e_minus_5():
    """coefficients(mesh) = argmin |d - mu - Q a|^2 / 1e-5 + |a|^2 (the MAP estimate under N(0, I) on the coefficients)."""
    ref, pdm = _model(40, 6)
    rng = np.random.default_rng(1)
    mesh = ref + rng.normal(size=ref.shape) * 2.0
    a = pdm.coefficients(mesh)
    Q = pdm.U * np.sqrt(pdm.lam)[None, :]
    d = (mesh - ref - pdm.mean).reshape(-1)
    A = np.vstack([Q / math.sqrt(1e-5), np.eye(pdm.rank)])
    b = np.concatenate([d / math.sqrt(1e-5), np.zeros(pdm.rank)])
    a_ref = np.linalg.lstsq(A, b, rcond=None)[0]
    assert np.allclose(a, a_ref, rtol=1e-7, atol=1e-9)


# ---- closest points --------------------------------------------------------------------------------------------------------------

def test_mesh_closest_point_against_a_densely_sampled_surface():
    """closestPointOnSurface: no sampled surface point is closer than the returned one, and the densest sample comes within the
    sampling resolution of it."""
    from scipy.spatial import cKDTree
    verts = _cloud(30, 21, 10.0)
    from scipy.spatial import ConvexHull
    tris = ConvexHull(verts).simplices.astype(np.int64)
    # barycentric grid on every triangle
    n = 24
    bary = np.array([(i / n, j / n, 1 - i / n - j / n) for i in range(n + 1) for j in range(n + 1 - i)])
    A, B, C = verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]
    samples = (bary[None, :, 0, None] * A[:, None] + bary[None, :, 1, None] * B[:, None] + bary[None, :, 2, None] * C[:, None]).reshape(-1, 3)
    tree = cKDTree(samples)
    Q = _cloud(200, 22, 14.0)
    cp, d2 = go.mesh_closest_point(Q, verts, tris)
    ds, _ = tree.query(Q)
    exact = np.sqrt(np.sum((cp - Q) ** 2, axis=1))
    edge = np.max(np.linalg.norm(B - A, axis=1))
    assert np.all(exact <= ds + 1e-12)
    assert np.all(ds <= exact + edge / n * 1.5)
    assert np.allclose(d2, exact * exact, rtol=1e-12)
    # the returned point lies on the surface: its own closest point is itself
    cp2, d22 = go.mesh_closest_point(cp, verts, tris)
    assert np.all(d22 <= 1e-18 * (1 + np.sum(cp * cp, axis=1)))


def test_graph_laplacian_pseudo_inverse_against_scipy():
    """LaplacianHelper: the pseudo-inverse with singular values below 1e-5 dropped = scipy's pinv with the matching cutoff; the
    Laplacian of a connected mesh has exactly one zero singular value (the constant vector)."""
    import scipy.linalg
    from scipy.spatial import ConvexHull
    verts = _cloud(25, 31, 10.0)
    verts /= np.linalg.norm(verts, axis=1, keepdims=True)            # on a sphere: every vertex is on the hull
    tris = ConvexHull(verts).simplices.astype(np.int64)
    assert np.unique(tris).size == 25
    Lm = go.graph_laplacian(25, tris)
    assert np.allclose(Lm.sum(axis=1), 0.0)
    s = np.linalg.svd(Lm, compute_uv=False)
    assert np.sum(s < 1e-5) == 1
    pi = go.pinv_svd(Lm)
    assert np.allclose(pi, scipy.linalg.pinv(Lm, atol=1e-5), atol=1e-10)
    assert np.allclose(Lm @ pi @ Lm, Lm, atol=1e-9)

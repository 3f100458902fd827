"""The remaining kernels of GPMMTriangleMesh3D / SimpleTriangleModels3D.create (G/api/gpmm/GPMMHelper.scala:103-142,
KernelHelper.scala, LaplacianHelper.scala, G/simple/SimpleModels.scala:52-75) built on the device -- linear ("Dot"), x-mirrored
("Symmetry": different kernels per coordinate, so the generic pivot order interleaves two scalar factorisations) and the
inverse graph Laplacian lookup -- against the oracle's generic restatement of scalismo's pivoted Cholesky + approximate
eigen-decomposition.  Compared as in test_gpu_gpmm.py: rank, spectrum, covariance."""
import os

import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def cov_of(U, lam, rows):
    A = U[rows] * np.sqrt(lam)[None, :]
    return A @ A.T


def cloud(M=400, seed=5, scale=40.0):
    return np.random.default_rng(seed).normal(0, scale, (M, 3))


def check_model_against_oracle(dm_host, mo, tol_lam=1e-9, tol_cov=1e-9):
    assert dm_host.variance.shape[0] == mo.rank
    assert np.linalg.norm(dm_host.variance - mo.lam) < tol_lam * np.linalg.norm(mo.lam)
    U = np.asarray(dm_host.basis)
    G = U.T @ U                                      # unit, mutually orthogonal columns
    assert np.abs(G - np.eye(G.shape[0])).max() < 1e-9
    rows = np.random.default_rng(0).permutation(U.shape[0])[:300]
    Cd, Co = cov_of(U, dm_host.variance, rows), cov_of(mo.U, mo.lam, rows)
    assert np.abs(Cd - Co).max() < tol_cov * np.abs(Co).max()


def femur_small(n=220):
    from gingr_amd.simple import cluster_decimate
    d = np.load(os.path.join(HERE, "golden", "inputs.npz"))
    m = np.load(os.path.join(HERE, "golden", "femur_mesh.npz"))
    return cluster_decimate(d["femur"].astype(np.float64), m["femur_cells"], n)


@pytest.mark.parametrize("scaling,tol", [(0.05, 0.01), (2.0, 0.0)])
def test_dot_product_kernel_model(ctx, scaling, tol):
    import gingr_amd as ga
    ref = cloud(300, 21) + np.array([30.0, -10.0, 5.0])
    mo = go.build_gpmm_diagonal(ref, go.dot_kernel_fun(ref, scaling), tol, 9)
    g = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=tol, maxRank=9)
    dm = g.GaussianDot(sigma=123.0, scaling=scaling)                  # sigma is ignored by the reference's DotProductKernel
    assert dm.rank == mo.rank <= 9
    check_model_against_oracle(dm.to_host(), mo, tol_lam=1e-8, tol_cov=1e-8)
    dm2 = g.InverseLaplacianDot(scaling=scaling, gamma=7.0)           # = the same linear kernel
    assert np.array_equal(dm2.to_host().variance, dm.to_host().variance)


@pytest.mark.parametrize("sigma,scaling,tol,max_rank", [(60.0, 30.0, 0.02, 0), (45.0, 10.0, 0.0, 40), (45.0, 10.0, 0.0, 41),
                                                         (80.0, 5.0, 0.1, 0)])
def test_mirrored_gaussian_kernel_model(ctx, sigma, scaling, tol, max_rank):
    import gingr_amd as ga
    ref = cloud(260, 22, 35.0)
    mo = go.build_gpmm_diagonal(ref, go.symmetric_gauss_kernel_fun(ref, sigma, scaling), tol, max_rank or None)
    dm = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=tol, maxRank=max_rank).GaussianSymmetry(sigma, scaling)
    assert dm.rank == mo.rank, (dm.rank, mo.rank)
    check_model_against_oracle(dm.to_host(), mo, tol_lam=1e-8, tol_cov=1e-8)
    # the model is symmetric about the plane x = 0: mirroring a sample's input point mirrors its deformation -- covariance check
    host = dm.to_host()
    A = host.basis * np.sqrt(host.variance)[None, :]
    C = A[:30] @ A[:30].T                                             # first 10 points, all coordinates
    assert np.allclose(C, C.T, atol=1e-9)


def test_mirrored_model_on_an_exactly_symmetric_cloud(ctx):
    """Mirror pairs (x, y, z) / (-x, y, z): the x-coordinate kernel k - km vanishes on the plane and makes pairs anti-correlated,
    the y / z kernel k + km makes them identical -- and initial diagonals tie exactly between partners (first index wins)."""
    import gingr_amd as ga
    half = cloud(90, 23, 30.0)
    half[:, 0] = np.abs(half[:, 0]) + 1.0
    ref = np.concatenate([half, half * np.array([-1.0, 1.0, 1.0])])
    mo = go.build_gpmm_diagonal(ref, go.symmetric_gauss_kernel_fun(ref, 50.0, 20.0), 0.01)
    dm = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).GaussianSymmetry(50.0, 20.0)
    assert dm.rank == mo.rank
    check_model_against_oracle(dm.to_host(), mo, tol_lam=1e-8, tol_cov=1e-8)
    host = dm.to_host()
    a = np.random.default_rng(1).normal(0, 1, host.rank)
    disp = (host.basis @ (np.sqrt(host.variance) * a)).reshape(-1, 3)
    assert np.allclose(disp[90:], disp[:90] * np.array([-1.0, 1.0, 1.0]), atol=1e-8)    # samples are mirror symmetric


def test_inverse_laplacian_kernel_model(ctx):
    import gingr_amd as ga
    from gingr_amd.api import LaplacianHelper
    v, c = femur_small(220)
    L = go.graph_laplacian(v.shape[0], c)
    h = LaplacianHelper(v.shape[0], c)
    assert np.array_equal(h.laplacianMatrix(), L)
    m = go.pinv_svd(L)
    assert np.allclose(h.inverseLaplacianMatrix(), m, atol=1e-12) and np.allclose(L @ m @ L, L, atol=1e-9)
    # the inverse Laplacian's spectrum decays slowly: 1 % of the trace needs 629 of the 660 components (above the model limit of
    # 512), so the comparison runs at 30 % and at a fixed rank
    for tol, max_rank in ((0.3, 0), (0.0, 100)):
        mo = go.build_gpmm_diagonal(v, go.lookup_kernel_fun(m, 30.0), tol, max_rank or None)
        dm = ga.GPMMTriangleMesh3D(ctx, v, relativeTolerance=tol, maxRank=max_rank, cells=c).InverseLaplacian(scaling=30.0)
        assert dm.rank == mo.rank, (dm.rank, mo.rank)
        check_model_against_oracle(dm.to_host(), mo, tol_lam=1e-8, tol_cov=1e-8)
    with pytest.raises(ValueError):
        ga.GPMMTriangleMesh3D(ctx, v, relativeTolerance=0.01).InverseLaplacian(scaling=30.0)


def test_simple_triangle_models_dispatch_and_registration(ctx):
    """SimpleTriangleModels3D.create for every kernel choice; a CPD registration runs with each model."""
    import gingr_amd as ga
    from gingr_amd import simple as sp
    v, c = femur_small(400)
    mesh = ga.TriangleMesh3D(v, c)
    rng = np.random.default_rng(5)
    want_rank = {}
    for k in (sp.GaussKernel(50.0, 70.0), sp.GaussMixKernel(), sp.GaussDotKernel(0.05, 70.0), sp.GaussMirrorKernel(50.0, 70.0),
              sp.InvLapKernel(30.0), sp.InvLapDotKernel(0.02, 1.0)):
        model = sp.SimpleTriangleModels3D.create(ctx, mesh, k, relativeTolerance=0.05, maxRank=120)
        assert model.rank >= 3 and model.cells is not None, k
        want_rank[k.name] = model.rank
        host = model.to_host()
        # a sample of the model plus measurement noise (an exact sample lets sigma2 collapse to 0: the reference's w = 0 hazard)
        target = v + (host.basis @ (np.sqrt(host.variance) * rng.normal(0, 0.5, host.rank))).reshape(-1, 3) + rng.normal(0, 0.2, v.shape)
        cpd = ga.CpdRegistration(ctx)
        st = cpd.run(cpd.createInitialState(model, target, ga.CpdConfiguration(maxIterations=20, w=0.05),
                                            transform=ga.GlobalTranformationType.NoTransforms))
        assert st.general.status in (ga.FittingStatuses.Converged, ga.FittingStatuses.MaxIteration), k
        d0 = np.sqrt(((v - target) ** 2).sum(1)).mean()
        d1 = np.sqrt(((st.general.fit - target) ** 2).sum(1)).mean()
        assert d1 < 0.6 * d0, (k, d0, d1)
    assert want_rank["GaussDot"] == want_rank["InvLapDot"] <= 9


def test_row_shards_of_a_two_kernel_model(ctx):
    import gingr_amd as ga
    ref = cloud(300, 24, 35.0)
    dm = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.02).GaussianSymmetry(55.0, 12.0)
    full = dm.to_host()
    for lo, hi in ((0, 100), (100, 300)):
        part = ga.DeviceModel(ctx, dm, lo, hi).download()
        assert np.array_equal(part.variance, full.variance)
        assert np.array_equal(part.basis, full.basis[3 * lo:3 * hi])


def test_argument_errors_of_the_new_entry_points(ctx):
    """gingr_gpmm_build_diagonal / gingr_mesh_closest_points / gingr_model_new_reference refuse bad input with BAD_ARGUMENT."""
    import ctypes
    import gingr_amd as ga
    from gingr_amd import _native as nat
    from gingr_amd.api import ScalarKernelSpec, DevicePointDistributionModel, InterpolatedDevicePointDistributionModel
    ref = cloud(50, 30)

    def build(kx, ky, kz, tol=0.01):
        return DevicePointDistributionModel(ctx, ref, [], [], tol, 0, kernels=(kx, ky, kz)).rank

    g = ScalarKernelSpec("gauss", (40.0,), (5.0,))
    for bad in (ScalarKernelSpec("gauss", (40.0,), (5.0,), mirror=0.5), ScalarKernelSpec("gauss", (-1.0,), (5.0,)),
                ScalarKernelSpec("gauss", (), ()), ScalarKernelSpec("dot", scaling=0.0), ScalarKernelSpec("dot", scaling=float("nan"))):
        with pytest.raises(ga.GingrNativeError) as e:
            build(bad, g, g)
        assert e.value.code == nat.ERR_BAD_ARGUMENT
    with pytest.raises(ga.GingrNativeError):
        build(g, g, g, tol=1.5)
    k = nat.ScalarKernel()
    k.kind = nat.KERNEL_LOOKUP
    k.scaling = 1.0                                       # lookup table missing
    h = ctypes.c_void_p()
    rc = ctx._lib.gingr_gpmm_build_diagonal(ctx.handle, 50, nat.dptr(ref), ctypes.byref(k), ctypes.byref(k), ctypes.byref(k), 0.01, 0, 0, 0,
                                            ctypes.byref(h))
    assert rc == nat.ERR_BAD_ARGUMENT and not h.value
    # closest points: vertex id out of range
    V = cloud(10, 31)
    with pytest.raises(ga.GingrNativeError):
        ctx.mesh_closest_points(ref[:5], V, np.array([[0, 1, 10]]))
    # new reference: source ids out of range, and a row-sharded source
    model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.05).Gaussian(40.0, 5.0)
    ids = np.zeros((4, 3), dtype=np.int32)
    ids[2, 1] = 50
    with pytest.raises(ga.GingrNativeError):
        InterpolatedDevicePointDistributionModel(ctx, model, ref[:4], ids, np.tile([1.0, 0, 0], (4, 1))).rank
    shard = ga.DeviceModel(ctx, model, 0, 25)
    out = ctypes.c_void_p()
    ids[:] = 0
    w = np.tile([1.0, 0.0, 0.0], (4, 1))
    rc = ctx._lib.gingr_model_new_reference(ctx.handle, shard.handle, 4, nat.dptr(np.ascontiguousarray(ref[:4])), nat.iptr(ids), nat.dptr(w),
                                            0, 0, ctypes.byref(out))
    assert rc == nat.ERR_BAD_ARGUMENT and not out.value


def _regular_grid(n=9, size=40.0):
    xs = np.linspace(-size, size, n)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)


@pytest.mark.parametrize("max_rank", [15, 16, 17, 18, 21, 24, 30])
def test_exact_ties_follow_the_permuted_order_single_kernel(ctx, max_rank):
    """An exactly regular grid under a stationary kernel: symmetric points tie bit for bit, and scalismo's loop takes the first
    maximum in its PERMUTED index order (every step swaps the pivot to the front) -- a rank cut inside a group of tied pivots
    depends on that order (lowest-index tie breaking picks point 4 where the permuted order picks 36)."""
    import gingr_amd as ga
    ref = _regular_grid()
    mo = go.build_gpmm_mixture(ref, [30.0], [10.0], 0.0, max_rank)
    dm = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.0, maxRank=max_rank).Gaussian(30.0, 10.0)
    assert dm.rank == mo.rank == max_rank
    check_model_against_oracle(dm.to_host(), mo, tol_lam=1e-8, tol_cov=1e-8)


@pytest.mark.parametrize("max_rank,tol", [(14, 0.0), (23, 0.0), (0, 0.05)])
def test_exact_ties_follow_the_permuted_order_mirrored_kernel(ctx, max_rank, tol):
    """The mirrored kernel on a grid: 2 n points share every |x|, so the x coordinate's residuals tie in groups, and the y / z
    coordinates tie pairwise all the time."""
    import gingr_amd as ga
    rng = np.random.default_rng(2)
    ref = _regular_grid(8, 35.0)
    ref[:, 2] = rng.normal(0, 3.0, ref.shape[0])                     # heights break the y / z symmetry, not the |x| ties
    mo = go.build_gpmm_diagonal(ref, go.symmetric_gauss_kernel_fun(ref, 50.0, 20.0), tol, max_rank or None)
    dm = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=tol, maxRank=max_rank).GaussianSymmetry(50.0, 20.0)
    assert dm.rank == mo.rank, (dm.rank, mo.rank)
    check_model_against_oracle(dm.to_host(), mo, tol_lam=1e-8, tol_cov=1e-8)

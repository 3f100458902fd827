"""Structured inputs -- regular grids, queries exactly between grid points, duplicated points -- where distances tie BIT FOR BIT and
every discrete choice (closest point index, closest triangle, pivot) depends on the tie rule rather than on the data.  Random
clouds never exercise these rules; the pivoted-Cholesky tie rule of the GPMM builder was found this way."""
import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def grid_points(n, size, z=0.0):
    xs = np.linspace(-size, size, n)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), np.full(X.size, z)], axis=1)


def grid_mesh(n, size):
    v = grid_points(n, size)
    cells = []
    for i in range(n - 1):
        for j in range(n - 1):
            a, b, c, d = i * n + j, (i + 1) * n + j, i * n + j + 1, (i + 1) * n + j + 1
            cells += [[a, b, c], [b, d, c]]
    return v, np.array(cells, dtype=np.int32)


def test_nearest_neighbour_ties_on_a_grid(ctx):
    tgt = grid_points(40, 39.0)                                  # spacing 2
    xs = np.linspace(-38.0, 38.0, 39)                            # cell centres: equidistant to four grid points
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    centres = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    edges = centres + np.array([1.0, 0.0, 0.0])                  # edge midpoints: two candidates
    above = tgt[::7] + np.array([0.0, 0.0, 3.0])                 # straight above a grid point
    dup = np.concatenate([tgt, tgt[100:140]])                    # duplicated target points: lowest index wins
    for q, t in ((centres, tgt), (edges[:-39], tgt), (above, tgt), (centres, dup), (tgt[::3], dup)):
        idx, d2, mean = ctx.nn(q, t)
        oidx, od2, omean = go.icp_closest_point(q, t)
        assert np.array_equal(idx, oidx)
        assert np.array_equal(d2, od2) and abs(mean - omean) < 1e-13 * max(omean, 1.0)


def test_surface_closest_point_ties_on_a_triangulated_grid(ctx):
    v, c = grid_mesh(30, 29.0)                                   # spacing 2
    rng = np.random.default_rng(0)
    q = np.concatenate([v[::5] + np.array([0, 0, 2.5]),                          # above vertices: up to six triangles tie
                        0.5 * (v[c[::7, 0]] + v[c[::7, 1]]) + np.array([0, 0, 1.0]),   # above edges: two triangles tie
                        v[c[::9]].mean(axis=1) + np.array([0, 0, -1.5]),         # below triangle centres
                        rng.uniform(-29, 29, (200, 3)) * np.array([1, 1, 0.1])])
    cp, d2, tid, bary = ctx.mesh_closest_points(q, v, c)
    ocp, od2 = go.mesh_closest_point(q, v, c)
    assert np.abs(cp - ocp).max() < 1e-12 and np.abs(d2 - od2).max() < 1e-12
    # the reported triangle is the LOWEST of the tied ones (the oracle's argmin), and its weights rebuild the point
    A, B, C = v[c[:, 0]], v[c[:, 1]], v[c[:, 2]]
    for i in range(0, q.shape[0], 11):
        pts = go.closest_point_on_triangles(q[i], A, B, C)
        dist = ((pts - q[i]) ** 2).sum(1)
        assert tid[i] == int(np.argmin(dist)), i
    assert np.abs((bary[:, :, None] * v[c[tid]]).sum(1) - cp).max() < 1e-12


def test_icp_updates_on_grid_meshes_match_the_oracle(ctx):
    """Whole ICP updates (point-cloud and surface correspondence) with a grid template over a shifted grid target: ties in the
    closest point, in the closest triangle and in the nearest vertex of the rejection tests."""
    import gingr_amd as ga
    ref, cells = grid_mesh(16, 30.0)
    ref[:, 2] = 0.02 * (ref[:, 0] ** 2 - ref[:, 1] ** 2) / 30.0   # a saddle, symmetric under x <-> -x and y <-> -y
    tgt, tcells = grid_mesh(18, 34.0)
    tgt[:, 2] = 0.02 * (tgt[:, 0] ** 2 - tgt[:, 1] ** 2) / 30.0 + 1.0
    mo = go.build_gpmm_mixture(ref, [25.0], [4.0], 0.0, 18)
    model = ga.PointDistributionModel(reference=ref, mean=np.zeros_like(ref), basis=mo.U, variance=mo.lam, cells=cells)
    for method in ("PointcloudClosestPoint", "TriangularClosestPoint"):
        icp = ga.IcpRegistration(ctx)
        cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=4.0, endSigma=1.0, correspondenceMethod=method)
        st = icp.createInitialState(model, tgt, cfg, transform=ga.GlobalTranformationType.RigidTransforms, targetCells=tcells)
        ost = go.initial_state(mo, 4.0)
        for it in range(3):
            st = icp.update(st)
            if method == "PointcloudClosestPoint":
                ost, _ = go.icp_update(mo, tgt, ost, 4.0, 1.0, 10)
            else:
                ost, _ = go.icp_surface_update(mo, cells, tgt, tcells, ost, 4.0, 1.0, 10)
            assert st.general.status == ost.status == 0
            # The rejection tests of the surface flavour (opposite normals, self-intersection) sit on decision boundaries in a
            # symmetric geometry: from the second iteration on, rounding-level differences of the input flip single accept /
            # reject decisions -- the first iteration, from identical inputs, must agree exactly.
            tol = 1e-8 if (method == "PointcloudClosestPoint" or it == 0) else 1e-3
            assert np.abs(st.general.fit - ost.fit).max() < tol, (method, it, np.abs(st.general.fit - ost.fit).max())
            assert abs(st.general.sigma2 - ost.sigma2) < 1e-12
        icp.close()


def test_rigid_icp_on_a_grid(ctx):
    from gingr_amd import classic
    tgt = grid_points(25, 24.0)
    tpl = grid_points(25, 24.0) @ go.euler_to_rot(0.03, 0.0, 0.0).T + np.array([1.0, 1.0, 0.5])   # half a cell off: ties everywhere
    task = classic.ICPFactory(ctx, tpl).registerRigidly(tgt)
    fit = tpl
    for it in range(3):
        got, dist = task.Iteration()
        fit, wdist, _ = go.rigid_icp_iteration(fit, tgt)
        assert abs(dist - wdist) < 1e-12 * max(wdist, 1.0) and np.abs(got - fit).max() < 1e-9, it
    task.close()

"""The reference's optimal-step non-rigid ICP baselines (G/other/algorithms/icp/NonRigidOptimalStepICP.scala: N-ICP-T, N-ICP-A)
on the device against the oracle's restatement (dense stacked least squares through numpy's lstsq)."""
import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def sphere_mesh(n, seed, radius=30.0, noise=0.0):
    """closed triangle mesh: convex hull of n points on a sphere, outward orientation"""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(n, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    hull = ConvexHull(p)
    tris = hull.simplices.astype(np.int32).copy()
    a, b, c = p[tris[:, 0]], p[tris[:, 1]], p[tris[:, 2]]
    flip = np.einsum("ij,ij->i", np.cross(b - a, c - a), a) < 0
    tris[flip] = tris[flip][:, [0, 2, 1]]
    v = p * radius
    if noise:
        v = v * (1.0 + noise * rng.normal(size=(n, 1)))
    return v, tris


def pair(seed=0, n=220):
    tv, tt = sphere_mesh(n, seed)
    gv, gt = sphere_mesh(n + 37, seed + 1, radius=31.5, noise=0.01)
    gv = gv * np.array([1.05, 0.97, 1.02]) + np.array([0.8, -0.5, 0.3])
    lm_t = {"a": tv[3] + 0.1, "b": tv[n // 4] - 0.1, "c": tv[n // 2], "only_template": tv[7]}
    lm_g = {"a": gv[10], "b": gv[n // 3] + 0.05, "c": gv[n - 5], "only_target": gv[1]}
    return (tv, tt), (gv, gt), lm_t, lm_g


def oracle_landmarks(tv, gv, lm_t, lm_g):
    common = [k for k in lm_t if k in lm_g]
    return go.nicp_landmarks(tv, gv, np.array([lm_t[k] for k in common]), np.array([lm_g[k] for k in common]))


@pytest.mark.parametrize("kind", ["T", "A"])
def test_iterations_match_the_oracle(ctx, kind):
    from gingr_amd import classic
    (tv, tt), (gv, gt), lm_t, lm_g = pair()
    task = classic.NonRigidOptimalStepICP(ctx, (tv, tt), (gv, gt), lm_t, lm_g, gamma=0.7, kind=kind)
    ids, ul = oracle_landmarks(tv, gv, lm_t, lm_g)
    assert np.array_equal(task.lmIdsOnTemplate, ids) and np.array_equal(task.UL, ul)
    edges = go.nicp_edges(tt)
    assert np.array_equal(task.edges, edges)
    fit = tv
    for it, (alpha, beta) in enumerate([(10.0, 10.0), (4.0, 2.0), (1.0, 0.5)]):
        got, dist, lm = task.Iteration(fit, alpha, beta)
        cp, w, _ = task.getClosestPoints(fit)
        ocp, ow, odist = go.surface_correspondence(fit, tt, gv, gt)
        assert np.array_equal(w, ow) and np.abs(cp - ocp).max() < 1e-10 and abs(dist - odist) < 1e-12 * odist
        if kind == "T":
            want, wdist = go.nicp_iteration_t(fit, tt, gv, gt, edges, ids, ul, alpha, beta)
            wlm = want[ids]
        else:
            want, wdist, wlm = go.nicp_iteration_a(fit, tt, gv, gt, edges, ids, ul, alpha, beta, 0.7)
        err = np.abs(got - want).max()
        assert err < 1e-7, (kind, it, err)                 # normal equations against lstsq on the stacked system
        assert np.abs(lm - wlm).max() < 1e-7
        fit = want
    task.close()


def test_registration_loop_and_defaults(ctx):
    from gingr_amd import classic
    (tv, tt), (gv, gt), lm_t, lm_g = pair(seed=4, n=150)
    assert classic.NICP_DEFAULT_ALPHA == go.NICP_DEFAULT_ALPHA == [10.0] * 11
    task = classic.NonRigidOptimalStepICP_T(ctx, (tv, tt), (gv, gt), lm_t, lm_g)
    got = task.Registration(2, tolerance=0.001, alpha=[10.0, 2.0], beta=[10.0, 1.0])
    assert task.iterations == 4
    # the loop = the chained iterations (each of them is checked against the oracle above; a whole trajectory is not compared
    # with the oracle's: one rejection decision that flips on a 1e-9 difference moves a vertex by its whole data term)
    fit = tv
    for a, b in [(10.0, 10.0), (10.0, 10.0), (2.0, 1.0), (2.0, 1.0)]:
        fit = task.Iteration(fit, a, b)[0]
    assert np.array_equal(got, fit)
    common = [k for k in lm_t if k in lm_g]
    want = go.nicp_registration(tv, tt, gv, gt, np.array([lm_t[k] for k in common]), np.array([lm_g[k] for k in common]), "T", 1,
                                0.001, [10.0], [10.0])
    assert np.abs(task.Registration(1, alpha=[10.0], beta=[10.0]) - want).max() < 1e-7
    # a stage stops at once when the template already lies on the target (distance measured before the step)
    same = classic.NonRigidOptimalStepICP_T(ctx, (gv, gt), (gv, gt))
    out = same.Registration(5, tolerance=0.001, alpha=[10.0], beta=[10.0])
    assert same.iterations == 1 and np.abs(out - gv).max() < 1e-9
    same.close()
    task.close()


def test_bad_arguments(ctx):
    from gingr_amd import classic
    import gingr_amd as ga
    (tv, tt), (gv, gt), _, _ = pair(seed=2, n=80)
    with pytest.raises(ValueError):
        classic.NonRigidOptimalStepICP(ctx, (tv, tt), (gv, gt), gamma=-1.0)
    task = classic.NonRigidOptimalStepICP_A(ctx, (tv, tt), (gv, gt))
    with pytest.raises(ValueError):
        task.Iteration(tv, -1.0, 1.0)
    # an isolated vertex without weight: its row of the normal equations is zero -> reported, not hidden
    tv2 = np.concatenate([tv, [[100.0, 0.0, 0.0]]])
    w = np.ones(tv2.shape[0])
    w[-1] = 0.0
    out = np.empty_like(tv2)
    from gingr_amd._native import dptr, iptr
    rc = ctx._lib.gingr_nicp_solve(ctx.handle, 0, tv2.shape[0], dptr(tv2), task.edges.shape[0], iptr(task.edges), dptr(w), dptr(tv2), 0, None,
                                   None, 10.0, 1.0, 1.0, dptr(out), None)
    assert rc != 0 and b"positive definite" in ctx._lib.gingr_last_error(ctx.handle)
    bad_edges = np.array([[5, 3]], dtype=np.int32)           # not p1 < p2
    rc = ctx._lib.gingr_nicp_solve(ctx.handle, 0, tv.shape[0], dptr(tv), 1, iptr(bad_edges), dptr(np.ones(tv.shape[0])), dptr(tv), 0, None,
                                   None, 10.0, 1.0, 1.0, dptr(np.empty_like(tv)), None)
    assert rc != 0 and b"edge" in ctx._lib.gingr_last_error(ctx.handle)
    task.close()

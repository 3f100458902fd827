"""GPU tests of the probabilistic registration around the update map (BASELINE config 5): the
closest-point likelihood (IndependentPointDistanceEvaluator.scala:54-82), the accuracy metrics (RegistrationComparison.scala)
and the Metropolis-Hastings chain (GingrAlgorithm.scala:115-190) against the oracle's restatement, same random draws."""
import math
import os

import numpy as np
import pytest

from oracle import gingr_oracle as go
from .test_gpu_surface_icp import femur, grid_mesh, make_state, oracle_state_of, rel

pytestmark = pytest.mark.gpu


def close(a, b, tol=1e-10):
    return abs(a - b) <= tol * max(1.0, abs(b))


def check_stats(got, want, tol=1e-10):
    assert got[2] == want[2], (got, want)
    assert close(got[0], want[0], tol) and close(got[1], want[1], tol) and close(got[3], want[3], tol), (got, want)


def test_fitter_distance_stats_both_directions(ctx):
    ref, cells, target, tcells = femur()
    mo, algo, state = make_state(ctx, ref, cells, target, tcells, initial_pose=((0.02, -0.03, 0.01), (1.0, -2.0, 0.5)))
    state = algo.update(state)
    fit = np.asarray(state.general.fit)
    check_stats(algo.surfaceDistanceStats(state, 0, sdev=5.0), go.surface_distance_stats(fit, target, tcells, False, 5.0))
    check_stats(algo.surfaceDistanceStats(state, 1, sdev=2.0), go.surface_distance_stats(target, fit, cells, False, 2.0))
    # the first n' vertices of the sample / an explicit list of target points (numberOfPointsForComparison, :43-50)
    check_stats(algo.surfaceDistanceStats(state, 0, n_points=300, sdev=5.0), go.surface_distance_stats(fit[:300], target, tcells, False, 5.0))
    pts = target[np.random.default_rng(3).permutation(target.shape[0])[:211]] + 0.25
    check_stats(algo.surfaceDistanceStats(state, 1, points=pts, sdev=1.5), go.surface_distance_stats(pts, fit, cells, False, 1.5))
    # sdev = 0: no likelihood
    assert algo.surfaceDistanceStats(state, 0)[3] == 0.0
    algo.close()


def test_boundary_aware_stats_on_open_meshes(ctx):
    import gingr_amd as ga
    v1, t1 = grid_mesh(24, 30.0, 4.0, 1)
    v2, t2 = grid_mesh(30, 36.0, 5.0, 2)                      # larger sheet: the rim of v1 maps inside, the rim of v2 to v1's boundary
    v2 = v2 + np.array([0.5, -0.4, 1.5])
    for a, b, tb in ((v1, v2, t2), (v2, v1, t1)):
        want = go.surface_distance_stats(a, b, tb, True)
        got = ctx.mesh_distance_stats(a, b, tb, boundary_aware=True)
        check_stats(got, want)
        assert got[2] <= a.shape[0]
    assert ctx.mesh_distance_stats(v2, v1, t1, boundary_aware=True)[2] < v2.shape[0]   # some points were dropped
    # through the fitter (model = v1 sheet, target = v2 sheet)
    mo, algo, state = make_state(ctx, v1, t1, v2, t2, rank=12)
    fit = np.asarray(state.general.fit)
    check_stats(algo.surfaceDistanceStats(state, 0, boundary_aware=True), go.surface_distance_stats(fit, v2, t2, True))
    check_stats(algo.surfaceDistanceStats(state, 1, boundary_aware=True), go.surface_distance_stats(v2, fit, t1, True))
    algo.close()


def test_registration_comparison_metrics(ctx):
    import gingr_amd as ga
    ref, cells, target, tcells = femur()
    rec, gt = ga.TriangleMesh3D(ref[:], cells), ga.TriangleMesh3D(target, tcells)
    rc = ga.RegistrationComparison(ctx, verbose=False)
    avg, md, hd = rc.evaluateReconstruction2GroundTruth("femur", rec, gt)
    assert close(avg, go.avg_distance(ref, target, tcells))
    assert close(md, go.max_distance(ref, target, tcells))
    assert close(hd, go.hausdorff_distance(ref, cells, target, tcells))
    avg2, hd2 = rc.evaluateReconstruction2GroundTruthDouble("femur", rec, gt)
    assert close(avg2, (go.avg_distance(ref, target, tcells) + go.avg_distance(target, ref, cells)) / 2.0) and hd2 == hd
    a, m = rc.evaluateReconstruction2GroundTruthBoundaryAware("femur", rec, gt)   # closed meshes: nothing is dropped
    assert close(a, avg2) and close(m, hd)


def test_mesh_distance_argument_errors(ctx):
    import gingr_amd as ga
    v, t = grid_mesh(6, 5.0, 1.0, 0)
    bad = t.copy()
    bad[0, 0] = v.shape[0]
    with pytest.raises(ga.GingrNativeError):
        ctx.mesh_distance_stats(v, v, bad)
    with pytest.raises(ga.GingrNativeError):
        ctx.mesh_distance_stats(v, v, t, sdev=-1.0)


def test_evaluators(ctx):
    import gingr_amd as ga
    from gingr_amd import sampling as sp
    ref, cells, target, tcells = femur()
    mo, algo, state = make_state(ctx, ref, cells, target, tcells, rank=20)
    state = algo.update(state)
    fit = np.asarray(state.general.fit)
    for mode, name in ((sp.ModelToTargetEvaluation, "ModelToTarget"), (sp.TargetToModelEvaluation, "TargetToModel"),
                       (sp.SymmetricEvaluation, "Symmetric")):
        ev = sp.IndependentPointDistanceEvaluator(algo, state, 5.0, mode)
        want = go.independent_point_distance_logvalue(fit, cells, target, tcells, 5.0, name)
        assert close(ev.logValue(state), want), (name, ev.logValue(state), want)
    prod = sp.IndependentPoints(algo, state, 5.0).productEvaluator()
    want = go.model_evaluator_logvalue(state.general.modelParameters.shape) + \
        go.independent_point_distance_logvalue(fit, cells, target, tcells, 5.0)
    assert close(prod.logValue(state), want)
    assert sp.EvaluatorWrapper(False, sp.IndependentPoints(algo, state, 5.0)).logValue(state) == 0.0
    algo.close()


def test_random_walk_proposals(ctx):
    """Each stock proposal changes exactly its block of the parameters, re-instantiates the fit on the device and reports the
    Gaussian density of its own step; any other kind of change has density 0 under it."""
    from gingr_amd import sampling as sp
    v1, t1 = grid_mesh(14, 20.0, 3.0, 5)
    v2, t2 = grid_mesh(14, 20.0, 3.5, 6)
    mo, algo, s0 = make_state(ctx, v1, t1, v2, t2, rank=10)
    rnd = sp.Random(5)
    shape = sp.RandomShapeUpdateProposal(algo, 0.1, rnd)
    rot = sp.GaussianAxisRotationProposal(algo, 0.01, sp.PitchAxis, rnd)
    tr = sp.GaussianAxisTranslationProposal(algo, 0.1, 2, rnd)
    s1, s2, s3 = shape.propose(s0), rot.propose(s0), tr.propose(s0)
    for s in (s1, s2, s3):
        assert s.general.iteration == s0.general.iteration + 1
        st = oracle_state_of(s.general, 1)
        assert rel(s.general.fit, go.model_instance_shape_pose_scale(mo, st)) < 1e-12
    d = np.asarray(s1.general.modelParameters.shape) - np.asarray(s0.general.modelParameters.shape)
    assert close(shape.logTransitionProbability(s0, s1), float(go.gaussian_logpdf(d, 0.1).sum()))
    assert close(rot.logTransitionProbability(s0, s2), float(go.gaussian_logpdf(s2.general.modelParameters.rotation.theta, 0.01)))
    assert close(tr.logTransitionProbability(s0, s3), float(go.gaussian_logpdf(s3.general.modelParameters.translation[2], 0.1)))
    assert s2.general.modelParameters.rotation.phi == 0.0 and s3.general.modelParameters.translation[:2] == (0.0, 0.0)
    for g, s in ((shape, s2), (shape, s3), (rot, s1), (rot, s3), (tr, s1), (tr, s2)):
        assert g.logTransitionProbability(s0, s) == -math.inf
    mix = sp.Generator(algo, rnd).DefaultRandom()
    assert math.isfinite(mix.logTransitionProbability(s0, s1)) and math.isfinite(mix.logTransitionProbability(s0, s3))
    algo.close()


def _cpd_chain_setup(ctx, rank=12):
    import gingr_amd as ga
    v1, t1 = grid_mesh(16, 20.0, 3.0, 7)
    v2, t2 = grid_mesh(18, 22.0, 3.5, 8)
    v2 = v2 @ go.euler_to_rot(0.02, -0.015, 0.03).T + np.array([0.4, -0.3, 0.6])
    mo = go.build_gaussian_gpmm(v1, 25.0, 4.0, rel_tol=1e-9, max_rank=rank)
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam, cells=t1)
    return mo, model, v2, t1, t2


def test_metropolis_hastings_chain_matches_oracle(ctx):
    """Config-5 shaped run: CPD informed proposals mixed 50/50 with the stock random walks, prior x model-to-target likelihood;
    the same draws must give the same accept / reject sequence and the same states."""
    import gingr_amd as ga
    from gingr_amd import sampling as sp
    mo, model, target, cells, tcells = _cpd_chain_setup(ctx)
    algo = ga.CpdRegistration(ctx)
    cfg = ga.CpdConfiguration(maxIterations=26, w=0.05)
    s0 = algo.createInitialState(model, target, cfg, targetCells=tcells)
    chain_states, flags = [], []

    class Log:
        def accept(self, cur, prop, gen, ev):
            flags.append(True)

        def reject(self, cur, prop, gen, ev):
            flags.append(False)

    settings = sp.ProbabilisticSettings(sp.IndependentPoints(algo, s0, 1.0), randomMixture=0.5)
    best = algo.run(s0, callBackLogger=chain_states.append, acceptRejectLogger=Log(), probabilisticSettings=settings, rnd=sp.Random(42))
    flags = flags[1:]                                              # run() logs accept(initial, initial) first (:130)

    st0 = oracle_state_of(s0.general, 1)
    upd = lambda st, z: go.cpd_update(mo, target, st, w=0.05, z=z)
    def logq(f, t):
        try:
            pids, pts, var = go.cpd_observations(mo, target, f, 0.05)
            return go.posterior_logpdf_of_mesh(mo, f, pids, pts, var, f.fit)
        except np.linalg.LinAlgError:
            return -math.inf
    logv = lambda st: go.model_evaluator_logvalue(st.alpha) + go.independent_point_distance_logvalue(st.fit, cells, target, tcells, 1.0)
    obest, ostates, oflags = go.mh_run(mo, st0, 26, upd, logq, logv, 0.5, go.ChainRandom(42))
    assert len(chain_states) == len(ostates) == 26
    assert flags == oflags, (flags, oflags)
    assert any(flags) and not all(flags)
    for k, (s, o) in enumerate(zip(chain_states, ostates)):
        mp = s.general.modelParameters
        assert np.abs(np.asarray(mp.shape) - o.alpha).max() < 1e-7, k
        assert np.abs(np.asarray(mp.translation) - o.translation).max() < 1e-7 and np.abs(np.asarray(
            [mp.rotation.phi, mp.rotation.theta, mp.rotation.psi]) - np.asarray(o.euler)).max() < 1e-9, k
        assert rel(s.general.fit, o.fit) < 1e-7, k
    assert np.abs(np.asarray(best.general.modelParameters.shape) - obest.alpha).max() < 1e-7
    assert best.general.status == ga.FittingStatuses.MaxIteration
    algo.close()


def test_deterministic_run_through_the_chain_equals_plain_run(ctx):
    import gingr_amd as ga
    from gingr_amd import sampling as sp
    mo, model, target, cells, tcells = _cpd_chain_setup(ctx)
    cfg = ga.CpdConfiguration(maxIterations=12, w=0.0)
    a1, a2 = ga.CpdRegistration(ctx), ga.CpdRegistration(ctx)
    r1 = a1.run(a1.createInitialState(model, target, cfg))
    r2 = sp.run(a2, a2.createInitialState(model, target, cfg))
    assert r1.general.iteration == r2.general.iteration and r1.general.status == r2.general.status
    assert np.array_equal(np.asarray(r1.general.fit), np.asarray(r2.general.fit))
    assert r2.general.generatedBy == "Deterministic" or r2.general.generatedBy == a2.name
    a1.close(); a2.close()


def test_surface_icp_chain_runs_and_improves(ctx, tmp_path):
    """The reference's DemoICP configuration in small: surface ICP proposals + random walks; the best sample's posterior value
    is not below the initial one and the chain states stay finite."""
    import gingr_amd as ga
    from gingr_amd import sampling as sp
    ref, cells, target, tcells = femur()
    mo, algo, s0 = make_state(ctx, ref, cells, target, tcells, rank=20, sigma=(1.0, 1.0), iters=15)
    settings = sp.ProbabilisticSettings(sp.IndependentPoints(algo, s0, 5.0), randomMixture=0.5)
    ev = sp.EvaluatorWrapper(True, settings.evaluators)
    v0 = ev.logValue(s0)
    log = sp.JSONStateLogger(settings.evaluators, str(tmp_path / "targetFittingICP.json"))      # as SimpleRegistrator.run (:141)
    best = algo.run(s0, acceptRejectLogger=log, probabilisticSettings=settings, rnd=sp.Random(1))
    assert np.all(np.isfinite(best.general.fit))
    assert ev.logValue(best) >= v0
    assert log.totalSamples == 15 and log.log[0].status and log.log[0].index == 0       # accept(initial, initial) + 14 steps
    assert all(set(e.logvalue) == {"Prior", "Distance", "product"} for e in log.log)
    log.writeLog()
    back = ga.io.read_log(str(tmp_path / "targetFittingICP.json"))
    last = ga.io.parameters_of_log_entry(back, len(back) - 1)
    assert last.shape.shape[0] == s0.general.model.rank
    algo.close()


def test_tiny_and_degenerate_meshes(ctx):
    """Single triangles, triangle counts around the 64 / 256 tile sizes, and zero-area cells (their interior branch is 0/0 and never
    wins; their corners and edges still serve)."""
    rng = np.random.default_rng(0)
    for T, K in [(1, 1), (1, 70), (63, 5), (64, 64), (65, 200), (256, 257), (257, 300)]:
        v = rng.normal(0, 5, (3 * T, 3))
        t = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
        p = rng.normal(0, 8, (K, 3))
        check_stats(ctx.mesh_distance_stats(p, v, t, sdev=2.0), go.surface_distance_stats(p, v, t, False, 2.0))
        assert ctx.mesh_distance_stats(p, v, t, boundary_aware=True)[2] == 0            # every vertex of a soup is a boundary vertex
    v = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 1, 0], [0, 0, 0.0]])
    t = np.array([[0, 1, 2], [0, 1, 3], [0, 4, 3]], dtype=np.int32)
    p = rng.normal(0, 2, (50, 3))
    check_stats(ctx.mesh_distance_stats(p, v, t), go.surface_distance_stats(p, v, t))


def _icosphere(level):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v), np.asarray(f, dtype=np.int32)


def test_full_size_surface_scan_properties(ctx):
    """82k triangles / 41k vertices (the surface-ICP benchmark size), where the oracle is too slow: properties that do not depend
    on the size -- points of the surface are at distance 0, the surface is never farther than the nearest vertex, distances to a
    sphere-like mesh are bounded by the radial offset."""
    v, f = _icosphere(6)
    v = v * 50.0
    assert v.shape[0] == 40962 and f.shape[0] == 81920
    cen = (v[f[:, 0]] + v[f[:, 1]] + v[f[:, 2]]) / 3.0
    s, mx, n, _ = ctx.mesh_distance_stats(cen[::7], v, f)
    assert n == cen[::7].shape[0] and mx < 1e-10
    assert ctx.mesh_distance_stats(v[::5], v, f)[1] == 0.0
    rng = np.random.default_rng(1)
    p = v[rng.permutation(v.shape[0])[:20000]] * (1.0 + rng.uniform(-0.05, 0.08, (20000, 1)))
    s, mx, n, _ = ctx.mesh_distance_stats(p, v, f)
    idx, d2, _ = ctx.nn(p, v)
    assert s <= float(np.sqrt(d2).sum()) * (1 + 1e-12) and mx <= float(np.sqrt(d2).max()) * (1 + 1e-12)
    radial = np.abs(np.linalg.norm(p, axis=1) - 50.0)
    assert mx <= radial.max() + 50.0 * 2e-3               # the facets lie within 0.2 % of the sphere at this subdivision
    assert s >= radial.sum() - 20000 * 50.0 * 2e-3


def test_full_size_surface_scan_against_the_brute_force_checker(ctx):
    """82k triangles / 41k vertices (the surface-ICP benchmark size): the device scan (tile / quarter boxes, pair queue, tie rule)
    against the C checker's brute force over ALL triangles, for 4 000 sampled queries: same closest points, the same squared
    distances bit for bit where the winning triangle is the same, the same triangle unless two triangles are exactly as close."""
    from oracle import c_oracle as co
    v, f = _icosphere(6)
    rng = np.random.default_rng(9)
    v = v * 50.0 * (1.0 + 0.03 * np.sin(3.0 * v[:, :1]) * np.cos(2.0 * v[:, 1:2]))          # bumpy: not every facet equidistant
    n = 4000
    p = v[rng.permutation(v.shape[0])[:n]] * (1.0 + rng.uniform(-0.06, 0.08, (n, 1))) + rng.normal(0, 0.3, (n, 3))
    cp, d2, tri, bary = ctx.mesh_closest_points(p, v, f)
    ocp, od2, otri = co.mesh_closest_point(p, v, f)
    assert np.abs(cp - ocp).max() < 1e-10
    assert np.abs(d2 - od2).max() <= 1e-12 * od2.max()
    same = tri == otri
    assert same.mean() > 0.95 and np.array_equal(d2[same], od2[same])
    # a different triangle only where both are (numerically) as close: shared edge / vertex of the two
    assert np.all(np.abs(d2[~same] - od2[~same]) <= 1e-12 * (1.0 + od2[~same]))
    # barycentric weights reproduce the point on the reported triangle
    rec = (bary[:, :, None] * v[f[tri]]).sum(1)
    assert np.abs(rec - cp).max() < 1e-9

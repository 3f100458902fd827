"""`GingrInterface` / `SimpleRegistrator` (G/simple/GingrInterface.scala, G/api/registration/SimpleRegistrator.scala) -- the entry
points of every shipped demo -- on the femur pair: the wrappers add control flow around `GingrAlgorithm.run`, so the checks are
(i) equivalence with the direct path, (ii) the state surgery of `runDecimated` (model / target / fit swapped, parameters kept,
sigma2 re-initialised by the algorithm), (iii) `newReference(NearestNeighborInterpolator)` = a row gather by the exact closest
point, (iv) the coarse-to-fine schedule of examples/DemoMultiResolution.scala runs and improves the fit."""
import dataclasses
import json
import os

import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def femur(ctx):
    import gingr_amd as ga
    d = np.load(os.path.join(HERE, "golden", "inputs.npz"))
    m = np.load(os.path.join(HERE, "golden", "femur_mesh.npz"))
    ref, tgt = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
    model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).Gaussian(sigma=70.0, scaling=50.0).to_host()
    model.cells = m["femur_cells"]
    target = ga.TriangleMesh3D(tgt, m["femur_target_cells"])
    lm_m = [ga.io.Landmark(f"L{k}", p.astype(np.float64)) for k, p in enumerate(d["femur_lm"])]
    lm_t = [ga.io.Landmark(f"L{k}", p.astype(np.float64)) for k, p in enumerate(d["femur_target_lm"])]
    return model, target, lm_m, lm_t


def test_new_reference_is_a_row_gather_by_the_closest_old_point(ctx, femur):
    from gingr_amd.simple import cluster_decimate, new_reference_nearest_neighbor
    import gingr_amd as ga
    model, _, _, _ = femur
    rng = np.random.default_rng(3)
    new_ref = model.reference[rng.choice(model.numberOfPoints, 300, replace=False)] + rng.normal(0, 0.3, (300, 3))
    dm = new_reference_nearest_neighbor(ctx, model, new_ref)
    d2 = ((new_ref[:, None, :] - model.reference[None, :, :]) ** 2).sum(-1)
    idx = d2.argmin(1)
    assert np.array_equal(dm.reference, new_ref) and np.array_equal(dm.variance, model.variance) and dm.rank == model.rank
    assert np.array_equal(dm.mean, model.mean[idx])
    B = model.basis.reshape(model.numberOfPoints, 3, model.rank)
    assert np.allclose(dm.basis.reshape(300, 3, model.rank), B[idx], atol=1e-13)     # gathered in HBM as U sqrt(lambda)
    # hence: the displacement field of any instance is the full model's displacement at the closest old point
    a = rng.normal(0, 1, model.rank)
    full = ga.DeviceModel(ctx, model).instance(a) - model.reference
    part = ga.DeviceModel(ctx, dm).instance(a) - new_ref
    assert np.allclose(part, full[idx], atol=1e-11)
    dv, dc = cluster_decimate(model.reference, model.cells, 400)
    dm2 = new_reference_nearest_neighbor(ctx, model, dv, dc)
    assert np.array_equal(dm2.cells, dc)
    # a row shard of the transferred model holds the rows of the whole one
    part = ga.DeviceModel(ctx, dm, 100, 300).download()
    assert np.allclose(part.basis, dm.basis[300:900], atol=1e-15) and np.array_equal(part.mean, dm.mean[100:300])


def test_closest_points_with_triangle_and_barycentric_weights(ctx, femur):
    model, target, _, _ = femur
    rng = np.random.default_rng(4)
    V, C = target.points, target.cells
    P = np.concatenate([model.reference[::7] + rng.normal(0, 1.0, model.reference[::7].shape),
                        V[:40],                                                   # exactly on vertices
                        0.5 * (V[C[:40, 0]] + V[C[:40, 1]]),                      # exactly on edges
                        (V[C[:40, 0]] + V[C[:40, 1]] + V[C[:40, 2]]) / 3.0])      # inside triangles
    cp, d2, tid, bary = ctx.mesh_closest_points(P, V, C)
    want_cp, want_d2 = go.mesh_closest_point(P, V, C)
    assert np.allclose(cp, want_cp, atol=1e-10) and np.allclose(d2, want_d2, atol=1e-9)
    assert tid.min() >= 0 and tid.max() < C.shape[0]
    assert np.all(bary >= 0.0) and np.allclose(bary.sum(1), 1.0, atol=1e-14)
    rebuilt = (bary[:, :, None] * V[C[tid]]).sum(1)                               # the weights reproduce the closest point
    assert np.allclose(rebuilt, cp, atol=1e-10)
    n0 = model.reference[::7].shape[0]
    assert np.allclose(d2[n0:n0 + 120], 0.0, atol=1e-20)
    assert np.all(np.sort(bary[n0:n0 + 40], axis=1)[:, :2] == 0.0)                # vertex queries: one weight is 1


def test_triangle_mesh_interpolated_model_as_the_demo_loader_builds_it(ctx, femur):
    """DemoDatasetLoader.model (examples/DemoHelper/DemoDatasetLoader.scala:43-64): model on the decimated reference, then
    newReference(fullReference, TriangleMeshInterpolator3D())."""
    import gingr_amd as ga
    from gingr_amd.simple import cluster_decimate, new_reference_triangle_mesh
    model, target, _, _ = femur
    dv, dc = cluster_decimate(model.reference, model.cells, 500)
    dec = ga.GPMMTriangleMesh3D(ctx, dv, relativeTolerance=0.01, cells=dc).Gaussian(70.0, 50.0)
    full = new_reference_triangle_mesh(ctx, dec, model.reference, model.cells)
    assert full.rank == dec.rank and full.numberOfPoints == model.numberOfPoints
    assert np.array_equal(full.variance, dec.to_host().variance)
    cp, _, tid, bary = ctx.mesh_closest_points(model.reference, dv, dc)
    U = dec.to_host().basis.reshape(dv.shape[0], 3, dec.rank)
    want = (bary[:, :, None, None] * U[dc[tid]]).sum(1)                           # (M, 3, r)
    assert np.allclose(full.basis.reshape(-1, 3, dec.rank), want, atol=1e-13)
    assert not full.mean.any()
    # the decimated vertices are a subset of the full mesh: there the transferred model IS the decimated one
    lut = {tuple(p): i for i, p in enumerate(model.reference)}
    ids = np.array([lut[tuple(p)] for p in dv])
    assert np.allclose(full.basis.reshape(-1, 3, dec.rank)[ids], U, atol=1e-13)
    # and it registers: CPD against the femur target, full resolution
    cpd = ga.CpdRegistration(ctx)
    # (20 iterations: run on, sigma2 falls until some vertex has P1 = 0 -- the reference's NaN hazard)
    st = cpd.run(cpd.createInitialState(full, target.points, ga.CpdConfiguration(maxIterations=20, w=0.05)))
    cmp_ = ga.RegistrationComparison(ctx, verbose=False)
    d0 = cmp_.avgDistance(ga.TriangleMesh3D(model.reference, model.cells), target)
    d1 = cmp_.avgDistance(ga.TriangleMesh3D(st.general.fit, model.cells), target)
    assert st.general.status in (ga.FittingStatuses.Converged, ga.FittingStatuses.MaxIteration) and d1 < 0.5 * d0, (d0, d1)


def test_run_equals_the_direct_path_and_returns_the_full_resolution_fit(ctx, femur):
    import gingr_amd as ga
    model, target, lm_m, lm_t = femur
    cfg = ga.CpdConfiguration(maxIterations=15, w=0.1)
    reg = ga.GingrInterface(ctx, model, target, modelLandmarks=lm_m, targetLandmarks=lm_t, verbose=False).CPD(cfg)
    got = reg.run(globalTransformation=ga.GlobalTranformationType.RigidTransforms)
    cpd = ga.CpdRegistration(ctx)
    lms = ga.io.landmark_correspondences(model.reference, lm_m, lm_t)
    want = cpd.run(cpd.createInitialState(model, target.points, cfg, transform=ga.GlobalTranformationType.RigidTransforms,
                                          landmarks=lms, targetCells=target.cells))
    assert got.general.iteration == want.general.iteration == 14 and got.general.status == want.general.status
    assert got.general.sigma2 == want.general.sigma2
    assert np.array_equal(got.general.modelParameters.shape, want.general.modelParameters.shape)
    assert np.allclose(got.general.fit, want.general.fit, atol=1e-12)          # same model: the re-instantiated fit is the state's
    avg, mx = reg.lastComparison
    assert 0.0 < avg < mx < 50.0
    assert "MaxIterations" in got.general.statusText()


def test_initial_transform_enters_as_euler_angles_about_the_origin(ctx, femur):
    import gingr_amd as ga
    from gingr_amd.simple import TranslationAfterRotation
    model, target, _, _ = femur
    t0 = TranslationAfterRotation.fromEuler((50.0, -20.0, 5.0), 0.1, -0.2, 0.3)
    reg = ga.GingrInterface(ctx, model, target, initialModelParameterTransform=t0, verbose=False).ICP(
        ga.IcpConfiguration(maxIterations=3, initialSigma=2.0, endSigma=1.0))
    st = reg.createInitialState(model, target, ga.GlobalTranformationType.NoTransforms, t0)
    mp = st.general.modelParameters
    assert np.allclose([mp.rotation.phi, mp.rotation.theta, mp.rotation.psi], [0.1, -0.2, 0.3], atol=1e-13)
    assert tuple(mp.translation) == (50.0, -20.0, 5.0) and tuple(mp.center) == (0.0, 0.0, 0.0) and mp.scale == 1.0
    want = (model.reference + model.mean) @ go.euler_to_rot(0.1, -0.2, 0.3).T + np.array([50.0, -20.0, 5.0])
    assert np.allclose(st.general.fit, want, atol=1e-10)
    assert st.general.sigma2 == 2.0 and st.general.globalTransformation == ga.GlobalTranformationType.NoTransforms


def test_decimated_state_keeps_parameters_and_swaps_model_target_fit(ctx, femur):
    import gingr_amd as ga
    model, target, lm_m, lm_t = femur
    gi = ga.GingrInterface(ctx, model, target, modelLandmarks=lm_m, targetLandmarks=lm_t, verbose=False)
    coarse = gi.CPD(ga.CpdConfiguration(maxIterations=10)).runDecimated(modelPoints=100, targetPoints=120)
    g = coarse.general
    assert 100 <= g.model.numberOfPoints <= 130 and 120 <= g.target.shape[0] <= 155
    assert g.fit.shape == (model.numberOfPoints, 3)                         # the returned fit is on the FULL model (:151,157)
    assert g.landmarkCorrespondences is not None and g.landmarkCorrespondences.pids.max() < g.model.numberOfPoints
    # next stage from that state: parameters carried over, sigma2 from the new configuration, iteration / status cleared
    reg2 = gi.CPD(ga.CpdConfiguration(maxIterations=10, initialSigma=g.sigma2))
    init = reg2._decimateState(g, ga.GlobalTranformationType.RigidTransforms, 400, 400)
    assert init.iteration == 0 and init.status == ga.FittingStatuses.None_ and init.sigma2 == g.sigma2
    assert np.array_equal(init.modelParameters.shape, g.modelParameters.shape) and init.modelParameters.rotation == g.modelParameters.rotation
    assert 400 <= init.model.numberOfPoints <= 505 and init.fit.shape[0] == init.model.numberOfPoints
    mp = init.modelParameters
    mo = go.PDM(init.model.reference, init.model.mean, init.model.basis, init.model.variance)
    st = go.State(alpha=np.array(mp.shape), euler=(mp.rotation.phi, mp.rotation.theta, mp.rotation.psi), center=np.zeros(3),
                  translation=np.array(mp.translation), scale=mp.scale, sigma2=1.0, fit=None, iteration=0, status=0,
                  global_transformation=go.RIGID_TRANSFORMS, step_length=1.0)
    assert np.allclose(init.fit, go.model_instance_shape_pose_scale(mo, st), atol=1e-9)
    # without an initialSigma the CPD start value is recomputed from the PASSED state's model mean and target (combineStates runs
    # before the swap, SimpleRegistrator.scala:94-96 / CPD.scala:92-102)
    init3 = gi.CPD(ga.CpdConfiguration(maxIterations=10))._decimateState(g, ga.GlobalTranformationType.RigidTransforms, 400, 400)
    assert abs(init3.sigma2 - ctx.cpd_initial_sigma2(g.model.reference + g.model.mean, g.target)) < 1e-9 * init3.sigma2


def test_multi_resolution_schedule_of_the_demo(ctx, femur):
    """examples/DemoMultiResolution.scala:31-47 on the femur pair: CPD 100 -> CPD 500 (sigma2 carried over) -> ICP 1000."""
    import gingr_amd as ga
    model, target, _, _ = femur
    gi = ga.GingrInterface(ctx, model, target, verbose=False)
    cmp_ = ga.RegistrationComparison(ctx, verbose=False)
    mesh = lambda fit: ga.TriangleMesh3D(fit, model.cells)
    start = cmp_.avgDistance(mesh(model.reference + model.mean), target)
    coarse = gi.CPD(ga.CpdConfiguration(maxIterations=50)).runDecimated(100, 100, globalTransformation=ga.GlobalTranformationType.RigidTransforms)
    medium = gi.CPD(ga.CpdConfiguration(maxIterations=50, initialSigma=coarse.general.sigma2)).runDecimated(
        500, 500, generalState=coarse.general, globalTransformation=ga.GlobalTranformationType.RigidTransforms)
    fine = gi.ICP(ga.IcpConfiguration(maxIterations=100, initialSigma=2.0, endSigma=0.01)).runDecimated(
        1000, 1000, generalState=medium.general, globalTransformation=ga.GlobalTranformationType.NoTransforms)
    d = [cmp_.avgDistance(mesh(s.general.fit), target) for s in (coarse, medium, fine)]
    assert all(s.general.status in (ga.FittingStatuses.Converged, ga.FittingStatuses.MaxIteration) for s in (coarse, medium, fine))
    assert d[0] < start and d[2] < d[0] and d[2] < 1.0, (start, d)
    assert fine.general.iteration == 99 and fine.general.model.numberOfPoints >= 1000


def test_probabilistic_run_decimated_writes_the_log_and_returns_the_best_sample(ctx, femur, tmp_path):
    import gingr_amd as ga
    model, target, _, _ = femur
    log = tmp_path / "fit.json"
    gi = ga.GingrInterface(ctx, model, target, evaluatorUncertainty=5.0, evaluatedPoints=60, logFileFittingParameters=str(log),
                           rnd=ga.sampling.Random(7), verbose=False)
    best = gi.ICP(ga.IcpConfiguration(maxIterations=25, initialSigma=1.0, endSigma=1.0)).runDecimated(
        100, 100, globalTransformation=ga.GlobalTranformationType.NoTransforms, probabilistic=True)
    entries = json.load(open(log))
    assert len(entries) == 25 and entries[0]["status"] is True            # the initial state + 24 Metropolis-Hastings steps
    assert best.general.status in (ga.FittingStatuses.MaxIteration, ga.FittingStatuses.ModelFlexibilityError)
    assert best.general.fit.shape == (model.numberOfPoints, 3) and np.all(np.isfinite(best.general.fit))


def test_posterior_visualisation_flow_of_the_demo(ctx, femur, tmp_path):
    """examples/DemoPosteriorVisualizationFemur.scala: chain log -> samples -> shapes (device) -> per-vertex variance maps, with
    the SimpleLogger call-back scoring the fit while the chain runs."""
    import gingr_amd as ga
    from gingr_amd import helper
    model, target, _, _ = femur
    log = tmp_path / "chain.json"
    gi = ga.GingrInterface(ctx, model, target, evaluatorUncertainty=5.0, logFileFittingParameters=str(log), rnd=ga.sampling.Random(3),
                           verbose=False)
    cb = helper.SimpleLogger(ctx, printUpdateFrequency=20, verbose=False)
    best = gi.ICP(ga.IcpConfiguration(maxIterations=61, initialSigma=1.0, endSigma=1.0)).runDecimated(
        150, 150, globalTransformation=ga.GlobalTranformationType.NoTransforms, probabilistic=True, callback=cb)
    assert cb.counter == 61 and [c for c, _, _ in cb.history] == [20, 40, 60] and all(a > 0 for _, a, _ in cb.history)
    full = helper.loadLog(str(log))
    assert len(full) == 61
    samples = helper.samplesFromLog(full, takeEveryN=5, total=10000, burnIn=10)
    shapes = helper.logSamples2shapes(ctx, model, [e for e, _ in samples])
    assert len(shapes) == len(samples) == 11 and all(sh.shape == (model.numberOfPoints, 3) for sh in shapes)
    # a logged shape is the instance of its parameters: the best state of the log reproduces the returned fit
    bestp = helper.jsonFormatToModelFittingParameters(helper.getBestStateFromLog(full))
    assert np.allclose(np.asarray(bestp.shape), np.asarray(best.general.modelParameters.shape), atol=0)
    bshape = helper.logSamples2shapes(ctx, model, [helper.getBestStateFromLog(full)])[0]
    assert np.allclose(bshape, best.general.fit, atol=1e-9)
    tot = helper.computeDistanceMapFromMeshesTotal(shapes)
    nrm = helper.computeDistanceMapFromMeshesNormal(shapes, ga.TriangleMesh3D(bshape, model.cells))
    assert tot.shape == nrm.shape == (model.numberOfPoints,) and np.all(tot >= 0) and np.all(nrm <= tot + 1e-9) and tot.max() > 0


def test_resident_run_loop_equals_the_generic_loop(ctx, femur):
    """`run` without a call-back keeps the state on the device and reads back scalars only; with a call-back it goes through the
    per-iteration `update`.  Same states, same stopping iteration, same final status -- CPD (converges) and ICP (runs out)."""
    import gingr_amd as ga
    model, target, lm_m, lm_t = femur
    lms = ga.io.landmark_correspondences(model.reference, lm_m, lm_t)
    seen = []
    for algo, cfg in ((ga.CpdRegistration(ctx), ga.CpdConfiguration(maxIterations=60, threshold=1e-3, w=0.05)),
                      (ga.IcpRegistration(ctx), ga.IcpConfiguration(maxIterations=12, initialSigma=5.0, endSigma=1.0)),
                      (ga.CpdRegistration(ctx), ga.CpdConfiguration(maxIterations=2))):
        init = algo.createInitialState(model, target.points, cfg, landmarks=lms, targetCells=target.cells)
        fast = algo.run(init)
        seen.clear()
        slow = algo.run(init, callBackLogger=lambda s: seen.append(s.general.iteration))
        assert fast.general.iteration == slow.general.iteration == seen[-1] and fast.general.status == slow.general.status
        assert fast.general.sigma2 == slow.general.sigma2
        assert np.array_equal(fast.general.modelParameters.shape, slow.general.modelParameters.shape)
        assert np.array_equal(fast.general.fit, slow.general.fit)
        assert fast.general.modelParameters.rotation == slow.general.modelParameters.rotation
    # the CPD run above stopped on its threshold, well before maxIterations
    assert len(seen) == 2


def test_stop_threshold_freezes_the_state_the_rule_fired_on(ctx, femur):
    """gingr_fitter_set_stop_threshold: with a threshold the kernel that commits sigma2 marks the state whose update moved sigma2 by less,
    and every update enqueued behind it is a no-op; clearing the rule lets the state move again (CPD.scala:108-110 as the run's
    dropWhile applies it, GingrAlgorithm.scala:142-153)."""
    import ctypes
    import gingr_amd as ga
    from gingr_amd import _native as nat
    from gingr_amd.sharded import ShardedFitter
    model, target, _, _ = femur
    f = ShardedFitter(ctx, model, np.asarray(target.points))
    s2 = ctx.cpd_initial_sigma2(np.asarray(model.reference), np.asarray(target.points))
    lib, hit = f._lib, ctypes.c_int32(-1)
    # the trajectory one update at a time
    f.set_state(np.zeros(model.rank), s2)
    sig = [s2]
    for _ in range(12):
        f.update_cpd(0.05, 1.0, 1)
        sig.append(f.get_state()[1].sigma2)
    d = np.abs(np.diff(sig))
    k = 6
    thr = float(0.5 * (d[k - 1] + d[k])) if d[k] < d[k - 1] else float(d[k - 1] * 1.0000001)
    first = int(np.argmax(d < thr)) + 1          # the first update that moves sigma2 by less than thr
    assert 1 <= first < 12
    # all twelve in one call under the rule: the chain ends at `first`
    f.set_state(np.zeros(model.rank), s2)
    assert lib.gingr_fitter_set_stop_threshold(f.handle, thr) == 0
    f.update_cpd(0.05, 1.0, 12)
    a, sc, fit = f.get_state()
    assert lib.gingr_fitter_stop_rule_hit(f.handle, ctypes.byref(hit)) == 0 and hit.value == 1
    assert sc.iteration == first and sc.sigma2 == sig[first] and sc.status == 0
    f.update_cpd(0.05, 1.0, 2)                   # still stopped
    assert f.get_state()[1].iteration == first
    assert lib.gingr_fitter_set_stop_threshold(f.handle, -1.0) == 0     # rule off, mark cleared
    f.update_cpd(0.05, 1.0, 1)
    a2, sc2, _ = f.get_state()
    assert sc2.iteration == first + 1 and sc2.sigma2 == sig[first + 1]
    assert lib.gingr_fitter_stop_rule_hit(f.handle, ctypes.byref(hit)) == 0 and hit.value == 0
    f.close()


def test_resident_icp_run_stops_at_the_failed_state(ctx, femur):
    """The resident ICP loop enqueues every remaining update in one native call; a failure on the way must leave exactly the state the
    per-iteration loop stops at (GingrAlgorithm.scala:149-157): a failed fit is never touched again on the device."""
    import gingr_amd as ga
    model, target, _, _ = femur
    algo = ga.IcpRegistration(ctx)
    # an infinite step length makes the blended coefficients non-finite: the projection fails, which is a ModelFlexibilityError at
    # any iteration (GingrAlgorithm.scala:248-251) -- the first of the eleven enqueued updates fails, ten more follow it on the device
    cfg = ga.IcpConfiguration(maxIterations=12, initialSigma=5.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    init = algo.createInitialState(model, target.points, cfg, stepLength=float("inf"))
    fast = algo.run(init)
    seen = []
    slow = algo.run(init, callBackLogger=lambda s: seen.append((s.general.iteration, s.general.status)))
    assert slow.general.status == ga.FittingStatuses.ModelFlexibilityError and 1 < len(seen) < cfg.maxIterations, seen
    assert fast.general.status == slow.general.status and fast.general.iteration == slow.general.iteration
    assert fast.general.sigma2 == slow.general.sigma2
    assert np.array_equal(fast.general.modelParameters.shape, slow.general.modelParameters.shape)
    assert np.array_equal(fast.general.fit, slow.general.fit, equal_nan=True)
    algo.close()

"""Consumer of REAL reference output: JSON dumps written by jvm/tools/RefDump.scala (schema gingr-refdump-1) on a machine with a
JVM + GiNGR 1.0-RC1 + scalismo 1.0-RC1, dropped into tests/golden/reference/.  Every file is replayed
  * on the CPU against the oracle (this pins oracle/ -- the [SCALISMO] restatements included), and
  * on the GPU against the HIP path (`-m gpu`),
with the north-star tolerances (vertex positions <= 1e-5 relative, correspondence indices exact).

While the directory holds no *.json the tests SKIP with "parity unpinned": neither this container nor the GPU boxes have a JVM
(SURVEY.md section 8c), so no such file can be produced here.  Dropping one file flips the parity claim without a code change.

Vertex order: the dump carries the model's reference points in scalismo's own vertex order, so point ids agree by construction
(the STL de-duplication order of scalismo's reader never enters)."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import gingr_oracle as go

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "reference", "*.json")))
UNPINNED = ("parity unpinned: no reference dump under tests/golden/reference/ (jvm/tools/RefDump.scala needs a JVM with GiNGR / "
            "scalismo; none exists in this image)")
TRANSFORMS = {"NoTransforms": 0, "RigidTransforms": 1, "SimilarityTransforms": 2}


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def load(path):
    d = json.load(open(path))
    assert d["schema"] == "gingr-refdump-1", path
    m = d["model"]
    mo = go.PDM(np.array(m["reference"], dtype=np.float64), np.array(m["mean"], dtype=np.float64).reshape(-1, 3),
                np.array(m["basis"], dtype=np.float64), np.array(m["variance"], dtype=np.float64))
    target = np.array(d["target"]["points"], dtype=np.float64)
    lm = d.get("landmarks") or {}
    landmarks = None
    if lm.get("pids"):
        landmarks = go.Landmarks(np.array(lm["pids"], dtype=np.int64), np.array(lm["points"], dtype=np.float64),
                                 np.array(lm["covs"], dtype=np.float64).reshape(-1, 3, 3))
    return d, mo, target, landmarks


def initial_state(d, mo):
    i, c = d["initial"], d["config"]
    st = go.State(alpha=np.array(i["alpha"], dtype=np.float64), euler=tuple(i["euler"]), center=np.array(i["center"], dtype=np.float64),
                  translation=np.array(i["translation"], dtype=np.float64), scale=float(i["scale"]), sigma2=float(i["sigma2"]),
                  fit=np.array(i["fit"], dtype=np.float64), iteration=int(i["iteration"]),
                  global_transformation=TRANSFORMS[c["globalTransformation"]], step_length=float(c["stepLength"]))
    return st


def check_state(tag, got_fit, got_alpha, got_sigma2, got_euler, got_translation, want, surface=False):
    """north-star tolerances; `surface`: the surface correspondence's self-intersection rule compares a computed point with the
    vertex bit for bit, so a 1e-16 difference of the posed mesh flips a few per cent of the accept / reject decisions
     -- only an agreement at the level of those flips can be asked for there"""
    f = 100.0 if surface else 1.0
    assert rel(got_fit, np.array(want["fit"])) < 1e-5 * f, (tag, "fit", rel(got_fit, np.array(want["fit"])))
    assert abs(got_sigma2 - want["sigma2"]) < 1e-6 * abs(want["sigma2"]), (tag, "sigma2")
    assert rel(got_alpha, np.array(want["alpha"])) < 1e-3 * f, (tag, "alpha")
    if surface:      # pose and shape trade off against each other when a few observations flip: the fit is the meaningful quantity
        return
    assert np.allclose(got_euler, want["euler"], atol=1e-6), (tag, "euler", got_euler, want["euler"])
    assert np.allclose(got_translation, want["translation"], atol=1e-4), (tag, "translation")


def oracle_step(d, mo, target, landmarks, st, cells, tcells):
    c = d["config"]
    if c["algorithm"] == "cpd":
        return go.cpd_update(mo, target, st, w=c["w"], lam=c["lambda"], landmarks=landmarks if c["useLandmarks"] else None)
    if c["method"] == "PointcloudClosestPoint":
        return go.icp_update(mo, target, st, c["initialSigma"], c["endSigma"], c["maxIterations"],
                             landmarks if c["useLandmarks"] else None)[0]
    return go.icp_surface_update(mo, cells, target, tcells, st, c["initialSigma"], c["endSigma"], c["maxIterations"],
                                 landmarks if c["useLandmarks"] else None)[0]


@pytest.mark.skipif(not FILES, reason=UNPINNED)
@pytest.mark.parametrize("path", FILES or ["-"])
def test_oracle_replays_the_reference_dump(path):
    replay_oracle(path)


def replay_oracle(path):
    """CPU: the oracle, started from the dump's initial state, reproduces every dumped iteration -- and the correspondence-level
    quantities of each pre-update state (P1, correspondence points, closest-point ids, surface weights)."""
    d, mo, target, landmarks = load(path)
    cells = np.array(d["model"]["cells"], dtype=np.int32)
    tcells = np.array(d["target"]["cells"], dtype=np.int32)
    st = initial_state(d, mo)
    assert rel(go.model_instance_shape_pose_scale(mo, st), st.fit) < 1e-9, "initial fit is not modelInstanceShapePoseScale"
    for want in d["iterations"]:
        c = d["config"]
        if "P1" in want:                                    # CPD: statistics of the state the update starts from
            stats = go.cpd_stats_dense(st.fit, target, st.sigma2, c["w"])
            assert np.allclose(stats.P1, want["P1"], rtol=1e-8, atol=1e-300), "P1"
            yhat = st.fit + (stats.PX / stats.P1[:, None] - st.fit)
            assert rel(yhat, np.array(want["correspondence"])) < 1e-9, "CPD correspondence points"
        if "closest_ids" in want and c.get("method") == "PointcloudClosestPoint":
            idx, _, _ = go.icp_closest_point(st.fit, target)
            assert np.array_equal(idx, np.array(want["closest_ids"])), "closest-point ids must be bit-exact"
        if "surface_weights" in want and c.get("method") == "TriangularClosestPoint":
            _, w, _ = go.surface_correspondence(st.fit, cells, target, tcells)
            agree = float(np.mean(w == np.array(want["surface_weights"])))
            # the self-intersection rule compares a computed point with the vertex bit for bit: agreement RATE
            assert agree > 0.9, ("surface weights agree on", agree)
        st = oracle_step(d, mo, target, landmarks, st, cells, tcells)
        assert st.iteration == want["iteration"]
        check_state((os.path.basename(path), want["iteration"]), st.fit, st.alpha, st.sigma2, st.euler, st.translation, want,
                    surface=c.get("method") == "TriangularClosestPoint")


@pytest.mark.gpu
@pytest.mark.skipif(not FILES, reason=UNPINNED)
@pytest.mark.parametrize("path", FILES or ["-"])
def test_hip_path_replays_the_reference_dump(ctx, path):
    replay_hip(ctx, path)


def replay_hip(ctx, path):
    """GPU: the HIP path through the host mirror of the plugin API, every iteration restarted from the DUMPED pre-update state
    (so that a discontinuous accept / reject decision cannot fork the trajectory) and compared with the dumped result."""
    import gingr_amd as ga
    d, mo, target, landmarks = load(path)
    c = d["config"]
    cells = np.array(d["model"]["cells"], dtype=np.int32)
    tcells = np.array(d["target"]["cells"], dtype=np.int32)
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam, cells=cells if cells.shape[0] else None)
    lms = None
    if landmarks is not None and c["useLandmarks"]:
        lms = ga.LandmarkCorrespondences(landmarks.pids, landmarks.points, landmarks.covs)
    if c["algorithm"] == "cpd":
        algo = ga.CpdRegistration(ctx)
        cfg = ga.CpdConfiguration(maxIterations=100, w=c["w"], lambda_=c["lambda"], useLandmarkCorrespondence=bool(c["useLandmarks"]),
                                  initialSigma=d["initial"]["sigma2"])
        kw = {}
    else:
        algo = ga.IcpRegistration(ctx)
        cfg = ga.IcpConfiguration(maxIterations=c["maxIterations"], initialSigma=c["initialSigma"], endSigma=c["endSigma"],
                                  correspondenceMethod=c["method"], useLandmarkCorrespondence=bool(c["useLandmarks"]))
        kw = {"targetCells": tcells} if tcells.shape[0] else {}
    state = algo.createInitialState(model, target, cfg, transform=TRANSFORMS[c["globalTransformation"]], stepLength=c["stepLength"],
                                    landmarks=lms, **kw)
    prev = d["initial"]
    for want in d["iterations"]:
        import dataclasses
        mp = ga.ModelFittingParameters(scale=prev["scale"], translation=tuple(prev["translation"]), rotation=ga.EulerAngles(*prev["euler"]),
                                       center=tuple(prev["center"]), shape=np.array(prev["alpha"], dtype=np.float64))
        g = dataclasses.replace(state.general, modelParameters=mp, fit=np.array(prev["fit"], dtype=np.float64), sigma2=prev["sigma2"],
                                iteration=prev["iteration"], status=0)
        state = algo.update(state.updateGeneral(g))
        gg = state.general
        if "closest_ids" in want and c.get("method") == "PointcloudClosestPoint":
            assert np.array_equal(algo.last_correspondence_indices(), np.array(want["closest_ids"])), "closest-point ids must be bit-exact"
        rot = gg.modelParameters.rotation
        check_state((os.path.basename(path), want["iteration"]), gg.fit, gg.modelParameters.shape, gg.sigma2,
                    (rot.phi, rot.theta, rot.psi), gg.modelParameters.translation, want,
                    surface=c.get("method") == "TriangularClosestPoint")
        prev = want
    algo.close()


def hot_path_flavours(files):
    """Which flavours of the hot path the dumps present cover.  A PARTIAL dump is enough to pin the hot path: the two deterministic
    flavours without bit-sensitive accept / reject decisions -- CPD and ICP with the point-cloud correspondence
    (`scala-cli run jvm/tools/RefDump.scala -- <data> <out> --hot-path-only`) -- are replayed with the full north-star tolerances;
    the surface-ICP file only adds an agreement-rate check."""
    got = set()
    for f in files:
        c = json.load(open(f))["config"]
        got.add("cpd" if c["algorithm"] == "cpd" else ("icp_pointcloud" if c.get("method") == "PointcloudClosestPoint" else "icp_surface"))
    return got


def test_reference_dump_status():
    """Always runs: states in the test report whether parity is pinned, and for which flavours of the hot path."""
    if not FILES:
        pytest.skip(UNPINNED)
    assert all(json.load(open(f))["schema"] == "gingr-refdump-1" for f in FILES)
    got = hot_path_flavours(FILES)
    missing = {"cpd", "icp_pointcloud"} - got
    if missing:
        pytest.skip("parity pinned for %s only; hot path still unpinned for %s" % (sorted(got), sorted(missing)))


def test_partial_dump_is_enough_for_the_hot_path(tmp_path):
    """The consumer accepts any subset of the RefDump files: CPD + point-cloud ICP alone count as the hot path pinned."""
    a, b, c = (str(tmp_path / n) for n in ("femur_cpd_w01.json", "femur_icp_pointcloud.json", "femur_icp_surface.json"))
    _write_oracle_dump(a, "cpd")
    _write_oracle_dump(b, "icp_pointcloud")
    assert hot_path_flavours([a, b]) == {"cpd", "icp_pointcloud"}
    assert hot_path_flavours([a]) == {"cpd"}
    replay_oracle(a)
    replay_oracle(b)


# ------------------------------------------------------------------------------------------------ self-check of the consumer
def _write_oracle_dump(path, algorithm):
    """A dump in the RefDump schema produced by the ORACLE (into a temporary directory, never into tests/golden/reference/): it
    cannot pin anything, it only proves that the consumer above parses the schema and replays it -- so that the day a real file
    arrives, a failure means a numerical disagreement and not a broken harness."""
    inp = np.load(os.path.join(HERE, "golden", "inputs.npz"))
    mesh = np.load(os.path.join(HERE, "golden", "femur_mesh.npz"))
    ref, target = inp["femur"].astype(np.float64)[:400], inp["femur_target"].astype(np.float64)[:380]
    rng = np.random.default_rng(0)
    if algorithm == "icp_surface":
        ref, target = inp["femur"].astype(np.float64), inp["femur_target"].astype(np.float64)
        cells, tcells = mesh["femur_cells"].astype(np.int32), mesh["femur_target_cells"].astype(np.int32)
    else:
        cells, tcells = np.zeros((0, 3), np.int32), np.zeros((0, 3), np.int32)
    mo = go.build_gaussian_gpmm(ref, 70.0, 50.0, rel_tol=1e-9, max_rank=16)
    if algorithm == "cpd":
        cfg = {"algorithm": "cpd", "w": 0.1, "lambda": 1.0, "initialSigma": None, "globalTransformation": "RigidTransforms",
               "stepLength": 1.0, "useLandmarks": True}
        lm = go.Landmarks(np.array([3, 200]), target[[5, 150]] + rng.normal(0, 0.1, (2, 3)), np.tile(np.eye(3) * 2.0, (2, 1, 1)))
        st = go.initial_state(mo, go.cpd_initial_sigma2(mo.ref + mo.mean, target))
    else:
        cfg = {"algorithm": "icp", "initialSigma": 100.0, "endSigma": 1.0, "maxIterations": 10,
               "method": "PointcloudClosestPoint" if algorithm == "icp_pointcloud" else "TriangularClosestPoint",
               "globalTransformation": "RigidTransforms", "stepLength": 1.0, "useLandmarks": False}
        lm = None
        st = go.initial_state(mo, 100.0)

    def js(s, extra=None):
        o = {"iteration": s.iteration, "status": "None", "sigma2": s.sigma2, "alpha": s.alpha.tolist(), "euler": list(s.euler),
             "center": np.asarray(s.center).tolist(), "translation": np.asarray(s.translation).tolist(), "scale": s.scale,
             "fit": s.fit.tolist()}
        o.update(extra or {})
        return o
    d = {"schema": "gingr-refdump-1", "case": "self-check (ORACLE output, pins nothing)", "versions": "oracle", "config": cfg,
         "model": {"reference": mo.ref.tolist(), "cells": cells.tolist(), "mean": mo.mean.reshape(-1).tolist(),
                   "variance": mo.lam.tolist(), "basis": mo.U.tolist()},
         "target": {"points": target.tolist(), "cells": tcells.tolist()},
         "landmarks": {"pids": lm.pids.tolist(), "points": lm.points.tolist(), "covs": lm.covs.reshape(-1, 9).tolist()} if lm else
                      {"pids": [], "points": [], "covs": []},
         "initial": js(st), "iterations": []}
    for _ in range(2):
        extra = {}
        if algorithm == "cpd":
            stats = go.cpd_stats_dense(st.fit, target, st.sigma2, cfg["w"])
            extra = {"P1": stats.P1.tolist(), "correspondence": (st.fit + (stats.PX / stats.P1[:, None] - st.fit)).tolist()}
        elif algorithm == "icp_pointcloud":
            extra = {"closest_ids": go.icp_closest_point(st.fit, target)[0].tolist()}
        else:
            extra = {"surface_weights": go.surface_correspondence(st.fit, cells, target, tcells)[1].tolist()}
        st = oracle_step(d, mo, target, lm, st, cells, tcells)
        d["iterations"].append(js(st, extra))
    json.dump(d, open(path, "w"))


@pytest.mark.parametrize("algorithm", ["cpd", "icp_pointcloud"])
def test_consumer_self_check_cpu(tmp_path, algorithm):
    p = str(tmp_path / f"selfcheck_{algorithm}.json")
    _write_oracle_dump(p, algorithm)
    replay_oracle(p)


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", ["cpd", "icp_pointcloud", "icp_surface"])
def test_consumer_self_check_gpu(ctx, tmp_path, algorithm):
    p = str(tmp_path / f"selfcheck_{algorithm}.json")
    _write_oracle_dump(p, algorithm)
    replay_hip(ctx, p)

"""CPU tests of gingr_amd/io.py: the wire / file formats shared with the Scala host."""
import datetime
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(__file__)


def test_model_fitting_parameters_json_layout_and_round_trip(tmp_path):
    import gingr_amd as ga
    from gingr_amd import io
    p = ga.ModelFittingParameters(scale=1.25, translation=(1.0, -2.0, 3.5), rotation=ga.EulerAngles(0.1, -0.2, 0.3),
                                  center=(4.0, 5.0, 6.0), shape=np.array([0.5, -1.5, 2.0]))
    d = io.model_fitting_parameters_to_json(p)
    # spray-json jsonFormatN of the case classes (ModelFittingParameters.scala:31-57,98-103): field names are the constructor names
    assert d == {"scale": {"s": 1.25},
                 "pose": {"translation": [1.0, -2.0, 3.5],
                          "rotation": {"angles": {"phi": 0.1, "theta": -0.2, "psi": 0.3}, "center": [4.0, 5.0, 6.0]}},
                 "shape": {"parameters": [0.5, -1.5, 2.0]}}
    f = tmp_path / "pars.json"
    io.save_model_fitting_parameters(p, str(f))
    q = io.load_model_fitting_parameters(str(f))
    assert q.scale == p.scale and q.translation == p.translation and q.rotation == p.rotation and q.center == p.center
    assert np.array_equal(q.shape, p.shape)
    with pytest.raises(ValueError):
        io.model_fitting_parameters_from_json({"scale": {"s": 1.0}})


def test_json_state_log_entries(tmp_path):
    import gingr_amd as ga
    from gingr_amd import io
    mp = ga.ModelFittingParameters(scale=1.0, translation=(1.0, 2.0, 3.0), rotation=ga.EulerAngles(0.01, 0.02, 0.03),
                                   center=(0.0, 0.0, 0.0), shape=np.array([1.0, 2.0]))
    g = ga.GeneralRegistrationState(model=None, modelParameters=mp, target=np.zeros((1, 3)), fit=np.zeros((1, 3)), generatedBy="CPD")
    when = datetime.datetime(2024, 10, 8, 12, 30, 5)
    acc = io.log_entry(0, g, {"product": -3.5, "eval": -1.0}, True, when)
    rej = io.log_entry(1, g, {"product": -9.0}, False, when)
    assert acc.datetime == "2024-10-08 12:30:05" and acc.name == "CPD" and acc.status and acc.modelParameters == [1.0, 2.0]
    assert rej.modelParameters == [] and rej.translation == [] and rej.rotation == [] and rej.rotationCenter == [] and not rej.status
    f = tmp_path / "log.json"
    io.write_log([acc, rej], str(f))
    raw = json.load(open(f))
    assert set(raw[0]) == {"index", "name", "logvalue", "status", "modelParameters", "translation", "rotation", "rotationCenter",
                           "scaling", "datetime"}                                   # jsonLogFormat, JSONStateLogger.scala:36-47
    back = io.read_log(str(f))
    assert back == [acc, rej]
    p1 = io.parameters_of_log_entry(back, 1)                                         # rejected: the last accepted state
    assert np.array_equal(p1.shape, mp.shape) and p1.translation == mp.translation and p1.rotation == mp.rotation


def test_landmark_json_both_shipped_variants(tmp_path):
    from gingr_amd import io
    plain = [{"coordinates": [-39.5, 26.4, -207.7], "id": "L0"}, {"id": "L1", "coordinates": [1.0, 2.0, 3.0]}]
    unc = [{"id": "A", "coordinates": [0.0, 1.0, 2.0],
            "uncertainty": {"stddevs": [5.0, 5.0, 5.0], "pcvectors": [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]}}]
    f1, f2 = tmp_path / "a.json", tmp_path / "b.json"
    json.dump(plain, open(f1, "w"))
    json.dump(unc, open(f2, "w"))
    a, b = io.read_landmarks(str(f1)), io.read_landmarks(str(f2))
    assert [l.id for l in a] == ["L0", "L1"] and a[0].covariance is None and np.allclose(a[0].coordinates, [-39.5, 26.4, -207.7])
    assert np.allclose(b[0].covariance, 25.0 * np.eye(3))
    # anisotropic uncertainty survives a write / read cycle
    R, _ = np.linalg.qr(np.random.default_rng(0).normal(size=(3, 3)))
    cov = (R * np.array([9.0, 4.0, 1.0])) @ R.T
    io.write_landmarks([io.Landmark("X", np.array([1.0, 2.0, 3.0]), cov)], str(f1))
    assert np.allclose(io.read_landmarks(str(f1))[0].covariance, cov, atol=1e-12)
    # landmark triples of GeneralRegistrationState.apply (GeneralRegistrationState.scala:43-62): paired by id, pid = closest
    # reference vertex, point = the TARGET landmark, covariance = the MODEL landmark's uncertainty (:55-57) or the identity
    ref = np.array([[0.0, 0, 0], [10.0, 0, 0], [0.0, 10, 0]])
    lc = io.landmark_correspondences(ref, [io.Landmark("p", np.array([9.0, 1.0, 0.0]), cov), io.Landmark("q", np.zeros(3))],
                                     [io.Landmark("p", np.array([5.0, 5.0, 5.0]), 4.0 * np.eye(3))])
    assert list(lc.pids) == [1] and np.allclose(lc.points, [[5.0, 5.0, 5.0]]) and np.allclose(lc.covs[0], cov)
    lc = io.landmark_correspondences(ref, [io.Landmark("p", np.array([9.0, 1.0, 0.0]))],
                                     [io.Landmark("p", np.array([5.0, 5.0, 5.0]), 4.0 * np.eye(3))])
    assert np.allclose(lc.covs[0], np.eye(3))            # no model-side uncertainty: identity, whatever the target file says


def test_stl_reader_reproduces_the_golden_vertex_numbering(tmp_path):
    from gingr_amd import io
    d = np.load(os.path.join(HERE, "golden", "inputs.npz"))
    m = np.load(os.path.join(HERE, "golden", "femur_mesh.npz"))
    f = tmp_path / "femur.stl"
    io.write_stl(str(f), d["femur"], m["femur_cells"])
    v, c = io.read_stl(str(f))
    assert np.array_equal(v, d["femur"].astype(np.float64)) and np.array_equal(c, m["femur_cells"])
    ascii_stl = "solid t\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nendloop\nendfacet\n" \
                "facet normal 0 0 1\nouter loop\nvertex 1 0 0\nvertex 1 1 0\nvertex 0 1 0\nendloop\nendfacet\nendsolid t\n"
    g = tmp_path / "a.stl"
    g.write_text(ascii_stl)
    v, c = io.read_stl(str(g))
    assert v.shape == (4, 3) and np.array_equal(c, [[0, 1, 2], [1, 3, 2]])
    with pytest.raises(ValueError):
        (tmp_path / "bad.stl").write_text("hello")
        io.read_stl(str(tmp_path / "bad.stl"))


def test_ply_binary_ascii_and_point_cloud(tmp_path):
    from gingr_amd import io
    rng = np.random.default_rng(1)
    v = rng.normal(size=(7, 3)).astype(np.float32).astype(np.float64)
    c = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6]], dtype=np.int32)
    f = tmp_path / "m.ply"
    io.write_ply(str(f), v, c)
    v2, c2 = io.read_ply(str(f))
    assert np.array_equal(v2, v) and np.array_equal(c2, c)
    io.write_ply(str(f), v)
    v3, c3 = io.read_ply(str(f))
    assert np.array_equal(v3, v) and c3 is None
    ascii_ply = "ply\nformat ascii 1.0\ncomment x\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n" \
                "property uchar red\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n" \
                "0 0 0 255\n1 0 0 255\n0 1 0 255\n3 0 1 2\n"
    g = tmp_path / "a.ply"
    g.write_text(ascii_ply)
    v4, c4 = io.read_ply(str(g))
    assert np.array_equal(v4, [[0, 0, 0], [1, 0, 0], [0, 1, 0]]) and np.array_equal(c4, [[0, 1, 2]])


# ------------------------------------------------------------------------------------------ scalismo .h5.json model file
def test_statistical_model_h5json_round_trip(tmp_path):
    """StatisticalModelIO.write/readStatisticalTriangleMeshModel3D on `.h5.json` (DemoDatasetLoader.scala:47-53).
    [SCALISMO-RECALL]: the layout is restated from memory (gingr_amd/io.py), so this pins OUR reader to OUR writer and to the
    documented statismo tree -- not to a file produced by scalismo."""
    import json
    from gingr_amd import io as gio
    import gingr_amd as ga
    rng = np.random.default_rng(4)
    M, r = 57, 6
    ref = rng.normal(0, 30, (M, 3))
    U, _ = np.linalg.qr(rng.normal(size=(3 * M, r)))
    model = ga.PointDistributionModel(reference=ref, mean=rng.normal(0, 0.5, (M, 3)), basis=U, variance=np.sort(rng.uniform(1, 50, r))[::-1],
                                      cells=rng.integers(0, M, (40, 3)).astype(np.int32))
    p64, p32 = str(tmp_path / "m64.h5.json"), str(tmp_path / "m32.h5.json")
    gio.write_statistical_mesh_model(model, p64, dtype="float64")
    gio.write_statistical_mesh_model(model, p32)                                     # float32 like scalismo's statismo writer
    back = gio.read_statistical_mesh_model(p64)
    assert np.array_equal(back.reference, ref) and np.array_equal(back.cells, model.cells)
    assert np.allclose(back.mean, model.mean, atol=1e-12) and np.array_equal(back.basis, U) and np.array_equal(back.variance, model.variance)
    b32 = gio.read_statistical_mesh_model(p32)
    assert np.allclose(b32.reference, ref, rtol=1e-6) and np.allclose(b32.basis, U, atol=1e-7) and np.allclose(b32.variance, model.variance, rtol=1e-6)
    # the documented tree: paths resolve through hard links from the root; points are 3 x M, the mean is the mean SHAPE
    doc = json.load(open(p64))
    pts = gio._h5json_array(doc, "/representer/points")
    assert pts.shape == (3, M) and np.array_equal(pts.T, ref)
    assert np.allclose(gio._h5json_array(doc, "/model/mean").reshape(M, 3), ref + model.mean, atol=1e-12)
    assert gio._h5json_array(doc, "/model/pcaBasis").shape == (3 * M, r)
    assert int(gio._h5json_array(doc, "/version/minorVersion", dtype=np.int64)[0]) == 9
    rep = gio._h5json_resolve(doc, "/representer")
    assert {a["name"]: a["value"] for a in rep["attributes"]}["datasetType"] == "POLYGON_MESH"
    # a statismo 0.81 file (basis scaled by the standard deviations) is normalised on read
    for obj in doc["datasets"].values():
        if obj["alias"] == ["/version/minorVersion"]:
            obj["value"] = [81]
        if obj["alias"] == ["/model/pcaBasis"]:
            obj["value"] = (U * np.sqrt(model.variance)[None, :]).tolist()
    p81 = str(tmp_path / "m81.h5.json")
    json.dump(doc, open(p81, "w"))
    assert np.allclose(gio.read_statistical_mesh_model(p81).basis, U, atol=1e-12)
    with pytest.raises(ValueError):
        gio._h5json_array(doc, "/model/doesNotExist")

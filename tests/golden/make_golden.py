#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run from the repo root IN THE BUILD CONTAINER:

    python tests/golden/make_golden.py

Inputs: the reference's own demo data files under /root/reference/examples/data (femur STL pair + landmark JSON,
bunny PLY) -- data, not source.  Outputs are produced by the CPU oracle (oracle/gingr_oracle.py), NOT by the
reference: GiNGR is Scala/JVM and cannot be executed here, and it ships no golden vectors of its own
(src/test/scala/DummyTest.scala.scala:3 is `assert(1 > 0)`).  The fixtures therefore pin the oracle and the HIP path
against regressions and against each other ("parity unpinned" w.r.t. the real reference).

Vertex order of the STL meshes: first-occurrence de-duplication of the binary STL's triangle corners (scalismo's own
reader order is unknown, SURVEY.md section 8c "fixture caveat").
"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gingr_oracle as go  # noqa: E402

DATA = "/root/reference/examples/data"
OUT = os.path.dirname(os.path.abspath(__file__))


def read_binary_stl_vertices(path):
    raw = open(path, "rb").read()
    n = struct.unpack("<I", raw[80:84])[0]
    tri = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    corners = tri["v"].reshape(-1, 3)
    seen, order = {}, []
    for c in corners:
        key = c.tobytes()
        if key not in seen:
            seen[key] = len(order)
            order.append(c)
    return np.asarray(order, dtype=np.float32)


def read_ply_vertices(path):
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode("ascii", "replace")
    nv = int([l for l in header.splitlines() if l.startswith("element vertex")][0].split()[-1])
    return np.frombuffer(raw, dtype="<f4", count=3 * nv, offset=end).reshape(nv, 3).copy()


def read_landmarks(path):
    lm = json.load(open(path))
    return [l["id"] for l in lm], np.asarray([l["coordinates"] for l in lm], dtype=np.float64)


def main():
    femur = read_binary_stl_vertices(f"{DATA}/femur/femur.stl")
    femur_t = read_binary_stl_vertices(f"{DATA}/femur/femur_target.stl")
    ids_m, lm_m = read_landmarks(f"{DATA}/femur/femur.json")
    ids_t, lm_t = read_landmarks(f"{DATA}/femur/femur_target.json")
    assert ids_m == ids_t
    bunny = read_ply_vertices(f"{DATA}/bunny/bunny.ply")
    sub = np.sort(np.random.default_rng(7).choice(bunny.shape[0], 5000, replace=False))
    bunny5k = bunny[sub]
    np.savez_compressed(f"{OUT}/inputs.npz", femur=femur, femur_target=femur_t, femur_lm=lm_m, femur_target_lm=lm_t,
                        bunny5k=bunny5k)
    print("inputs:", femur.shape, femur_t.shape, bunny5k.shape)

    y, x = femur.astype(np.float64), femur_t.astype(np.float64)
    out = {}
    # (a) CPD statistics on the real femur pair
    s2_init = go.cpd_initial_sigma2(y, x)
    out["femur_sigma2_init"] = np.float64(s2_init)
    for tag, s2, w in [("s1_w01", 1.0, 0.1), ("sinit_w0", s2_init, 0.0), ("s25_w0", 25.0, 0.0)]:
        st = go.cpd_stats_dense(y, x, s2, w)
        out[f"cpd_{tag}_args"] = np.array([s2, w])
        out[f"cpd_{tag}_den"] = st.den
        out[f"cpd_{tag}_P1"] = st.P1
        out[f"cpd_{tag}_PX"] = st.PX
        out[f"cpd_{tag}_scalars"] = np.array([st.Np, st.sigma2_next])
    # (b) nearest neighbour: bunny5k as target, rigidly moved + perturbed copy as query (BASELINE config 2)
    rng = np.random.default_rng(70)
    R = go.euler_to_rot(0.02, -0.01, 0.03)
    q = bunny5k.astype(np.float64) @ R.T + np.array([0.3, -0.2, 0.1]) + rng.normal(0, 1.0, bunny5k.shape)
    q = q.astype(np.float32).astype(np.float64)
    idx, d2, md = go.icp_closest_point(q, bunny5k.astype(np.float64))
    out["nn_query"] = q.astype(np.float32)
    out["nn_idx"] = idx
    out["nn_mean_distance"] = np.float64(md)
    # (c) update trajectories on the femur reference with a rank-32 Gaussian GPMM (femur kernel sigma=70, s=50)
    mo = go.build_gaussian_gpmm(y, sigma=70.0, scaling=50.0, rel_tol=1e-9, max_rank=32)
    out["gpmm_basis"] = mo.U.astype(np.float64)
    out["gpmm_variance"] = mo.lam
    lms = go.landmark_correspondences(y, lm_m, lm_t)
    out["lm_pids"] = lms.pids
    for tag, kw in [("cpd_rigid", dict(w=0.0)), ("cpd_rigid_lm_w", dict(w=0.1, landmarks=lms))]:
        st = go.initial_state(mo, s2_init, global_transformation=go.RIGID_TRANSFORMS)
        for it in range(1, 6):
            st = go.cpd_update(mo, x, st, **kw)
            if it in (1, 2, 5):
                out[f"{tag}_it{it}_alpha"] = st.alpha
                out[f"{tag}_it{it}_pose"] = np.array([*st.euler, *st.translation, st.scale, st.sigma2, st.status])
        out[f"{tag}_it5_fit"] = st.fit
    st = go.initial_state(mo, 100.0, global_transformation=go.NO_TRANSFORMS)
    for it in range(1, 4):
        st, idx = go.icp_update(mo, x, st, 100.0, 1.0, 10)
    out["icp_it3_alpha"] = st.alpha
    out["icp_it3_idx"] = idx
    out["icp_it3_fit"] = st.fit
    out["icp_it3_sigma2"] = np.float64(st.sigma2)
    np.savez_compressed(f"{OUT}/expected.npz", **out)
    print("expected: %d arrays, %.1f KB" % (len(out), os.path.getsize(f"{OUT}/expected.npz") / 1024))


if __name__ == "__main__":
    main()

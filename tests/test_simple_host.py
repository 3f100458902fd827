"""Host-side pieces of gingr_amd.simple that need no GPU: the rotation convention helpers and the default decimator."""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_euler_round_trip_and_convention():
    from gingr_amd.simple import euler_to_rotation_matrix, rotation_matrix_to_euler
    from oracle import gingr_oracle as go
    rng = np.random.default_rng(0)
    for _ in range(50):
        phi, theta, psi = rng.uniform(-3.0, 3.0), rng.uniform(-1.5, 1.5), rng.uniform(-3.0, 3.0)
        R = euler_to_rotation_matrix(phi, theta, psi)
        assert np.allclose(R, go.euler_to_rot(phi, theta, psi), atol=1e-15)           # same convention as the oracle
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1.0) < 1e-14
        assert np.allclose(rotation_matrix_to_euler(R), (phi, theta, psi), atol=1e-12)
        assert np.allclose(rotation_matrix_to_euler(R), go.rot_to_euler(R), atol=1e-15)
    # gimbal branches (|R20| = 1): phi is fixed to 0, the rotation is reproduced
    for theta in (math.pi / 2.0, -math.pi / 2.0):
        R = euler_to_rotation_matrix(0.3, theta, -0.7)
        e = rotation_matrix_to_euler(R)
        assert e[0] == 0.0 and np.allclose(euler_to_rotation_matrix(*e), R, atol=1e-12)


def test_cluster_decimate_is_a_deterministic_subset_with_valid_triangles():
    from gingr_amd.simple import cluster_decimate
    d = np.load(os.path.join(HERE, "golden", "inputs.npz"))
    m = np.load(os.path.join(HERE, "golden", "femur_mesh.npz"))
    v, c = d["femur"].astype(np.float64), m["femur_cells"]
    for n in (100, 500, 1000):
        dv, dc = cluster_decimate(v, c, n)
        assert n <= dv.shape[0] <= int(1.25 * n) + 5, (n, dv.shape)
        # vertices are a subset of the input's, in the input's order
        lut = {tuple(p): i for i, p in enumerate(v)}
        ids = [lut[tuple(p)] for p in dv]
        assert ids == sorted(ids) and len(set(ids)) == len(ids)
        assert dc.dtype == np.int32 and dc.min() >= 0 and dc.max() < dv.shape[0]
        assert np.all(dc[:, 0] != dc[:, 1]) and np.all(dc[:, 1] != dc[:, 2]) and np.all(dc[:, 0] != dc[:, 2])
        assert np.unique(np.sort(dc, axis=1), axis=0).shape[0] == dc.shape[0]
        assert np.unique(dc).shape[0] >= 0.95 * dv.shape[0]                           # (almost) every vertex is used by a triangle
        dv2, dc2 = cluster_decimate(v, c, n)
        assert np.array_equal(dv, dv2) and np.array_equal(dc, dc2)
    same_v, same_c = cluster_decimate(v, c, 10 ** 6)
    assert np.array_equal(same_v, v) and np.array_equal(same_c, c)
    pv, pc = cluster_decimate(v, None, 200)                                           # point cloud
    assert pc is None and 200 <= pv.shape[0] <= 260

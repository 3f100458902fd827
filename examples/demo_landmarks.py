#!/usr/bin/env python3
"""The reference's examples/DemoLandmarks.scala on the femur pair (the armadillo meshes of the original are not in the tree; needs
an MI355X):    PYTHONPATH=. python examples/demo_landmarks.py

Deterministic CPD through `GingrInterface(...).CPD(cfg).runDecimated(100, 100)` without and with landmark correspondences (the six
femur landmarks L0..L5 become observations that override the CPD observation of their closest model vertex,
GingrAlgorithm.scala:281-302), from a start where the target is rotated away so that the landmarks matter."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (first: one HIP runtime per process)
import gingr_amd as ga  # noqa: E402
from gingr_amd.io import Landmark  # noqa: E402
from gingr_amd.simple import euler_to_rotation_matrix  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
d = np.load(os.path.join(HERE, "..", "tests", "golden", "inputs.npz"))
m = np.load(os.path.join(HERE, "..", "tests", "golden", "femur_mesh.npz"))
ref = d["femur"].astype(np.float64)
R, t = euler_to_rotation_matrix(2.2, 0.9, -1.4), np.array([30.0, -20.0, 25.0])
tgt = d["femur_target"].astype(np.float64) @ R.T + t
lm_model = [Landmark(f"L{i}", p) for i, p in enumerate(d["femur_lm"].astype(np.float64))]
lm_target = [Landmark(f"L{i}", p @ R.T + t) for i, p in enumerate(d["femur_target_lm"].astype(np.float64))]

ctx = ga.Context(0)
model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).Gaussian(sigma=70.0, scaling=50.0).to_host()
model.cells = m["femur_cells"]
target = ga.TriangleMesh3D(tgt, m["femur_target_cells"])
cfg = ga.CpdConfiguration(maxIterations=30)


def report(name, state):
    fit = np.asarray(state.general.fit)
    s, mx, n, _ = ctx.mesh_distance_stats(fit, tgt, m["femur_target_cells"])
    print(f"{name:14s}: average distance to the target surface {s / n:.3f} mm, max {mx:.3f} mm")


no_lm = ga.GingrInterface(ctx, model, target, evaluatorUncertainty=2.0).CPD(cfg).runDecimated(100, 100)
no_lm.general.printStatus()
report("NoLMs", no_lm)
with_lm = ga.GingrInterface(ctx, model, target, modelLandmarks=lm_model, targetLandmarks=lm_target, evaluatorUncertainty=2.0).CPD(cfg).runDecimated(100, 100)
with_lm.general.printStatus()
report("WithLMs", with_lm)

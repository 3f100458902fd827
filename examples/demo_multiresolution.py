#!/usr/bin/env python3
"""The reference's examples/DemoMultiResolution.scala through `GingrInterface.runDecimated` (needs an MI355X):

    PYTHONPATH=. python examples/demo_multiresolution.py

Coarse CPD (100 points) -> medium CPD (500 points, sigma2 carried over) -> fine ICP (1000 points, no global transform), every
stage starting from the previous stage's parameters, on the femur pair with a rigid offset of the target.  The coarse meshes come
from the package's deterministic vertex clustering (scalismo's decimation is not restated)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (first: one HIP runtime per process)
import gingr_amd as ga  # noqa: E402
from gingr_amd.simple import euler_to_rotation_matrix  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
d = np.load(os.path.join(HERE, "..", "tests", "golden", "inputs.npz"))
m = np.load(os.path.join(HERE, "..", "tests", "golden", "femur_mesh.npz"))
ref = d["femur"].astype(np.float64)
# rigidOffset = TranslationAfterRotation(Translation(50, 50, 50), Rotation(0.1, 0.1, 0.1, origin))   (DemoMultiResolution.scala:16)
tgt = d["femur_target"].astype(np.float64) @ euler_to_rotation_matrix(0.1, 0.1, 0.1).T + 50.0

ctx = ga.Context(0)
model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).Gaussian(sigma=70.0, scaling=50.0).to_host()
model.cells = m["femur_cells"]
target = ga.TriangleMesh3D(tgt, m["femur_target_cells"])
gi = ga.GingrInterface(ctx, model, target)
Rigid, NoT = ga.GlobalTranformationType.RigidTransforms, ga.GlobalTranformationType.NoTransforms

t0 = time.perf_counter()
coarse = gi.CPD(ga.CpdConfiguration(maxIterations=50)).runDecimated(100, 100, globalTransformation=Rigid)
coarse.general.printStatus()
medium = gi.CPD(ga.CpdConfiguration(maxIterations=50, initialSigma=coarse.general.sigma2)).runDecimated(
    500, 500, generalState=coarse.general, globalTransformation=Rigid)
medium.general.printStatus()
fine = gi.ICP(ga.IcpConfiguration(maxIterations=100, initialSigma=2.0, endSigma=0.01)).runDecimated(
    1000, 1000, generalState=medium.general, globalTransformation=NoT)
fine.general.printStatus()
print(f"three stages in {time.perf_counter() - t0:.2f} s")

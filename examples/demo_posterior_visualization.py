#!/usr/bin/env python3
"""The reference's examples/DemoPosteriorVisualizationFemur.scala (needs an MI355X):

    PYTHONPATH=. python examples/demo_posterior_visualization.py

A probabilistic ICP registration of the femur pair writes its chain log (DemoICP.scala's second half); the log is read back,
thinned after a burn-in phase, the logged samples are instantiated on the GPU and turned into the two per-vertex variance maps
the demo colours the mesh with (total variance, variance along the normal)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (first: one HIP runtime per process)
import gingr_amd as ga  # noqa: E402
from gingr_amd import helper  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
d = np.load(os.path.join(HERE, "..", "tests", "golden", "inputs.npz"))
m = np.load(os.path.join(HERE, "..", "tests", "golden", "femur_mesh.npz"))
ctx = ga.Context(0)
model = ga.GPMMTriangleMesh3D(ctx, d["femur"].astype(np.float64), relativeTolerance=0.01, cells=m["femur_cells"]).Gaussian(70.0, 50.0)
target = ga.TriangleMesh3D(d["femur_target"].astype(np.float64), m["femur_target_cells"])
log_file = "/tmp/targetFittingICP.json"

t0 = time.perf_counter()
gi = ga.GingrInterface(ctx, model, target, evaluatorUncertainty=5.0, logFileFittingParameters=log_file, verbose=False)
best = gi.ICP(ga.IcpConfiguration(maxIterations=1000, initialSigma=1.0, endSigma=1.0)).runDecimated(
    100, 100, globalTransformation=ga.GlobalTranformationType.NoTransforms, probabilistic=True)
best.general.printStatus()
print(f"probabilistic ICP, 1000 chain states: {time.perf_counter() - t0:.2f} s")

burn_in = 100
full = helper.loadLog(log_file)
samples = helper.samplesFromLog(full, takeEveryN=50, total=10000, burnIn=burn_in)
print(f"Number of samples from log: {len(samples)}/{len(full) - burn_in}")
shapes = helper.logSamples2shapes(ctx, model, [e for e, _ in samples])
best_shape = helper.logSamples2shapes(ctx, model, [helper.getBestStateFromLog(full)])[0]
total = helper.computeDistanceMapFromMeshesTotal(shapes)
normal = helper.computeDistanceMapFromMeshesNormal(shapes, ga.TriangleMesh3D(best_shape, model.cells))
print(f"posterior variance per vertex: total mean {total.mean():.4f} max {total.max():.4f} mm^2, along the normal mean "
      f"{normal.mean():.4f} max {normal.max():.4f} mm^2")

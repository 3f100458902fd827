#!/usr/bin/env python3
"""The reference's DemoCPD / DemoICP on the femur pair, through the Python host layer (needs an MI355X):

    PYTHONPATH=. python examples/demo_femur.py

Mirrors examples/DemoCPD.scala and examples/DemoICP.scala of the reference: a Gaussian GPMM over the femur reference
(kernel defaults of DemoDatasetLoader.scala:113-114, built in HBM), deterministic CPD, then deterministic ICP with the default
surface correspondence, landmarks from femur.json / femur_target.json.  Data: tests/golden (the reference's demo meshes).
"""
import os
import time

import numpy as np
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch  # noqa: F401  (first: one HIP runtime per process)

import gingr_amd as ga

HERE = os.path.dirname(os.path.abspath(__file__))
d = np.load(os.path.join(HERE, "..", "tests", "golden", "inputs.npz"))
m = np.load(os.path.join(HERE, "..", "tests", "golden", "femur_mesh.npz"))
ref, target = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
cells, tcells = m["femur_cells"], m["femur_target_cells"]

ctx = ga.Context(0)
t0 = time.perf_counter()
model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).Gaussian(sigma=70.0, scaling=50.0)
model.cells = cells
print(f"GPMM: rank {model.rank} over {model.numberOfPoints} vertices, built in {time.perf_counter() - t0:.3f} s")

# landmark correspondences (GeneralRegistrationState.apply): pid = closest reference vertex to the model landmark
lms = ga.io.landmark_correspondences(ref, [ga.io.Landmark(f"L{k}", p) for k, p in enumerate(d["femur_lm"])],
                                     [ga.io.Landmark(f"L{k}", p) for k, p in enumerate(d["femur_target_lm"])])


def mean_surface_distance(fit):
    idx, d2, _ = ctx.nn(fit, target)
    return float(np.sqrt(d2).mean())


cpd = ga.CpdRegistration(ctx)
state = cpd.createInitialState(model, target, ga.CpdConfiguration(maxIterations=100, w=0.0, threshold=1e-10),
                               transform=ga.GlobalTranformationType.RigidTransforms, landmarks=lms)
t0 = time.perf_counter()
best = cpd.run(state)
print(f"CPD : {best.general.iteration} iterations in {time.perf_counter() - t0:.3f} s, sigma2 {best.general.sigma2:.4f}, "
      f"mean vertex distance to the target {mean_surface_distance(best.general.fit):.3f} mm (start "
      f"{mean_surface_distance(state.general.fit):.3f})")

icp = ga.IcpRegistration(ctx)
cfg = ga.IcpConfiguration(maxIterations=100, initialSigma=10.0, endSigma=1.0, correspondenceMethod="TriangularClosestPoint")
state = icp.createInitialState(model, target, cfg, targetCells=tcells, landmarks=lms)
t0 = time.perf_counter()
best = icp.run(state)
cp, w = icp.surfaceCorrespondence(best)
print(f"ICP : {best.general.iteration} iterations in {time.perf_counter() - t0:.3f} s (surface correspondence, "
      f"{int(w.sum())} of {w.shape[0]} pairs accepted at the end), mean vertex distance {mean_surface_distance(best.general.fit):.3f} mm")
ga.io.save_model_fitting_parameters(best.general.modelParameters, "/tmp/femur_fit_parameters.json")
print("wrote /tmp/femur_fit_parameters.json (ModelFittingParameters JSON, loadable by the Scala host)")

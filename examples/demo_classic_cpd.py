#!/usr/bin/env python3
"""The reference's classic CPD baselines (src/main/scala/gingr/other/algorithms/CPDRegistration.scala) on the femur pair through the
Python host layer (needs an MI355X):    PYTHONPATH=. python examples/demo_classic_cpd.py"""
import os
import time

import numpy as np
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch  # noqa: F401  (first: one HIP runtime per process)

import gingr_amd as ga
from gingr_amd import classic as cl

here = os.path.dirname(os.path.abspath(__file__))
d = np.load(os.path.join(here, "..", "tests", "golden", "inputs.npz"))
m = np.load(os.path.join(here, "..", "tests", "golden", "femur_mesh.npz"))
ref, target = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
cells, tcells = m["femur_cells"], m["femur_target_cells"]
ctx = ga.Context(0)
rc = ga.RegistrationComparison(ctx, verbose=False)
gt = ga.TriangleMesh3D(target, tcells)
print(f"start: average distance to the target surface {rc.avgDistance(ga.TriangleMesh3D(ref, cells), gt):.3f} mm")
for name, Reg, kw in (("rigid", cl.RigidCPDRegistration, {}), ("affine", cl.AffineCPDRegistration, {}),
                      ("non-rigid", cl.NonRigidCPDRegistration, dict(lambda_=2.0, beta=30.0))):
    t0 = time.perf_counter()
    fit = Reg(ctx, ref, w=0.0, max_iterations=100, **kw).register(target)
    dt = time.perf_counter() - t0
    avg, hd = rc.evaluateReconstruction2GroundTruthDouble(name, ga.TriangleMesh3D(fit, cells), gt)
    print(f"{name:9s} CPD: {dt:6.3f} s, average surface distance (both ways) {avg:.3f} mm, Hausdorff {hd:.3f} mm")

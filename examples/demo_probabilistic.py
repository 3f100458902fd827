#!/usr/bin/env python3
"""The reference's DemoICP in probabilistic mode (examples/DemoICP.scala with `probabilistic = true`) through the Python host
layer (needs an MI355X):

    PYTHONPATH=. python examples/demo_probabilistic.py [chain steps]

Metropolis-Hastings over GiNGR updates: informed proposals = surface-ICP posterior samples, mixed 50/50 with the stock random
walks; evaluator = prior on the coefficients x independent point distances (uncertainty 5 mm); JSON accept/reject log in the
reference's layout; accuracy report like SimpleRegistrator.run (:153-156)."""
import os
import sys
import time

import numpy as np
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch  # noqa: F401  (first: one HIP runtime per process)

import gingr_amd as ga
from gingr_amd import sampling as sp

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
here = os.path.dirname(os.path.abspath(__file__))
d = np.load(os.path.join(here, "..", "tests", "golden", "inputs.npz"))
m = np.load(os.path.join(here, "..", "tests", "golden", "femur_mesh.npz"))
ref, target = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
cells, tcells = m["femur_cells"], m["femur_target_cells"]

ctx = ga.Context(0)
model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).Gaussian(sigma=70.0, scaling=50.0)
model.cells = cells
algo = ga.IcpRegistration(ctx)
cfg = ga.IcpConfiguration(maxIterations=steps, initialSigma=1.0, endSigma=1.0)          # DemoICP.scala:30
state = algo.createInitialState(model, target, cfg, targetCells=tcells)
evaluator = sp.IndependentPoints(algo, state, uncertainty=5.0)                           # evaluatorUncertainty = 5.0 (:20)
log = sp.JSONStateLogger(evaluator, "/tmp/targetFittingICP.json")
t0 = time.perf_counter()
best = algo.run(state, acceptRejectLogger=log, probabilisticSettings=sp.ProbabilisticSettings(evaluator, randomMixture=0.5),
                rnd=sp.Random(2024))
dt = time.perf_counter() - t0
print(f"{steps - 1} Metropolis-Hastings steps in {dt:.2f} s ({(steps - 1) / dt:.0f} steps/s)")
log.printAcceptInfo()
log.writeLog()
ev = sp.EvaluatorWrapper(True, evaluator)
print(f"log posterior value: initial {ev.logValue(state):.1f} -> best sample {ev.logValue(best):.1f}")
print("Final registration with full resolution meshes:")
ga.RegistrationComparison(ctx).evaluateReconstruction2GroundTruthBoundaryAware(
    "", ga.TriangleMesh3D(np.asarray(best.general.fit), cells), ga.TriangleMesh3D(target, tcells))

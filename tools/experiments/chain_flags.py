"""Accept / reject flags of the 60-step femur chain of tests/test_gpu_mh_step.py (points = 700 and 0), for comparing library builds:
GINGR_HIP_LIB=<other .so> python tools/experiments/chain_flags.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gingr_amd as ga
from gingr_amd import sampling as sp
from tests.test_gpu_surface_icp import femur, make_state
from tests.test_gpu_mh_step import _run
ctx = ga.Context(0)
ref, cells, target, tcells = femur()
for points in (700, 0):
    mo, algo, s0 = make_state(ctx, ref, cells, target, tcells, rank=24, sigma=(1.0, 1.0), iters=61)
    settings = sp.ProbabilisticSettings(sp.IndependentPoints(algo, s0, 5.0, modelPointCount=points or None), randomMixture=0.5, fusedSteps=True)
    best, states, log = _run(algo, s0, settings, 11)
    print("points", points, "".join(("A" if f else "r") if k == "ICP" else ("a" if f else ".") for f, k in zip(log.flags, log.kinds)))
    print("   shape[:4] of the last state", np.asarray(states[-1].general.modelParameters.shape)[:4])
    if len(sys.argv) > 1:
        np.save(f"{sys.argv[1]}_{points}.npy", np.stack([np.asarray(s.general.modelParameters.shape) for s in states]))
    algo.close()

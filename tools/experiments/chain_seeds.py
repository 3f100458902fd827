"""Accept / reject patterns of the femur chain of tests/test_gpu_mh_step.py over a few seeds (choosing a seed whose 60 steps hold
accepted AND rejected informed proposals with some margin)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gingr_amd as ga
from gingr_amd import sampling as sp
from tests.test_gpu_surface_icp import femur, make_state
from tests.test_gpu_mh_step import _run
ctx = ga.Context(0)
ref, cells, target, tcells = femur()
for seed in range(11, 19):
    for points in (700, 0):
        mo, algo, s0 = make_state(ctx, ref, cells, target, tcells, rank=24, sigma=(1.0, 1.0), iters=61)
        settings = sp.ProbabilisticSettings(sp.IndependentPoints(algo, s0, 5.0, modelPointCount=points or None), randomMixture=0.5, fusedSteps=True)
        best, states, log = _run(algo, s0, settings, seed)
        pat = "".join(("A" if f else "r") if k == "ICP" else ("a" if f else ".") for f, k in zip(log.flags, log.kinds))
        print("seed", seed, "points", points, pat, "rejected informed:", pat[1:].count("r"))
        algo.close()

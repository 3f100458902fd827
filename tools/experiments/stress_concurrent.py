"""Bit stability of single-fitter updates while OTHER processes keep the device busy: run several instances of this script at once
(`for i in 1 2 3; do python tools/experiments/stress_concurrent.py 3000 & done; wait`).  Alternates between five states per flavour
(CPD, point-cloud ICP, surface ICP on the femur pair), ranks on every Gram / solve path; every result must have the bits of the first
result for its state."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_group import _case, _group, _femur_case
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ranks = [int(a) for a in sys.argv[2:]] or [40, 100, 130, 150, 200, 256, 300]
tag = os.getpid()
total_bad = 0
for rank in ranks:
    mo, target = _case(rank=rank)
    g = _group([0], mo, target)
    rng = np.random.default_rng(rank)
    for flavour, params in ((0, (0.1, 1.0)), (1, (4.0, 1.0, 20))):
        states = [(rng.normal(0, 0.5, mo.rank), float(s2)) for s2 in (30.0, 12.0, 50.0, 20.0, 8.0)]
        first, bad = {}, 0
        for k in range(reps):
            j = int(rng.integers(0, len(states)))
            g.set_state(*states[j])
            g.update(flavour, params, 1)
            a, sc, fit = g.get_state()
            if j not in first:
                first[j] = (a.copy(), fit.copy())
            elif not (np.array_equal(a, first[j][0]) and np.array_equal(fit, first[j][1])):
                bad += 1
        total_bad += bad
        print(tag, "rank", rank, "flavour", flavour, "updates", reps, "differing", bad, flush=True)
    g.close()
mo, cells, target, tcells = _femur_case()
g = _group([0], mo, target, cells, tcells)
rng = np.random.default_rng(7)
states = [(rng.normal(0, 0.3, mo.rank), float(s2)) for s2 in (20.0, 8.0, 30.0, 12.0)]
first, bad = {}, 0
for k in range(reps):
    j = int(rng.integers(0, len(states)))
    g.set_state(*states[j])
    g.update(2, (20.0, 1.0, 30), 1)
    a, sc, fit = g.get_state()
    if j not in first:
        first[j] = (a.copy(), fit.copy())
    elif not (np.array_equal(a, first[j][0]) and np.array_equal(fit, first[j][1])):
        bad += 1
total_bad += bad
print(tag, "surface ICP femur rank", mo.rank, "updates", reps, "differing", bad, flush=True)
g.close()
print(tag, "TOTAL differing", total_bad)

"""Run-to-run bit stability of one CPD update (and one point-cloud ICP update) from the same state on ONE fitter: `reps` times
set_state + update; every result must have the bits of the first.  Ranks on every path (gram_tri / gram_wide, LDS / wide solve,
the fit passes with boxes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_group import _case, _group
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ranks = [int(a) for a in sys.argv[2:]] or [40, 100, 130, 150, 200, 256]
for rank in ranks:
    mo, target = _case(rank=rank, M=1203 if rank < 400 else 1500)
    g = _group([0] * int(os.environ.get("SHARDS", "1")), mo, target)
    for flavour, params, s2 in ((0, (0.1, 1.0), 30.0), (1, (4.0, 1.0, 20), 4.0)):
        first, bad, worst = None, 0, 0.0
        for k in range(reps):
            g.set_state(np.zeros(mo.rank), s2)
            g.update(flavour, params, 2)
            a, sc, fit = g.get_state()
            if first is None:
                first = (a.copy(), fit.copy())
            elif not (np.array_equal(a, first[0]) and np.array_equal(fit, first[1])):
                bad += 1
                worst = max(worst, float(np.abs(fit - first[1]).max() / np.abs(first[1]).max()))
        print("rank", rank, "flavour", flavour, "updates", reps, "differing", bad, "worst", worst, flush=True)
    g.close()

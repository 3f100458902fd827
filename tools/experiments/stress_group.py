"""Flake hunt: three logical shards against one at rank 150 (tests/test_gpu_group.py::...sample_and_logpdf[0-150]).  Groups are created
`outer` times; with each pair the state is set and two CPD updates compared `inner` times.  A mismatch that stays for all inner
repetitions of a pair = something went wrong at creation (model constants); one that comes and goes = the update path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_group import _case, _group, _devices, rel
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 150
mo, target = _case(rank=rank)
outer, inner = int(sys.argv[1]), int(sys.argv[2])
for o in range(outer):
    single = _group([0], mo, target)
    multi = _group(_devices(3), mo, target)
    errs = []
    for i in range(inner):
        for g in (single, multi):
            g.set_state(np.zeros(mo.rank), 30.0)
            g.update(0, (0.1, 1.0), 2)
        a0, sc0, fit0 = single.get_state()
        am, scm, fitm = multi.get_state()
        errs.append(rel(fitm, fit0))
    bad = [f"{e:.1e}" for e in errs if e > 1e-9]
    if bad:
        print("creation", o, "mismatches", len(bad), "of", inner, [f"{e:.1e}" for e in errs], flush=True)
    single.close()
    multi.close()
print("done", outer, "creations x", inner)

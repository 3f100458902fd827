"""Flake hunt 5: the surface flavours on three logical shards against one, alternating states: closest point on the surface and along
the normal, forward and reversed direction (femur pair, rank 24 and 130)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_group import _femur_case, _group, rel
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 800
for rank in (24, 130):
    mo, cells, target, tcells = _femur_case(rank=rank)
    for method in (0, 1):
        for reversed_ in (False, True):
            single = _group([0], mo, target, cells, tcells, method=method)
            multi = _group([0, 0, 0], mo, target, cells, tcells, method=method)
            for g in (single, multi):
                g.set_correspondence_direction(reversed_)
            rng = np.random.default_rng(3)
            states = [(rng.normal(0, 0.3, mo.rank), float(s2)) for s2 in (20.0, 8.0, 30.0, 12.0, 5.0)]
            want = []
            for a, s2 in states:
                single.set_state(a, s2)
                single.update(2, (20.0, 1.0, 30), 1)
                want.append(single.get_state()[2].copy())
            bad = []
            for k in range(reps):
                j = int(rng.integers(0, len(states)))
                multi.set_state(*states[j])
                multi.update(2, (20.0, 1.0, 30), 1)
                e = rel(multi.get_state()[2], want[j])
                if e > 1e-9:
                    bad.append((k, j, f"{e:.1e}"))
            print("rank", rank, "method", method, "reversed", reversed_, "updates", reps, "mismatches", len(bad), bad[:5], flush=True)
            single.close()
            multi.close()

"""Flake hunt 4: sampled proposals and log transition densities of three logical shards against the single shard's, alternating
between states (see stress_group2.py), CPD and point-cloud ICP flavours, ranks on every solve / density path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_group import _case, _group, rel
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ranks = [int(a) for a in sys.argv[2:]] or [40, 104, 130, 150, 200, 256, 300]
for rank in ranks:
    mo, target = _case(rank=rank)
    single = _group([0], mo, target)
    multi = _group([0, 0, 0], mo, target)
    rng = np.random.default_rng(2)
    for flavour, params in ((0, (0.1, 1.0)), (1, (4.0, 1.0, 20))):
        states = [(rng.normal(0, 0.5, mo.rank), float(s2), rng.standard_normal(mo.rank)) for s2 in (30.0, 12.0, 50.0, 20.0)]
        want = []
        for a, s2, z in states:
            single.set_state(a, s2)
            single.update(flavour, params, 1, z=z)
            fit1 = single.get_state()[2].copy()
            single.set_state(a, s2)
            want.append((fit1, single.posterior_logpdf(flavour, params, fit1)))
        bad = []
        for k in range(reps):
            j = int(rng.integers(0, len(states)))
            a, s2, z = states[j]
            multi.set_state(a, s2)
            multi.update(flavour, params, 1, z=z)
            e = rel(multi.get_state()[2], want[j][0])
            multi.set_state(a, s2)
            l = multi.posterior_logpdf(flavour, params, want[j][0])
            el = abs(l - want[j][1]) / abs(want[j][1])
            if e > 1e-9 or el > 1e-9:
                bad.append((k, j, f"{e:.1e}", f"{el:.1e}"))
        print("rank", rank, "flavour", flavour, "sampled updates + densities", reps, "mismatches", len(bad), bad[:5], flush=True)
    single.close()
    multi.close()

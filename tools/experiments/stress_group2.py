"""Flake hunt 2: three logical shards against one at rank R, ALTERNATING between different states, so that a shard that reads a peer's
exchange buffer too early (and finds the previous update's numbers there) shows: with identical repetitions stale data is invisible."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_group import _case, _group, _devices, rel
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 150
nsh = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mo, target = _case(rank=rank)
single = _group([0], mo, target)
multi = _group([0] * nsh, mo, target)
rng = np.random.default_rng(1)
states = [(rng.normal(0, 0.5, mo.rank), float(s2)) for s2 in (30.0, 12.0, 50.0, 20.0, 8.0)]
want = []
for a, s2 in states:
    single.set_state(a, s2)
    single.update(0, (0.1, 1.0), 1)
    want.append(single.get_state()[2].copy())
bad = []
for k in range(reps):
    j = int(rng.integers(0, len(states)))
    a, s2 = states[j]
    multi.set_state(a, s2)
    multi.update(0, (0.1, 1.0), 1)
    e = rel(multi.get_state()[2], want[j])
    if e > 1e-9:
        bad.append((k, j, f"{e:.1e}"))
print("rank", rank, "shards", nsh, "updates", reps, "mismatches", len(bad), bad[:8])

"""Flake hunt 3 (needs a temporary patch of group.hip / fitter.hip that logs per-shard checksums of the posterior coefficients, alpha,
fit and state after every update into /tmp/gingr_xchlog_<shard>_of_<n>.bin under GINGR_DEBUG_XCHLOG; not in the tree): which exchange segment of which shard deviates when an update of the
three-shard group differs from the single shard's."""
import os, sys
os.environ["GINGR_DEBUG_XCHLOG"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_group import _case, _group, rel
reps, rank, nsh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
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
single.close()
seq, bad = [], []
for k in range(reps):
    j = int(rng.integers(0, len(states)))
    a, s2 = states[j]
    multi.set_state(a, s2)
    multi.update(0, (0.1, 1.0), 1)
    e = rel(multi.get_state()[2], want[j])
    seq.append(j)
    if e > 1e-9:
        bad.append((k, j, f"{e:.1e}"))
multi.close()
print("mismatching updates", bad[:10])
logs = [np.fromfile(f"/tmp/gingr_xchlog_{q}_of_{nsh}.bin").reshape(16384, 4)[:reps] for q in range(nsh)]
seq = np.array(seq)
names = ["posterior coefficients", "alpha", "fit (own rows)", "DevState"]
for q in range(nsh):
    for which in range(4):
        v = logs[q][:, which]
        for j in range(len(states)):
            idx = np.flatnonzero(seq == j)
            vals = v[idx]
            ref = np.median(vals)
            dev = idx[vals != ref]
            if len(dev):
                print(f"shard {q} {names[which]}: state {j} deviates at updates {dev[:8].tolist()} by {[float(abs(v[d] - ref) / abs(ref)) for d in dev[:4]]}")

import sys, os, numpy as np
sys.path.insert(0, '.')
import gingr_amd as ga
GOLD = os.path.join('tests', 'golden')
d, e = np.load(os.path.join(GOLD, "inputs.npz")), np.load(os.path.join(GOLD, "expected.npz"))
target, query = d["bunny5k"].astype(np.float64), e["nn_query"].astype(np.float64)
ctx = ga.Context(0)
idx, d2, md = ctx.nn(query, target)
print("exact", bool(np.array_equal(idx, e["nn_idx"])))
ctx.timing_enable(True); ctx.timing_reset()
for _ in range(200): ctx.nn(query, target)
ms, k = ctx.timing_read(8)
print("timer8 us/call", ms / 200 * 1e3)

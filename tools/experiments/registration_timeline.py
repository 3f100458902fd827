"""Where a whole CPD registration at the metric size spends its time outside the update loop: model build, state creation
(target upload, spatial orders, allocations), the first update (lazy set-up), 100 steady updates, result read-back."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gingr_amd as ga
from bench import synth_clouds
n, rank = 50000, 100
y, x = synth_clouds(n)
ctx = ga.Context(0)
for rep in range(3):
    t = [time.perf_counter()]
    model = ga.GPMMTriangleMesh3D(ctx, y, relativeTolerance=0.0, maxRank=rank).Gaussian(70.0, 50.0)
    r = model.rank
    ctx.synchronize(); t.append(time.perf_counter())
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(model, x, ga.CpdConfiguration(maxIterations=100, w=0.1))
    ctx.synchronize(); t.append(time.perf_counter())
    state = algo.update(state)
    ctx.synchronize(); t.append(time.perf_counter())
    for _ in range(99):
        state = algo.update(state)
    ctx.synchronize(); t.append(time.perf_counter())
    fit = state.general.fit.points if hasattr(state.general.fit, "points") else np.asarray(state.general.fit)
    t.append(time.perf_counter())
    names = ["model build", "createInitialState", "first update", "99 updates", "fit read-back"]
    print(f"rep {rep}: " + ", ".join(f"{nm} {1e3 * (b - a):.2f} ms" for nm, a, b in zip(names, t[:-1], t[1:])), f"| total {1e3 * (t[-1] - t[0]):.1f} ms")
    del state, algo
    model.device().close()

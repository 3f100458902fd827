#!/usr/bin/env python3
"""Where the host side of ONE `update(state)` call goes (state pushed / pulled through the boundary every iteration): cProfile over
200 surface-ICP updates of the icosphere workload (tools/bench_icp_surface.py), or CPD on the femur-sized synthetic clouds.
    python3 tools/prof_update_host.py [surface|cpd]"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import gingr_amd as ga

mode = sys.argv[1] if len(sys.argv) > 1 else "surface"
ctx = ga.Context(0)
if mode == "surface":
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_icp_surface.py")).read().split("OPTS = dict")[0]
    ns = {"__file__": __file__, "__name__": "ico"}
    sys.argv = sys.argv[:1]
    exec(compile(src, "ico", "exec"), ns)
    verts, cells = ns["icosphere"](6)
    ref = verts * 80.0
    bump = 1.0 + 0.08 * np.sin(3 * verts[:, 0]) * np.cos(2 * verts[:, 1]) + 0.05 * np.sin(5 * verts[:, 2])
    c, s = np.cos(0.05), np.sin(0.05)
    target = (ref * bump[:, None]) @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]).T + np.array([1.5, -1.0, 0.5])
    model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.0, maxRank=100).Gaussian(40.0, 10.0)
    model.cells = cells
    algo = ga.IcpRegistration(ctx)
    cfg = ga.IcpConfiguration(maxIterations=1000, initialSigma=10.0, endSigma=1.0, correspondenceMethod="TriangularClosestPoint")
    state = algo.createInitialState(model, target, cfg, targetCells=cells)
else:
    from bench import synth_clouds
    y, x = synth_clouds(1622)
    model = ga.GPMMTriangleMesh3D(ctx, y, relativeTolerance=0.0, maxRank=100).Gaussian(70.0, 50.0)
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(model, x, ga.CpdConfiguration(maxIterations=1000, w=0.1))
for _ in range(5):
    state = algo.update(state)
ctx.synchronize()
n = 200 if mode == "surface" else 40
t0 = time.perf_counter()
s = state
for _ in range(n):
    s = algo.update(s)
ctx.synchronize()
print("plain: %.1f us per update" % ((time.perf_counter() - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
s = state
for _ in range(n):
    s = algo.update(s)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
print(out.getvalue()[:6000])

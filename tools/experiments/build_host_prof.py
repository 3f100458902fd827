"""Host side of the on-device model build at the metric size (50k points, rank 100): wall time per build and, under
rocprofv3 --hip-trace --stats, which HIP calls it spends it in."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gingr_amd as ga
ctx = ga.Context(0)
ref = np.random.default_rng(1234).normal(0, 100, (50000, 3))
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for rep in range(4):
    t0 = time.perf_counter()
    g = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.0, maxRank=rank)
    t1 = time.perf_counter()
    dm = g.Gaussian(70.0, 50.0)
    r = dm.rank  # the build happens here
    ctx.synchronize()
    t2 = time.perf_counter()
    dm.device().close()
    print(f"rep {rep}: helper {1e3 * (t1 - t0):.2f} ms, build {1e3 * (t2 - t1):.2f} ms")

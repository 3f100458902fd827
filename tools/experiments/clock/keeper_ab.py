"""Does a row shard's iteration run faster when the chip never looks idle between its wide kernels?  (tools/experiments/clock/keeper.hip)
Rank 0 of `world` ranks of the metric workload (one-rank RCCL communicator, as bench.py --emulate-world): ms per iteration without and
with keeper launches on a second stream.   python3 keeper_ab.py [world] [points]"""
import ctypes, os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import torch  # noqa
import gingr_amd as ga
from gingr_amd.sharded import ShardedFitter
from bench import synth_clouds

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
points = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
y, x = synth_clouds(points)
ctx = ga.Context(0)
model = ga.GPMMTriangleMesh3D(ctx, y, relativeTolerance=0.0, maxRank=100).Gaussian(70.0, 50.0)
if world > 1:
    ctx.rccl_init(ctx.rccl_unique_id(), 1, 0)
    fitter = ShardedFitter(ctx, model, x, rank=0, world=world, all_reduce=None, rccl=True)
else:
    fitter = ShardedFitter(ctx, model, x)
s2 = ctx.cpd_initial_sigma2(y, x)
K = ctypes.CDLL(os.path.join(HERE, "libkeeper.so"))
K.keeper_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]
assert K.keeper_init() == 0
N = 100


def run(keeper):
    fitter.set_state(np.zeros(100), s2)
    fitter.update_cpd(0.1, 1.0, 10)
    ctx.synchronize()
    if keeper:
        blocks, threads, usec, mode = keeper
        est_ms = 0.45 * N * (8.0 / world if world > 1 else 5.5)
        K.keeper_launch(blocks, threads, usec, mode, int(est_ms * 1e3 / usec) + 2)
    t0 = time.perf_counter()
    fitter.update_cpd(0.1, 1.0, N)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    K.keeper_sync()
    return dt


for rep in range(2):
    print("no keeper                      %.4f ms" % run(None))
    for blocks, threads in ((256, 256), (256, 64), (1024, 64), (32, 64)):
        for mode in (0, 1):
            print("keeper %4d x %3d %s 500 us  %.4f ms" % (blocks, threads, "sleep" if mode == 0 else "fma  ", run((blocks, threads, 500.0, mode))))
    print("no keeper                      %.4f ms" % run(None))

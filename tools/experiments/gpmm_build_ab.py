"""Model builds of two libraries bit for bit (the blocked pc_gram / lv kernels of round 6 against the one-entry kernels before):
    python tools/experiments/gpmm_build_ab.py dump OUT.npz        (with GINGR_HIP_LIB pointing at the library to use)
    python tools/experiments/gpmm_build_ab.py compare A.npz B.npz"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if sys.argv[1] == "compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    worst = 0
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k])
        if not same:
            worst += 1
            d = np.abs(a[k] - b[k]).max() if a[k].shape == b[k].shape else "shape"
            print("DIFFERENT", k, a[k].shape, b[k].shape, d)
    print(f"{len(a.files)} arrays compared, {worst} different")
    sys.exit(1 if worst else 0)
import torch  # noqa: F401
import gingr_amd as ga
ctx = ga.Context(0)
out = {}
def keep(name, dm, dt):
    h = dm.to_host()
    out[name + "_basis"], out[name + "_variance"] = np.asarray(h.basis), np.asarray(h.variance)
    print(name, "rank", dm.rank, f"{dt * 1e3:.2f} ms", file=sys.stderr)
    dm.device().close()
def timed(make):
    dm = make(); dm.rank; ctx.synchronize(); dm.device().close()      # (the build is lazy: asking for the rank runs it)
    t0 = time.perf_counter(); dm = make(); dm.rank; ctx.synchronize()
    return dm, time.perf_counter() - t0
ref = np.random.default_rng(1234).normal(0, 100, (50000, 3))
for name, tol, mr in [("g50k_r100", 0.0, 100), ("g50k_r512", 0.01, 0), ("g50k_r300", 0.0, 300)]:     # (a new factory per build: models are memoised)
    keep(name, *timed(lambda: ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=tol, maxRank=mr).Gaussian(70.0, 50.0)))
keep("sym50k_r100", *timed(lambda: ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.0, maxRank=100).GaussianSymmetry(70.0, 50.0)))
keep("sym20k_r301", *timed(lambda: ga.GPMMTriangleMesh3D(ctx, ref[:20000], relativeTolerance=0.0, maxRank=301).GaussianSymmetry(70.0, 50.0)))
for name, npts, mr in [("sym8k_r193", 8000, 193), ("sym10k_r200", 10000, 200), ("sym20k_r512", 20000, 512)]:   # more than 192 columns
    keep(name, *timed(lambda: ga.GPMMTriangleMesh3D(ctx, ref[:npts], relativeTolerance=0.0, maxRank=mr).GaussianSymmetry(70.0, 50.0)))
d = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
m = np.load(os.path.join(ROOT, "tests", "golden", "femur_mesh.npz"))
fem, cells = d["femur"].astype(np.float64), m["femur_cells"]
g = lambda: ga.GPMMTriangleMesh3D(ctx, fem, relativeTolerance=0.01, maxRank=300, cells=cells)
keep("femur_gauss", *timed(lambda: g().Gaussian(70.0, 50.0)))
keep("femur_dot", *timed(lambda: g().GaussianDot(70.0, 0.05)))
keep("femur_sym", *timed(lambda: g().GaussianSymmetry(70.0, 50.0)))
np.savez(sys.argv[2], **out)

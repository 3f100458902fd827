#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: a Metropolis-Hastings chain over GiNGR updates on the femur pair, set up like the reference's
DemoICP (examples/DemoICP.scala:20-32): IcpConfiguration(maxIterations, initialSigma = 1, endSigma = 1) with the default surface
correspondence, evaluatorUncertainty = 5, randomMixture = 0.5, model-to-target likelihood over all vertices (the demo decimates
both meshes to 100 points with scalismo's decimation, which is not restated: the full 1 622-vertex meshes are used).
One step = proposal (informed: surface ICP update with a posterior sample; or a random walk re-instantiated on the device)
+ likelihood of the proposal + both transition densities + accept / reject.
    PYTHONPATH=. python tools/bench_mh_chain.py [steps] [seed] [nofuse]     (nofuse: the call-by-call path instead of one native call per step)
Config 5 proper is 8 such chains, one per GPU, no communication ("replicas only"): `python bench.py --config 5
--gpus N` starts N of these processes, process i with HIP_VISIBLE_DEVICES=i, and passes barrier=<dir> index=<i> so that the chains
start together (every process writes <dir>/ready_<i> when it is set up and waits for <dir>/go)."""
import json
import os
import sys
import time

import numpy as np
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch  # noqa: F401  (first: one HIP runtime per process)

import gingr_amd as ga
from gingr_amd import sampling as sp

OPTS = dict(a.split("=") for a in sys.argv[1:] if "=" in a)       # tri_grid=0|1: GINGR_OPT_TRI_GRID for this run
POS = [a for a in sys.argv[1:] if "=" not in a]
steps = int(POS[0]) if len(POS) > 0 else 300
seed = int(POS[1]) if len(POS) > 1 else 0
fused = not (len(POS) > 2 and POS[2] == "nofuse")
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
d, m = np.load(os.path.join(root, "inputs.npz")), np.load(os.path.join(root, "femur_mesh.npz"))
ref, target = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
ctx = ga.Context(0)
if "tri_grid" in OPTS:
    from gingr_amd import _native as nat
    ctx.set_option(nat.OPT_TRI_GRID, int(OPTS["tri_grid"]))
model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).Gaussian(sigma=70.0, scaling=50.0)
model.cells = m["femur_cells"]
algo = ga.IcpRegistration(ctx)
cfg = ga.IcpConfiguration(maxIterations=steps + 1, initialSigma=1.0, endSigma=1.0, correspondenceMethod="TriangularClosestPoint")
s0 = algo.createInitialState(model, target, cfg, targetCells=m["femur_target_cells"])
settings = sp.ProbabilisticSettings(sp.IndependentPoints(algo, s0, 5.0), randomMixture=0.5, fusedSteps=fused)
counts = {}


class Log:
    def accept(self, cur, prop, gen, ev):
        c = counts.setdefault(prop.general.generatedBy, [0, 0])
        c[0] += 1

    def reject(self, cur, prop, gen, ev):
        c = counts.setdefault(prop.general.generatedBy, [0, 0])
        c[1] += 1


ev = sp.EvaluatorWrapper(True, settings.evaluators)
v0 = ev.logValue(s0)
t_go = None
if "barrier" in OPTS:       # several chains side by side: start the chain loops together
    bdir, bidx = OPTS["barrier"], OPTS.get("index", "0")
    open(os.path.join(bdir, f"ready_{bidx}"), "w").close()
    t_wait = time.time()
    while not os.path.exists(os.path.join(bdir, "go")):
        if time.time() - t_wait > 600:
            sys.exit("barrier: no go file after 600 s")
        time.sleep(0.002)
    t_go = time.time()
t0 = time.perf_counter()
best = algo.run(s0, acceptRejectLogger=Log(), probabilisticSettings=settings, rnd=sp.Random(seed))
dt = time.perf_counter() - t0
rc = ga.RegistrationComparison(ctx, verbose=False)
fit = ga.TriangleMesh3D(np.asarray(best.general.fit), model.cells)
avg, mx = rc.evaluateReconstruction2GroundTruthBoundaryAware("", fit, ga.TriangleMesh3D(target, m["femur_target_cells"]))
print(json.dumps({"what": "MH-in-GiNGR chain, femur, surface ICP proposals (config 5, one chain)", "vertices": int(ref.shape[0]),
                  "rank": int(model.rank), "steps": steps, "fused_steps": fused, "steps_per_s": steps / dt, "ms_per_step": dt / steps * 1e3,
                  "log_value_initial": v0, "log_value_best": ev.logValue(best),
                  "accepted_rejected_by_proposal": counts, "avg_surface_distance_best": avg, "max_surface_distance_best": mx,
                  "status": int(best.general.status), "seed": seed,
                  "device": {"name": torch.cuda.get_device_properties(0).name, "uuid": str(getattr(torch.cuda.get_device_properties(0), "uuid", "")),
                             "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES")},
                  "loop_started_unix": t_go, "loop_seconds": dt}))

"""CPD with landmark observations at the metric size (35 landmarks as in E/data/armadillo/armadillo.json, full 3 x 3 covariances):
what landmarks_kernel costs per iteration.   rocprofv3 --kernel-trace --stats -- python3 tools/experiments/prof_landmarks.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gingr_amd as ga
from bench import synth_clouds
n, rank, nlm = 50000, 100, 35
y, x = synth_clouds(n)
ctx = ga.Context(0)
model = ga.GPMMTriangleMesh3D(ctx, y, relativeTolerance=0.0, maxRank=rank).Gaussian(70.0, 50.0)
rng = np.random.default_rng(0)
pids = rng.choice(n, nlm, replace=False)
covs = np.tile(np.eye(3), (nlm, 1, 1)) * 4.0 + rng.normal(0, 0.2, (nlm, 3, 3))
covs = 0.5 * (covs + covs.transpose(0, 2, 1)) + 2.0 * np.eye(3)
lms = ga.LandmarkCorrespondences(pids, y[pids] + rng.normal(0, 1.0, (nlm, 3)), covs)
algo = ga.CpdRegistration(ctx)
state = algo.createInitialState(model, x, ga.CpdConfiguration(maxIterations=100, w=0.1, useLandmarkCorrespondence=True), landmarks=lms)
import time
state = algo.update(state)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    state = algo.update(state)
ctx.synchronize()
print("ms per update (host boundary)", (time.perf_counter() - t0) / 20 * 1e3, "status", state.general.status)

import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
import gingr_amd as ga
from oracle import gingr_oracle as go
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "test_gpu_structured_inputs.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
ctx = ga.Context(0)
ref, cells = t.grid_mesh(16, 30.0); ref[:, 2] = 0.02 * (ref[:, 0] ** 2 - ref[:, 1] ** 2) / 30.0
tgt, tcells = t.grid_mesh(18, 34.0); tgt[:, 2] = 0.02 * (tgt[:, 0] ** 2 - tgt[:, 1] ** 2) / 30.0 + 1.0
mo = go.build_gpmm_mixture(ref, [25.0], [4.0], 0.0, 18)
model = ga.PointDistributionModel(reference=ref, mean=np.zeros_like(ref), basis=mo.U, variance=mo.lam, cells=cells)
icp = ga.IcpRegistration(ctx)
cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=4.0, endSigma=1.0, correspondenceMethod="TriangularClosestPoint")
st = icp.createInitialState(model, tgt, cfg, transform=ga.GlobalTranformationType.RigidTransforms, targetCells=tcells)
for it in range(4):
    cp, w = icp.surfaceCorrespondence(st)
    ocp, ow, _ = go.surface_correspondence(st.general.fit, cells, tgt, tcells)
    diff = np.flatnonzero(w != ow)
    print(f"iteration {it}: accepted device {int(w.sum())} oracle {int(ow.sum())} differing {diff.shape[0]} cp max diff {np.abs(cp-ocp).max():.2e}")
    for i in diff[:5]:
        print("   vertex", i, "device", w[i], "oracle", ow[i], "fit", st.general.fit[i])
    st = icp.update(st)

"""(with the counting patch of line_nearest_kernel only) tiles / quarters / groups visited per query of the along-normal search at 41k,
and where the expensive workgroups sit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "6", "method=AlongNormalClosestPoint", "n=1", "resident=1"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_icp_surface.py"))
cp, w = g["algo"].surfaceCorrespondence(g["state"])
fit = np.asarray(g["state"].general.fit)
print("tiles", cp[:, 0].mean(), cp[:, 0].max(), "quarters", cp[:, 1].mean(), cp[:, 1].max(), "groups", cp[:, 2].mean(), cp[:, 2].max())
q = cp[:, 1]
print("quarter histogram", np.histogram(q, bins=[0, 4, 8, 16, 32, 64, 128, 256])[0])
worst = np.argsort(-q)[:12]
for k in worst:
    print("  quarters", q[k], "vertex", np.round(fit[k], 1), "unit", np.round(fit[k] / np.linalg.norm(fit[k]), 2))

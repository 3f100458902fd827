#!/bin/bash
# round 3, step 35: one-off wider sweep of tests/test_gpu_nn_grid.py::test_random_geometries (seeds 24..423)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s35; mkdir -p $O; cd $R
timeout 2400 python3 - > $O/sweep.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import tests.test_gpu_nn_grid as t
import gingr_amd as ga
ctx = ga.Context(0)
bad = 0
for seed in range(24, 424):
    try:
        t.test_random_geometries(ctx, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, e)
print("seeds 24..423 done, failures:", bad)
ctx.close()
PY
tail -5 $O/sweep.txt

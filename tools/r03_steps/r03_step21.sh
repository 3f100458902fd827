#!/bin/bash
# round 3, step 21: grid closest-point search of the ICP update: parity tests, then ICP iteration rate and distance tests, grid on / off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s21; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_nn_grid.py -m gpu -x -q > $O/pytest_grid.txt 2>&1; echo "rc=$?" >> $O/pytest_grid.txt; tail -15 $O/pytest_grid.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_rigid_icp.py tests/test_gpu_cabi_from_c.py tests/test_gpu_configs_and_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_icp.txt 2>&1; echo "rc=$?" >> $O/pytest_icp.txt; tail -4 $O/pytest_icp.txt
for g in 0 1 0 1; do GINGR_NN_GRID=$g GINGR_BENCH_NN_COUNT=1 python3 tools/bench_icp.py 50000 2>/dev/null | tail -1 >> $O/icp50k.txt; done
for g in 0 1; do GINGR_NN_GRID=$g GINGR_BENCH_NN_COUNT=1 python3 tools/bench_icp.py 15000 2>/dev/null | tail -1 >> $O/icp15k.txt; done
for g in 0 1; do GINGR_NN_GRID=$g GINGR_BENCH_NN_COUNT=1 python3 tools/bench_icp.py 100000 2>/dev/null | tail -1 >> $O/icp100k.txt; done
cat $O/icp50k.txt $O/icp15k.txt $O/icp100k.txt

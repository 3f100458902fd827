#!/bin/bash
# round 3, step 26: random-geometry test of the grid search; whole gpu suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s26; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_nn_grid.py -m gpu -q > $O/pytest_grid.txt 2>&1; echo "rc=$?" >> $O/pytest_grid.txt; tail -12 $O/pytest_grid.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt | tail -2

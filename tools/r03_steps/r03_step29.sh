#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s29; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_nn_grid.py -m gpu -q -k "benchmark_size" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -8 $O/pytest.txt

#!/bin/bash
# round 3, step 31: the multi-rank bench line after the rendezvous timeout change (ranks sharing the one GPU)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s31; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_golden_and_shards.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt

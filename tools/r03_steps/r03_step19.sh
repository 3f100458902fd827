#!/bin/bash
# round 3, step 19: the new Gram tile-count test, then the round-end evidence run on the final tree (tools/final_profile.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s19; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "weighted_gram" > $O/pytest_gram.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gram.txt; grep -n "passed\|failed\|Error\|assert" $O/pytest_gram.txt | tail -8
bash tools/final_profile.sh > $O/final.txt 2>&1; tail -20 $O/final.txt

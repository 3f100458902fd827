#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s10; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
bash tools/final_profile.sh > $O/final.txt 2>&1; tail -25 $O/final.txt

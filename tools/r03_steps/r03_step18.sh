#!/bin/bash
# round 3, step 18: gpu tests on the final Gram kernel; kernel averages at 50k (incl. the one-off unweighted launch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s18; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt | tail -2
bash tools/prof_stats.sh cur_50k --steps 20 --warmup 5 --roofline-steps 0 | grep -i "gram_tri" > $O/gram.txt; cat $O/gram.txt
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --steps 20 --warmup 5 > $O/ab50k.txt 2>&1; cat $O/ab50k.txt

#!/bin/bash
# round 3, step 14: Gram step with the vector instructions cut (producer-side scaling, incremental addresses), hand-over between the MFMAs; against the round-2 loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s14; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gram.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gram.txt; tail -3 $O/pytest_gram.txt
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --steps 20 --warmup 5 > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 3 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --points 1622 --steps 300 --warmup 20 --roofline-steps 3 > $O/ab_1622.txt 2>&1; cat $O/ab_1622.txt

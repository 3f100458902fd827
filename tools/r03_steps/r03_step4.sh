#!/bin/bash
# round 3, step 4: post-solve with up-front loads + 4-way row reduction: parity, A/B against round 2, kernel stats, SQ / MFMA counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s4; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
V=("r02=libgingr_hip_r02.so" "cur=" "cur_nopre=,GINGR_POST_PRELOAD=0")
bash tools/abn.sh "${V[@]}" -- > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/abn.sh "${V[@]}" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 0 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/abn.sh "${V[@]}" -- --points 15000 --steps 100 --warmup 10 > $O/ab15k.txt 2>&1; cat $O/ab15k.txt
bash tools/abn.sh "${V[@]}" -- --points 1622 --steps 300 --warmup 10 > $O/ab1622.txt 2>&1; cat $O/ab1622.txt
bash tools/prof_stats.sh s4 > $O/kernel_stats.txt 2>&1; cp gpurun_out/prof_s4/kernel_stats.csv $O/kernel_stats_50k.csv
bash tools/prof_emu8.sh > $O/prof_emu8.txt 2>&1; tail -30 $O/prof_emu8.txt
bash tools/pmc_sq.sh > $O/pmc_sq.txt 2>&1; tail -60 $O/pmc_sq.txt

#!/bin/bash
# round 3, step 5: 1024-thread row reduction, single-call sharded driver, group hardening: parity; A/B incl. Gram prefetch depths
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s5; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
V=("r02=libgingr_hip_r02.so" "cur=" "gd2=libgingr_hip_gd2.so" "gd4=libgingr_hip_gd4.so" "gd6=libgingr_hip_gd6.so")
bash tools/abn.sh "${V[@]}" -- > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/abn.sh "${V[@]}" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 3 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/abn.sh "${V[@]}" -- --points 15000 --steps 100 --warmup 10 > $O/ab15k.txt 2>&1; cat $O/ab15k.txt
bash tools/abn.sh "r02=libgingr_hip_r02.so" "cur=" -- --points 1622 --steps 300 --warmup 10 > $O/ab1622.txt 2>&1; cat $O/ab1622.txt
bash tools/prof_emu8.sh > $O/prof_emu8.txt 2>&1; head -24 $O/prof_emu8.txt

#!/bin/bash
# round 3, step 8: Gram prefetch ring without register moves (exact load counters): parity, depth 2 against 3; config 1 diagnosis
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s8; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
V=("r02=libgingr_hip_r02.so" "cur=" "gd3=libgingr_hip_gd3.so")
bash tools/abn.sh "${V[@]}" -- > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/abn.sh "${V[@]}" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 3 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/abn.sh "${V[@]}" -- --points 15000 --steps 100 --warmup 10 > $O/ab15k.txt 2>&1; cat $O/ab15k.txt
bash tools/abn.sh "${V[@]}" -- --points 1622 --steps 300 --warmup 10 > $O/ab1622.txt 2>&1; cat $O/ab1622.txt
timeout 600 python3 tools/diag_config1.py > $O/diag_config1.txt 2>&1; cat $O/diag_config1.txt | grep "^{" | cut -c1-200

#!/bin/bash
# round 3, step 11: integer-scaling (no-flush) form of the exponential in the dense regime: parity, A/B against the same build without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s11; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
V=("nonf=libgingr_hip_nonf.so" "cur=")
bash tools/abn.sh "${V[@]}" -- > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/abn.sh "${V[@]}" -- --points 100000 --steps 10 --warmup 2 > $O/ab100k.txt 2>&1; cat $O/ab100k.txt
bash tools/abn.sh "${V[@]}" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 3 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/abn.sh "${V[@]}" -- --points 15000 --steps 100 --warmup 10 > $O/ab15k.txt 2>&1; cat $O/ab15k.txt

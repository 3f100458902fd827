#!/bin/bash
# round 3, step 1: parity of the fused (last-arriver) CPD passes + same-box A/B against the round-2 library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s1; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
bash tools/ab.sh > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/ab.sh --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 0 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/ab.sh --points 15000 --steps 100 --warmup 10 > $O/ab15k.txt 2>&1; cat $O/ab15k.txt
bash tools/ab.sh --points 1622 --steps 300 --warmup 10 > $O/ab1622.txt 2>&1; cat $O/ab1622.txt

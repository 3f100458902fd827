#!/bin/bash
# round 3, step 6: look-ahead Cholesky + pipelined Gram hand-over: stage cycles, parity, A/B, one line per BASELINE config
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s6; mkdir -p $O; cd $R
timeout 120 tools/ubench_solve > $O/ubench_solve.txt 2>&1; cat $O/ubench_solve.txt
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
V=("r02=libgingr_hip_r02.so" "cur=")
bash tools/abn.sh "${V[@]}" -- > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/abn.sh "${V[@]}" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 3 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/abn.sh "${V[@]}" -- --points 15000 --steps 100 --warmup 10 > $O/ab15k.txt 2>&1; cat $O/ab15k.txt
bash tools/abn.sh "${V[@]}" -- --points 1622 --steps 300 --warmup 10 > $O/ab1622.txt 2>&1; cat $O/ab1622.txt
timeout 1500 python3 tools/bench_configs.py > $O/configs.txt 2> $O/configs.err; cut -c1-400 $O/configs.txt; tail -5 $O/configs.err

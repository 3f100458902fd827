#!/bin/bash
# round 3, step 3: granule-hand-off fused tail: parity + A/B against round 2 and against the three-launch tail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s3; mkdir -p $O; cd $R
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
V=("r02=libgingr_hip_r02.so" "cur=" "cur_notail=,GINGR_FUSED_TAIL=0")
bash tools/abn.sh "${V[@]}" -- > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/abn.sh "${V[@]}" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 0 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/abn.sh "${V[@]}" -- --points 15000 --steps 100 --warmup 10 > $O/ab15k.txt 2>&1; cat $O/ab15k.txt
bash tools/abn.sh "${V[@]}" -- --points 1622 --steps 300 --warmup 10 > $O/ab1622.txt 2>&1; cat $O/ab1622.txt
python3 tools/bench_icp.py > $O/icp.txt 2>&1; tail -3 $O/icp.txt
GINGR_FUSED_TAIL=0 python3 tools/bench_icp.py > $O/icp_notail.txt 2>&1; tail -3 $O/icp_notail.txt

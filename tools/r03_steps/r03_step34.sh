#!/bin/bash
# round 3, step 34: the Euler round trip of the pose step spread over the lanes of its wave: tests, post-solve kernel averages, iteration times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s34; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt | tail -1
for v in prev cur prev cur; do
  if [ $v = prev ]; then export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_prev.so; else unset GINGR_HIP_LIB; fi
  bash tools/prof_stats.sh ${v}_emu8 --emulate-world 8 --steps 50 --warmup 5 --roofline-steps 0 | grep -i "post_solve" | sed "s/^/$v emu8 /" >> $O/post_avgs.txt
  bash tools/prof_stats.sh ${v}_1622 --points 1622 --steps 100 --warmup 5 --roofline-steps 0 | grep -i "post_solve" | sed "s/^/$v 1622 /" >> $O/post_avgs.txt
done
unset GINGR_HIP_LIB
cat $O/post_avgs.txt | cut -c1-150
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 0 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt | cut -c1-80
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --points 1622 --steps 300 --warmup 20 --roofline-steps 0 > $O/ab_1622.txt 2>&1; cat $O/ab_1622.txt | cut -c1-80
for v in prev cur prev cur; do
  if [ $v = prev ]; then export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_prev.so; else unset GINGR_HIP_LIB; fi
  python3 tools/bench_icp.py 50000 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v icp50k', round(d['ms_per_iteration'],5), d['fit_checksum'])"
done

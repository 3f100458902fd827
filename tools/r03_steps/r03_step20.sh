#!/bin/bash
# round 3, step 20: post-solve kernel touching its own code lines (L2 warm-up) against the build without; rocprofv3 averages + iteration times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s20; mkdir -p $O; cd $R
for v in prev cur prev cur; do
  if [ $v = prev ]; then export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_prev.so; else unset GINGR_HIP_LIB; fi
  bash tools/prof_stats.sh ${v}_50k --steps 20 --warmup 5 --roofline-steps 0 | grep -i "post_solve\|post_matvecs" | sed "s/^/$v 50k  /" >> $O/post_avgs.txt
  bash tools/prof_stats.sh ${v}_emu8 --emulate-world 8 --steps 50 --warmup 5 --roofline-steps 0 | grep -i "post_solve\|post_matvecs" | sed "s/^/$v emu8 /" >> $O/post_avgs.txt
  bash tools/prof_stats.sh ${v}_1622 --points 1622 --steps 100 --warmup 5 --roofline-steps 0 | grep -i "post_solve\|post_matvecs" | sed "s/^/$v 1622 /" >> $O/post_avgs.txt
  bash tools/prof_stats.sh ${v}_15k --points 15000 --steps 50 --warmup 5 --roofline-steps 0 | grep -i "post_solve\|post_matvecs" | sed "s/^/$v 15k  /" >> $O/post_avgs.txt
done
unset GINGR_HIP_LIB
cat $O/post_avgs.txt
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 0 > $O/ab_emu8.txt 2>&1; cat $O/ab_emu8.txt
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --points 15000 --steps 100 --warmup 10 --roofline-steps 0 > $O/ab_15k.txt 2>&1; cat $O/ab_15k.txt
bash tools/abn.sh "prev=libgingr_hip_prev.so" "cur=" -- --points 1622 --steps 300 --warmup 20 --roofline-steps 0 > $O/ab_1622.txt 2>&1; cat $O/ab_1622.txt

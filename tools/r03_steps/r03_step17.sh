#!/bin/bash
# round 3, step 17: rocprofv3 kernel averages of the Gram kernel, round-2 loop (prev) against the new one (cur), three sizes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s17; mkdir -p $O; cd $R
for v in prev cur prev cur; do
  if [ $v = prev ]; then export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_prev.so; else unset GINGR_HIP_LIB; fi
  bash tools/prof_stats.sh ${v}_50k --steps 20 --warmup 5 --roofline-steps 0 | grep -i "gram_tri\|post_solve\|posterior_solve" | sed "s/^/$v 50k  /" >> $O/gram_avgs.txt
  bash tools/prof_stats.sh ${v}_emu8 --emulate-world 8 --steps 50 --warmup 5 --roofline-steps 0 | grep -i "gram_tri" | sed "s/^/$v emu8 /" >> $O/gram_avgs.txt
  bash tools/prof_stats.sh ${v}_1622 --points 1622 --steps 100 --warmup 5 --roofline-steps 0 | grep -i "gram_tri" | sed "s/^/$v 1622 /" >> $O/gram_avgs.txt
  bash tools/prof_stats.sh ${v}_15k --points 15000 --steps 50 --warmup 5 --roofline-steps 0 | grep -i "gram_tri" | sed "s/^/$v 15k /" >> $O/gram_avgs.txt
done
cat $O/gram_avgs.txt

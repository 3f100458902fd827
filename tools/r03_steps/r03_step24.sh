#!/bin/bash
# round 3, step 24: grid search variants (lanes per query x forced occupancy): ICP iteration time at three sizes + kernel average at 50k
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s24; mkdir -p $O; cd $R
for v in cur g16_8 g8_1 g8_8 cur g16_8 g8_1 g8_8; do
  if [ $v = cur ]; then unset GINGR_HIP_LIB; else export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_$v.so; fi
  for n in 50000 15000 100000; do python3 tools/bench_icp.py $n 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['points'], round(d['ms_per_iteration'],5), d['fit_checksum'])" >> $O/variants.txt; done
done
cat $O/variants.txt
cd /tmp && export TMPDIR=/tmp
for v in cur g16_8 g8_1 g8_8; do
  if [ $v = cur ]; then unset GINGR_HIP_LIB; else export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats -d $O/prof_$v -o p --output-format csv -- python3 $R/tools/bench_icp.py 50000 > $O/log_$v.txt 2>&1
  python3 - $O/prof_$v $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "nn_grid" in r["Name"]: print(sys.argv[2], r["Name"][:50], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done

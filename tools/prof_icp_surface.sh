#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_icp_surface
mkdir -p $OUT
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d $OUT -o icp --output-format csv -- python3 $R/tools/bench_icp_surface.py 6 > $OUT/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_icp_surface/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} {r["Percentage"]}%')
PY
tail -1 $OUT/log.txt | cut -c1-300

#!/bin/bash
# rocprofv3 kernel statistics of the ICP update (point-cloud closest point) on the synthetic workload: tools/prof_icp.sh [points]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_icp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $R/tools/bench_icp.py "$@" > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
cd $R
python3 - <<'PY'
import csv, glob, shutil
f = glob.glob("gpurun_out/prof_icp/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(f, "gpurun_out/prof_icp/kernel_stats.csv")
for r in list(csv.DictReader(open(f)))[:25]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f} {r["Percentage"]}%')
PY

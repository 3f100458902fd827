#!/bin/bash
# kernel statistics of one tools/*.py workload: tools/experiments/prof_tool.sh <tag> <script> [args]; top rows printed,
# CSV under gpurun_out/prof_<tag>/kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
S=$R/$1; shift
export PYTHONPATH=$R
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $S "$@" > $OUT/log.txt 2>&1
cd $R
echo "== $TAG"; tail -3 $OUT/log.txt | cut -c1-300
python3 - "$TAG" <<'PY'
import csv, glob, shutil, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(f, f"gpurun_out/prof_{sys.argv[1]}/kernel_stats.csv")
for r in list(csv.DictReader(open(f)))[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY

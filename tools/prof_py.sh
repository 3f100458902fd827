#!/bin/bash
# rocprofv3 kernel statistics of any python tool of this repository: tools/prof_py.sh <tag> <script> [args]; summary under gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
SCRIPT=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $R/$SCRIPT "$@" > $OUT/log.txt 2>&1
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
import shutil
shutil.copy(f, f"gpurun_out/prof_{sys.argv[1]}/kernel_stats.csv")
for r in rows[:32]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f} {r["Percentage"]}%')
PY

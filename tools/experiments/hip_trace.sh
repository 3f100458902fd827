#!/bin/bash
# HIP API statistics of one python workload: tools/experiments/hip_trace.sh <tag> <script> [args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
S=$R/$1; shift
export PYTHONPATH=$R
OUT=$R/gpurun_out/hip_$TAG
mkdir -p $OUT
rocprofv3 --hip-trace --stats -d $OUT -o p --output-format csv -- python3 $S "$@" > $OUT/log.txt 2>&1
cd $R
grep -E "^rep" $OUT/log.txt
python3 - "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/hip_{sys.argv[1]}/**/*hip_api_stats.csv", recursive=True)
if not f:
    f = glob.glob(f"gpurun_out/hip_{sys.argv[1]}/**/*stats*.csv", recursive=True)
    print(f)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print(f'{r["Name"][:40]:40s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_classic
mkdir -p $OUT
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d $OUT -o cc --output-format csv -- python3 $R/tools/bench_classic_cpd.py 5000 > $OUT/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_classic/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} {r["Percentage"]}%')
PY

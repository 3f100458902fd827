#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-8}
OUT=$R/gpurun_out/prof_emu$N
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o emu --output-format csv -- python3 $R/bench.py --emulate-world $N --no-cpu-baseline --steps 50 --warmup 5 --roofline-steps 0 > $OUT/log.txt 2>&1
cd $R
python3 - "$N" <<'PY'
import csv, glob, sys
n = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_emu{n}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_ns {float(r["AverageNs"]):10.0f} total_ms {float(r["TotalDurationNs"])/1e6:9.3f} {r["Percentage"]}%')
PY

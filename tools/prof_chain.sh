#!/bin/bash
# rocprofv3 kernel statistics of the Metropolis-Hastings chain (BASELINE config 5): tools/prof_chain.sh [steps]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_chain
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $R/tools/bench_mh_chain.py ${1:-300} > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-200
cd $R
python3 - <<'PY'
import csv, glob, shutil
f = glob.glob("gpurun_out/prof_chain/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(f, "gpurun_out/prof_chain/kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:22]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f} {r["Percentage"]}%')
PY

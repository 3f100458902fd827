#!/bin/bash
# kernel statistics over the single-process GPU tests (kernels whose MAXIMUM is large on test-sized inputs are the latency-serial ones)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_tests
rm -rf $OUT; mkdir -p $OUT
cd $R
FILES=$(ls tests/test_gpu_*.py | grep -v -E "multirank|cabi_from_c|test_gpu_fuzz|golden_and_shards|multi_device|no_leaks|fullsize|bench_workload")
rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 -m pytest $FILES -m gpu -x -q > $OUT/log.txt 2>&1
grep -E "passed|failed" $OUT/log.txt | tail -2
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_tests/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["MaxNs"]))
out = open("gpurun_out/prof_tests/by_max.txt", "w")
for r in rows[:60]:
    line = f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:10.1f} max_us {float(r["MaxNs"])/1e3:10.1f}'
    print(line); out.write(line + "\n")
PY
find $OUT -name "*kernel_trace.csv" -delete

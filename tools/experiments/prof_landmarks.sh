#!/bin/bash
# kernel statistics of tools/experiments/prof_landmarks.py; summary under gpurun_out/prof_landmarks/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_landmarks
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $R/tools/experiments/prof_landmarks.py > $OUT/log.txt 2>&1
cd $R
tail -2 $OUT/log.txt
python3 - <<'PY'
import csv, glob, shutil
f = glob.glob("gpurun_out/prof_landmarks/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(f, "gpurun_out/prof_landmarks/kernel_stats.csv")
for r in list(csv.DictReader(open(f)))[:12]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY

#!/bin/bash
# FETCH_SIZE calibration for 16 B, 8 B and Gram-style (4 x 128 B) loads: every kernel of tools/ubench_fetch reads 256 MiB once.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/ubench_fetch.hip -o /tmp/ubench_fetch || exit 1
OUT=$R/gpurun_out/pmc_fetch_cal
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o p --output-format csv -- /tmp/ubench_fetch > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if row["Counter_Name"] == "FETCH_SIZE":
        acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:20s} FETCH_SIZE {sum(v)/len(v):12.0f} KiB per launch for 262144 KiB read  => factor {262144.0/(sum(v)/len(v)):.3f}")
PY

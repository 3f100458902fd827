#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_icp
mkdir -p $OUT
export PYTHONPATH=$R
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --kernel-trace -d $OUT -o p --output-format csv -- python3 $R/tools/bench_icp_surface.py 6 > $OUT/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_icp/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = next((n for n in ("surface_cp", "self_intersect", "nn_kernel") if n in row["Kernel_Name"]), None)
    if k: acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY

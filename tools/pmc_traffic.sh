#!/bin/bash
# HBM traffic per kernel from PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes with --kernel-trace only.  Prints mean KiB per launch for the main kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$R/gpurun_out/pmc_$c
  mkdir -p $OUT
  rocprofv3 --pmc $c --kernel-trace -d $OUT -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --roofline-steps 0 > $OUT/log.txt 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
names = ("cpd_colsum", "cpd_rowstats", "rowstats_reduce", "chunk_reduce", "gram_tri", "gram_reduce", "sweep_kernel<0", "sweep_kernel<4", "posterior_solve")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = next((n for n in names if n in row["Kernel_Name"]), None)
        if k and row["Counter_Name"] == c:
            acc[k].append(float(row["Counter_Value"]))
    for k in names:
        if acc[k]:
            print(f"{c:10s} {k:18s} mean {sum(acc[k]) / len(acc[k]):12.0f} KiB per launch over {len(acc[k])} launches")
PY

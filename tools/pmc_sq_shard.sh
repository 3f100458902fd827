#!/bin/bash
# SQ issue counters of the two pair loops on the 8-rank row shard of the metric workload (one PMC pass, kernel-trace only): how much of a
# one-round launch the vector ALUs are issuing, next to the same figure of the full-size launch (tools/pmc_sq.sh).  Run via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq_shard
rm -rf $OUT; mkdir -p $OUT
for W in 8 1; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES \
  --kernel-trace -d $OUT/w$W -o p --output-format csv -- python3 $R/bench.py --emulate-world $W --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check --roofline-steps 0 --sustained-steps 0 > $OUT/w$W.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
print("# rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES ... --kernel-trace -- python3 bench.py --emulate-world W --steps 6 --warmup 2")
print("# valu_active = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines): the share of the launch's cycles in which a SIMD issues a vector instruction")
for W in (8, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_sq_shard/w{W}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = next((n for n in ("cpd_colsum", "cpd_rowstats") if n in row["Kernel_Name"]), None)
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        cyc = m["SQ_BUSY_CYCLES"] / 32.0
        print(f"world {W} {k:14s} launches {len(d['SQ_BUSY_CYCLES']):3d}  kernel cycles {cyc:10.0f}  waves {m['SQ_WAVES']:8.0f}  SQ_INSTS_VALU {m['SQ_INSTS_VALU']:14.0f}  "
              f"valu_active {4.0 * m['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc):.3f}  any_active {4.0 * m['SQ_ACTIVE_INST_ANY'] / (1024.0 * cyc):.3f}")
PY

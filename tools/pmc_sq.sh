#!/bin/bash
# SQ issue / LDS counters of the affinity kernels (two PMC passes, kernel-trace only).  Run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check --roofline-steps 1 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_SCA \
  --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check --roofline-steps 1 > $OUT/p2.log 2>&1
# third pass: the matrix-pipe counters (BASELINE north_star: "MFMA-util counters"): busy cycles of the MFMA pipe, MFMA instructions, f64 MFMA
# operations, and the kernel's wave / busy cycles to normalise them
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM \
  --kernel-trace -d $OUT/p3 -o p3 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check --roofline-steps 1 > $OUT/p3.log 2>&1
rocprofv3 -L > $OUT/counter_list.txt 2>&1
grep -i -E "MFMA" $OUT/counter_list.txt | head -40 > $OUT/mfma_counters_available.txt
cd $R
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"gpurun_out/pmc_sq/{p}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = next((n for n in ("cpd_colsum", "cpd_rowstats", "nn_kernel", "gram_kernel", "gram_tri", "posterior_solve", "chol_trailing") if n in row["Kernel_Name"]), None)
            if k is None: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        with open(f"gpurun_out/pmc_sq/{p}_summary.txt", "w") as o:
            for k, d in acc.items():
                o.write(k + "\n")
                for c, v in d.items():
                    o.write(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}\n")
PY
cat gpurun_out/pmc_sq/p1_summary.txt gpurun_out/pmc_sq/p2_summary.txt gpurun_out/pmc_sq/p3_summary.txt; cat gpurun_out/pmc_sq/mfma_counters_available.txt

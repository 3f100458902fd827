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
cat gpurun_out/pmc_sq/p1_summary.txt gpurun_out/pmc_sq/p2_summary.txt gpurun_out/pmc_sq/p3_summary.txt
# the matrix-pipe figures bench.py reports next to its time-derived ones: gpurun_out/pmc_mfma.json (copy to profiles/rNN_pmc_mfma.json)
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq/p3/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = next((n for n in ("gram_tri_kernel", "posterior_solve_lds_kernel", "chol_trailing_kernel") if n in row["Kernel_Name"]), None)
        if k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"workload": {"points": 50000, "rank": 100, "gpus": 1},
       "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY ... "
                 "--kernel-trace only, bench.py --steps 3; means per launch summed over the chip.  SQ_BUSY_CYCLES is summed over the 32 "
                 "shader engines that were busy; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (64 per v_mfma_f64_16x16x4)", "kernels": {}}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {"counters": m, "launches": len(next(iter(d.values())))}
    if k == "gram_tri_kernel" and m.get("SQ_BUSY_CYCLES"):
        cyc = m["SQ_BUSY_CYCLES"] / 32.0                     # every shader engine is busy for the whole launch (256 workgroups)
        e["kernel_cycles"] = cyc
        e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        e["mfma_instructions"] = m.get("SQ_INSTS_MFMA")
        e["note"] = "busy cycles of the matrix pipes / (1024 SIMDs x kernel cycles)"
    out["kernels"][k] = e
json.dump(out, open("gpurun_out/pmc_mfma.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out["kernels"].items()}))
PY

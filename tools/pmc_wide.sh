#!/bin/bash
# PMC counters of the rank-256 run (the wide Gram pass, the super-panel solve, the rank-256 fit pass): HBM traffic (FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes, gfx950 corrections as in tools/pmc_traffic.sh) and the matrix-pipe counters, --kernel-trace only.
# Writes gpurun_out/pmc_wide_r256.json (copy to profiles/rNN_pmc_wide_r256.json).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RANK=${RANK:-256}
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$R/gpurun_out/pmcw_$c; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $c --kernel-trace -d $OUT -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-parity-check --rank $RANK --steps 4 --warmup 1 --roofline-steps 0 --sustained-steps 0 > $OUT/log.txt 2>&1
done
OUT=$R/gpurun_out/pmcw_MFMA; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-trace -d $OUT -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-parity-check --rank $RANK --steps 4 --warmup 1 --roofline-steps 0 --sustained-steps 0 > $OUT/log.txt 2>&1
cd $R
python3 - "$RANK" <<'PY'
import csv, glob, collections, json, sys
rank = int(sys.argv[1])
kernels = ("gram_wide_kernel", "row_expand_kernel", "posterior_solve_wide_kernel", "sweep_fit_boxes_kernel", "phase1_finalize_kernel", "post_matvecs_kernel")
def collect(tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmcw_{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = next((n for n in kernels if n in row["Kernel_Name"]), None)
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
fe, wr, mf = collect("FETCH_SIZE"), collect("WRITE_SIZE"), collect("MFMA")
out = {"workload": {"points": 50000, "rank": rank, "gpus": 1},
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ matrix-pipe counters in separate passes, --kernel-trace only, bench.py --rank R --steps 4; means "
                 "per launch; FETCH_SIZE x 2 (gfx950: coalesced streaming reads report half, tools/pmc_fetch_calibration.sh), WRITE_SIZE exact, both KiB",
       "kernels": {}}
for k in kernels:
    e = {}
    if k in fe or k in wr:
        f_, w_ = fe.get(k, {}).get("FETCH_SIZE", 0.0), wr.get(k, {}).get("WRITE_SIZE", 0.0)
        e.update({"fetch_kib": f_, "write_kib": w_, "hbm_bytes_per_launch": (2.0 * f_ + w_) * 1024.0})
    m = mf.get(k)
    if m:
        e["counters"] = m
        if m.get("SQ_BUSY_CYCLES") and k in ("gram_wide_kernel",):
            cyc = m["SQ_BUSY_CYCLES"] / 32.0          # every shader engine busy for the whole launch (256 workgroups)
            e["kernel_cycles"] = cyc
            e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
            e["note"] = "busy cycles of the matrix pipes / (1024 SIMDs x kernel cycles)"
    if e:
        out["kernels"][k] = e
json.dump(out, open("gpurun_out/pmc_wide_r256.json", "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in e.items() if kk != "counters"})
PY

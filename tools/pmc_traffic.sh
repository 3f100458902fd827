#!/bin/bash
# HBM traffic per kernel from PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes with --kernel-trace only (no other trace domain).  Writes
# gpurun_out/pmc_traffic.json (copy it to profiles/rNN_pmc_traffic.json: bench.py reads roofline.traffic from that tracked file)
# and prints a table.  gfx950 corrections (same guide): FETCH_SIZE reports half the bytes of coalesced streaming reads; the guide
# calibrates that for 16 B / lane, tools/pmc_fetch_calibration.sh (tools/ubench_fetch.hip: three kernels reading 256 MiB once)
# shows the same factor 2.000 for 8 B / lane and for the Gram kernel's 4 x 128-byte segments, so x2 is applied to every kernel;
# WRITE_SIZE is exact.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
POINTS=${POINTS:-50000}
RANK=${RANK:-100}
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$R/gpurun_out/pmc_$c
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $c --kernel-trace -d $OUT -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-parity-check --points $POINTS --rank $RANK --steps 4 --warmup 1 --roofline-steps 0 > $OUT/log.txt 2>&1
done
cd $R
python3 - "$POINTS" "$RANK" <<'PY'
import csv, glob, collections, json, sys
points, rank = int(sys.argv[1]), int(sys.argv[2])
names = {"cpd_colsum_kernel": "cpd_colsum", "cpd_rowstats_kernel": "cpd_rowstats", "rowstats_reduce_kernel": "rowstats_reduce",
         "cpd_den_finalize_kernel": "cpd_den_finalize", "gram_tri_kernel": "gram_tri", "phase1_finalize_kernel": "phase1_finalize",
         "sweep_kernel": "sweep_kernel<", "sweep_fit_boxes_kernel": "sweep_fit_boxes", "posterior_solve_lds_kernel": "posterior_solve", "tile_bbox_kernel": "tile_bbox"}
wide = set(names)       # FETCH_SIZE x 2 on gfx950 for EVERY load width this library uses: calibrated with tools/pmc_fetch_calibration.sh
                        # (16 B / lane, 8 B / lane contiguous and the Gram kernel's 4 x 128-byte row segments all report exactly half)
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = next((n for n, pat in names.items() if pat in row["Kernel_Name"]), None)
        if k and row["Counter_Name"] == c:
            acc[k].append(float(row["Counter_Value"]))
    vals[c] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
out = {"workload": {"points": points, "rank": rank, "gpus": 1}, "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, "
       "--kernel-trace only; mean per launch; counters are KiB", "kernels": {}}
print(f'{"kernel":28s} {"FETCH KiB":>12s} {"WRITE KiB":>12s} {"HBM bytes/launch":>18s}')
for k in names:
    if k in vals["FETCH_SIZE"] or k in vals["WRITE_SIZE"]:
        fe, n1 = vals["FETCH_SIZE"].get(k, (0.0, 0))
        wr, n2 = vals["WRITE_SIZE"].get(k, (0.0, 0))
        corr = 2.0 if k in wide else 1.0
        hbm = (fe * corr + wr) * 1024.0
        out["kernels"][k] = {"fetch_kib": fe, "write_kib": wr, "fetch_correction": corr, "hbm_bytes_per_launch": hbm,
                             "launches": max(n1, n2),
                             "note": f"FETCH_SIZE x{corr:g} + WRITE_SIZE per launch, mean over {max(n1, n2)} launches"}
        print(f"{k:28s} {fe:12.0f} {wr:12.0f} {hbm:18.0f}")
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
PY

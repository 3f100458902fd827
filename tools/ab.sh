#!/bin/bash
# Same-box A/B of two builds of libgingr_hip.so (boxes of the pool differ by +-5 % in sustained clock, so versions are only
# comparable within ONE gpurun call): A = gingr_amd/libgingr_hip_prev.so, B = the in-tree library; alternating runs.
# usage: tools/ab.sh [bench.py arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/ab
for i in 1 2 3; do
  for v in prev cur; do
    if [ $v = prev ]; then export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_prev.so; else unset GINGR_HIP_LIB; fi
    python3 $R/bench.py --no-cpu-baseline --no-parity-check "$@" 2> $R/gpurun_out/ab/$v$i.err | tail -1 > $R/gpurun_out/ab/$v$i.json
  done
done
python3 - <<PY
import json, glob
for v in ("prev", "cur"):
    rows = [json.load(open(f)) for f in sorted(glob.glob("$R/gpurun_out/ab/%s?.json" % v))]
    print(v, " ".join("%.4f" % r["ms_per_step"] for r in rows), "| kernels (ms):",
          {k["kernel"].replace("_kernel", ""): round(k["avg_ms"], 4) for k in rows[-1]["kernels"]})
PY

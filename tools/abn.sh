#!/bin/bash
# Same-box comparison of several builds / settings of the library (boxes of the pool differ by +-5 % in sustained clock, so
# variants are only comparable within ONE gpurun call).  Each variant is "name=library.so[,ENV=VALUE...]" (library relative to
# the repository root, e.g. tools/bin/libgingr_hip_r03.so -- tools/bin/ is ignored by git and emptied at round end; empty = the
# in-tree build); runs alternate, three rounds.
# Variants that differ by a build-time knob are separate libraries: `make -C gingr_amd/csrc variant NAME=x DEFS="-DGINGR_...=v"`
# writes gingr_amd/libgingr_hip_x.so (nothing in the library reads the environment).
# usage: tools/abn.sh "r03=tools/bin/libgingr_hip_r03.so" "cur=" "chunks3=gingr_amd/libgingr_hip_chunks3.so" -- [bench.py arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done
shift
O=$R/gpurun_out/abn; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  for v in "${VARS[@]}"; do
    name=${v%%=*}; spec=${v#*=}
    lib=${spec%%,*}; envs=""
    if [[ "$spec" == *,* ]]; then envs=${spec#*,}; fi
    (
      if [ -n "$lib" ]; then export GINGR_HIP_LIB=$R/$lib GINGR_HIP_LIB_ALLOW_OLDER=1; fi
      IFS=',' read -ra E <<< "$envs"; for e in "${E[@]}"; do [ -n "$e" ] && export "$e"; done
      python3 $R/bench.py --no-cpu-baseline --no-parity-check --sustained-steps 0 "$@" 2> $O/$name$i.err | tail -1 > $O/$name$i.json
    )
  done
done
python3 - "$O" "${VARS[@]}" <<'PY'
import json, glob, sys
O = sys.argv[1]
for v in sys.argv[2:]:
    name = v.split("=")[0]
    rows = []
    for f in sorted(glob.glob(f"{O}/{name}?.json")):
        try: rows.append(json.load(open(f)))
        except Exception as e: rows.append({"ms_per_step": float("nan"), "kernels": []})
    print(f"{name:14s}", " ".join("%.4f" % r["ms_per_step"] for r in rows), "| valid", [r.get("valid") for r in rows][-1], "| kernels (ms):",
          {k["kernel"].replace("_kernel", ""): round(k["avg_ms"], 4) for k in rows[-1].get("kernels", [])})
PY

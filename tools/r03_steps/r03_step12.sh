#!/bin/bash
# round 3, step 12: final tree: gpu tests, smoke, one line per BASELINE config
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s12; mkdir -p $O; cd $R
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
timeout 1500 python3 tools/bench_configs.py > $O/configs.txt 2> $O/configs.err; cp gpurun_out/r03_configs.json $O/configs.json
python3 - <<'PY'
import json
for ln in open("gpurun_out/r03_s12/configs.txt"):
    c = json.loads(ln); print(c.get("config"), c.get("metric"), round(c.get("value", 0), 1), c.get("unit"), c.get("ms_per_step"), c.get("valid"), (c.get("roofline") or {}).get("frac"))
PY

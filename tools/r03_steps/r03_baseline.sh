#!/bin/bash
# round-3 baseline on this round's box: gpu tests, then the four sizes the review's targets are quoted on
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_base
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench50k.json
python3 bench.py --emulate-world 8 --no-cpu-baseline --no-parity-check --steps 100 --warmup 10 --roofline-steps 0 2>/dev/null | tail -1 > $O/emu8.json
python3 bench.py --points 15000 --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench15k.json
python3 bench.py --points 1622 --no-cpu-baseline --steps 300 --warmup 10 2>/dev/null | tail -1 > $O/bench1622.json
tail -3 $O/pytest.txt
python3 - <<'PY'
import json
for n in ("bench50k","emu8","bench15k","bench1622"):
    try:
        d=json.load(open(f"gpurun_out/r03_base/{n}.json")); print(n, d["ms_per_step"], d.get("valid"))
    except Exception as e: print(n, "ERR", e)
PY

#!/bin/bash
# round 3, step 32: what the driver runs at round end, on the final tree: smoke(), the gpu suite, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s32; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt | tail -1
python3 bench.py 2> $O/bench.err | tail -1 > $O/bench.json; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['valid'], d['roofline']['frac'], d['cpu_baseline']['value'])"

#!/bin/bash
# round 3, step 36: crowded-row cap of the grid search: tests, ICP rate unchanged?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s36; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt | tail -1
for n in 50000 15000 50000; do python3 tools/bench_icp.py $n 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['points'], round(d['ms_per_iteration'],5), d['fit_checksum'])"; done

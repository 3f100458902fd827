#!/bin/bash
# round 3, step 28: uniform-weight ICP with fewer launches (scaled Gram by the finalize kernel; observation inside the right-hand-side pass): tests + ICP rate
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s28; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt | tail -2
for n in 50000 15000 100000 1622 50000 15000; do python3 tools/bench_icp.py $n 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['points'], round(d['ms_per_iteration'],5), d['fit_checksum'])" >> $O/icp.txt; done
cat $O/icp.txt

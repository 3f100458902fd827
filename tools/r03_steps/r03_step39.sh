#!/bin/bash
# round 3, step 39: sanity of the tree after the look-ahead experiment was backed out
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s39; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt | tail -1
python3 tools/bench_icp.py 50000 2>/dev/null | tail -1 | cut -c1-200

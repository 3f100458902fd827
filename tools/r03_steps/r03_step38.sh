#!/bin/bash
# round 3, step 38: look-ahead closest-point search inside the fit pass (SWEEP_FIT_NN): tests, ICP rates with / without, kernel averages
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s38; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt | tail -1; grep -n "Error\|assert" $O/pytest.txt | head -5
for a in 0 1 0 1; do
  for n in 50000 15000 100000 1622; do GINGR_NN_LOOKAHEAD=$a python3 tools/bench_icp.py $n 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lookahead $a', d['points'], round(d['ms_per_iteration'],5), d['fit_checksum'])" >> $O/ab.txt; done
done
cat $O/ab.txt
bash tools/prof_icp.sh 50000 2>&1 | grep "sweep_kernel\|nn_" | cut -c1-140

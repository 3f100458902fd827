#!/bin/bash
# round 3, step 27: grid search, first pass over the ball of the warm bound instead of the fixed 27 cells: tests, ICP rates, kernel averages
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s27; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_nn_grid.py tests/test_gpu_rigid_icp.py tests/test_gpu_surface_icp.py -m gpu -x -q > $O/pytest_grid.txt 2>&1; echo "rc=$?" >> $O/pytest_grid.txt; tail -4 $O/pytest_grid.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi_from_c.py tests/test_gpu_configs_and_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_icp.txt 2>&1; echo "rc=$?" >> $O/pytest_icp.txt; tail -3 $O/pytest_icp.txt
for v in prev cur prev cur; do
  if [ $v = cur ]; then unset GINGR_HIP_LIB; else export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_$v.so; fi
  for n in 50000 15000 100000 1622; do GINGR_BENCH_NN_COUNT=1 python3 tools/bench_icp.py $n 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['points'], round(d['ms_per_iteration'],5), d['distance_tests_per_search'], d['fit_checksum'])" >> $O/variants.txt; done
done
cat $O/variants.txt
cd /tmp && export TMPDIR=/tmp
for v in prev cur; do
  if [ $v = cur ]; then unset GINGR_HIP_LIB; else export GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats -d $O/prof_$v -o p --output-format csv -- python3 $R/tools/bench_icp.py 50000 > $O/log_$v.txt 2>&1
  python3 - $O/prof_$v $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "nn_" in r["Name"]: print(sys.argv[2], r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done

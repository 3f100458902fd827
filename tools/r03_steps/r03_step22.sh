#!/bin/bash
# round 3, step 22: kernel averages of the point-cloud ICP iteration at 50k, grid search off / on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in 0 1; do
  export GINGR_NN_GRID=$g
  rocprofv3 --kernel-trace --stats -d $O/prof_grid$g -o p --output-format csv -- python3 $R/tools/bench_icp.py 50000 > $O/log$g.txt 2>&1
  python3 - $O/prof_grid$g <<'PY' > $O/kernels_grid$g.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f}')
PY
  echo "== grid $g"; cat $O/kernels_grid$g.txt
done

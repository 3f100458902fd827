#!/bin/bash
# round 3, step 37: ICP artefacts of the final tree: iteration rate (three runs) + rocprofv3 kernel statistics at 50k; surface ICP rate
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s37; mkdir -p $O; cd $R
for k in 1 2 3; do GINGR_BENCH_NN_COUNT=1 python3 tools/bench_icp.py 50000 2>/dev/null | tail -1 >> $O/icp50k_runs.txt; done
bash tools/prof_icp.sh 50000 > $O/prof.txt 2>&1; cp gpurun_out/prof_icp/kernel_stats.csv $O/icp50k_kernel_stats.csv; tail -16 $O/prof.txt | cut -c1-150
for k in 1 2 3; do python3 tools/bench_icp_surface.py 2>/dev/null | tail -1 >> $O/surface_runs.txt; done
cat $O/icp50k_runs.txt $O/surface_runs.txt | cut -c1-250

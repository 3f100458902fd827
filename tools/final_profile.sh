#!/bin/bash
# Round-end evidence: the default bench line, the rocprofv3 kernel statistics of the same command, SQ counters.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
cd $R && python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/stats -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/stats.log 2>&1
cd $R && bash tools/pmc_sq.sh > gpurun_out/final/pmc.log 2>&1
tail -c 600 gpurun_out/final/bench.json

#!/bin/bash
# round 3, step 15: what issues in the shadow of the float64 MFMA (microbenchmark)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s15; mkdir -p $O; cd $R
timeout 120 tools/bin/ubench_fill > $O/ubench_fill.txt 2>&1; cat $O/ubench_fill.txt

#!/bin/bash
# round 3, step 25: grid search for the nearest target vertex of the SURFACE ICP too: tests, then surface ICP and MH chain rates, grid off / on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s25; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_surface_icp.py tests/test_gpu_sampling.py tests/test_gpu_probabilistic.py tests/test_gpu_nn_grid.py tests/test_gpu_structured_inputs.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for g in 0 1 0 1; do GINGR_NN_GRID=$g python3 tools/bench_icp_surface.py 2>/dev/null | tail -1 | cut -c1-300 >> $O/surface.txt; done
for g in 0 1 0 1; do GINGR_NN_GRID=$g python3 tools/bench_mh_chain.py 2>/dev/null | tail -1 | cut -c1-300 >> $O/chain.txt; done
cat $O/surface.txt $O/chain.txt

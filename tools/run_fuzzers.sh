#!/bin/bash
# The long versions of the randomised parity sweeps (tests/fuzz_*.py), one summary line each -> gpurun_out/fuzz_summary.txt
# (copied to profiles/rNN_fuzz_summary.txt).  About five minutes on a GPU box.
R=${GRAFT_REPO_ROOT:-.}
cd $R
export PYTHONPATH=$R
OUT=$R/gpurun_out/fuzz_summary.txt
mkdir -p $R/gpurun_out
: > $OUT
run() {
  local t0=$(date +%s)
  local line
  line=$(timeout 1500 python3 tests/$1 $2 $3 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostn\|^Librccl\|amdgpu.ids" | tail -1)
  echo "$1 $2 cases seed $3 ($(( $(date +%s) - t0 )) s): $line" | tee -a $OUT
}
run fuzz_cpd_stats.py 200 101
run fuzz_parity.py 200 102
run fuzz_surface.py 150 103
run fuzz_tri_grid.py 150 104
run fuzz_wide_rank.py 150 105
run fuzz_model_setup.py 200 106
run fuzz_model_setup.py 120 107

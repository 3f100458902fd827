#!/bin/bash
# The long versions of the randomised parity sweeps (tests/fuzz_*.py), one summary line each -> gpurun_out/fuzz_summary.txt
# (copied to profiles/rNN_fuzz_summary.txt).  About five minutes on a GPU box.  SEED_BASE=<n> (default 100) shifts every seed: another
# n gives another independent sweep of the same length.
R=${GRAFT_REPO_ROOT:-.}
cd $R
export PYTHONPATH=$R
OUT=$R/gpurun_out/fuzz_summary.txt
mkdir -p $R/gpurun_out
: > $OUT
B=${SEED_BASE:-100}
run() {
  local t0=$(date +%s)
  local line
  line=$(timeout 1500 python3 tests/$1 $2 $3 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostn\|^Librccl\|amdgpu.ids" | tail -1)
  echo "$1 $2 cases seed $3 ($(( $(date +%s) - t0 )) s): $line" | tee -a $OUT
}
run fuzz_cpd_stats.py 200 $((B + 1))
run fuzz_parity.py 200 $((B + 2))
run fuzz_surface.py 150 $((B + 3))
run fuzz_tri_grid.py 150 $((B + 4))
run fuzz_wide_rank.py 150 $((B + 5))
run fuzz_model_setup.py 200 $((B + 6))
run fuzz_model_setup.py 120 $((B + 7))

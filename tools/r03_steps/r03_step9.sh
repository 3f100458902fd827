#!/bin/bash
# round 3, step 9: where does the post-solve kernel's time go (stamps build), at femur size and on an 8-rank shard of 50k; gpu tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s9; mkdir -p $O; cd $R
GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_stamps.so python3 bench.py --points 1622 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --roofline-steps 0 2>&1 | grep "post_solve stamps" | tail -4 > $O/stamps_1622.txt; cat $O/stamps_1622.txt
GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_stamps.so python3 bench.py --emulate-world 8 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --roofline-steps 0 2>&1 | grep "post_solve stamps" | tail -4 > $O/stamps_emu8.txt; cat $O/stamps_emu8.txt
GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_stamps.so python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --roofline-steps 0 2>&1 | grep "post_solve stamps" | tail -4 > $O/stamps_50k.txt; cat $O/stamps_50k.txt
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt

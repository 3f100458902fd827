#!/bin/bash
# round 3, step 13: where do the steps of the Gram kernel spend their time (stamps build: core-clock cycles per wave)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s13; mkdir -p $O; cd $R
for a in "" "--emulate-world 8" "--points 1622"; do
  n=$(echo "$a" | tr -d ' -'); n=${n:-50k}
  GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_gstamps.so timeout 600 python3 bench.py $a --steps 4 --warmup 2 --no-cpu-baseline --no-parity-check --roofline-steps 0 2>&1 | grep "gram stamps" | tail -16 > $O/stamps_$n.txt
  echo "== $n"; cat $O/stamps_$n.txt
done

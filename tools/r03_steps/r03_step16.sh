#!/bin/bash
# round 3, step 16: knock-out variants of the Gram loop (wrong results, timing only): which part of a step costs what
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s16; mkdir -p $O; cd $R
bash tools/abn.sh "cur=" "noloads=libgingr_hip_ko1.so" "nobarrier=libgingr_hip_ko2.so" "noldsread=libgingr_hip_ko4.so" "noldswrite=libgingr_hip_ko8.so" "novalu=libgingr_hip_ko16.so" "mfmaonly=libgingr_hip_ko31.so" "ko13=libgingr_hip_ko13.so" -- --steps 10 --warmup 3 > $O/ko50k.txt 2>&1; cat $O/ko50k.txt

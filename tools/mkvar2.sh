#!/bin/bash
# build-time variant of TWO translation units: tools/mkvar2.sh NAME unitA "-D..." unitB "-D..."  ->  gingr_amd/libgingr_hip_NAME.so
set -e
cd /root/repo/gingr_amd/csrc
NAME=$1; UA=$2; DA=$3; UB=$4; DB=$5
mkdir -p build_$NAME
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result"
/opt/rocm/bin/hipcc $F $DA -c $UA.hip -o build_$NAME/$UA.o &
/opt/rocm/bin/hipcc $F $DB -c $UB.hip -o build_$NAME/$UB.o &
wait
OBJS=$(for o in context affinity nn_grid gp gp_wide eig fitter group rccl_exchange gpmm surface classic_cpd rigid_icp; do if [ $o = $UA ] || [ $o = $UB ]; then echo build_$NAME/$o.o; else echo $o.o; fi; done)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libgingr_hip_$NAME.so $OBJS -lpthread -ldl
echo built $NAME

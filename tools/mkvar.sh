#!/bin/bash
# fast build-time variant of ONE translation unit (the other objects are the main build's): tools/mkvar.sh NAME unit "-D..."  ->  gingr_amd/libgingr_hip_NAME.so
set -e
cd /root/repo/gingr_amd/csrc
NAME=$1; UNIT=$2; DEFS=$3
mkdir -p build_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result $DEFS -c $UNIT.hip -o build_$NAME/$UNIT.o
OBJS=$(for o in context affinity nn_grid gp gp_wide eig fitter group rccl_exchange gpmm surface classic_cpd rigid_icp; do if [ $o = $UNIT ]; then echo build_$NAME/$o.o; else echo $o.o; fi; done)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libgingr_hip_$NAME.so $OBJS -lpthread -ldl
echo built $NAME

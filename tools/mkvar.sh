#!/bin/bash
# fast variant: only affinity.hip is recompiled with the given defines, the other objects are the main build's
# usage: mkvar.sh NAME "-DGINGR_...=v ..."
set -e
cd /root/repo/gingr_amd/csrc
NAME=$1; DEFS=$2
mkdir -p build_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result $DEFS -c affinity.hip -o build_$NAME/affinity.o
OBJS=$(for o in context nn_grid gp fitter group rccl_exchange gpmm surface classic_cpd rigid_icp; do echo $o.o; done)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libgingr_hip_$NAME.so build_$NAME/affinity.o $OBJS -lpthread -ldl
echo built $NAME

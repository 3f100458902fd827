#!/bin/bash
# Device ISA of one translation unit of the library: tools/isa.sh affinity [extra -D flags] -> /tmp/<name>.s, then the register / scratch /
# occupancy lines of the kernels whose mangled name matches $KERNELS (regular expression; default: all).
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -S --cuda-device-only "$@" \
  $R/gingr_amd/csrc/$name.hip -o /tmp/$name.s 2>&1 | grep -v "hip-link" 
awk -v k="${KERNELS:-.}" '/^_Z.*:/{name=$1} /; NumVgprs:/{v=$3} /; ScratchSize:/{s=$3} /; Occupancy:/{ if (name ~ k) printf "%-110s vgpr %s scratch %s occupancy %s\n", substr(name,1,110), v, s, $3 }' /tmp/$name.s

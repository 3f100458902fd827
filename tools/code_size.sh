#!/bin/bash
# ISA bytes of the device functions of one HIP source (kernels and out-of-line callees), largest last.
# usage: tools/code_size.sh gingr_amd/csrc/gp.hip [name filter]
SRC=$1; PAT=${2:-.}
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -c --cuda-device-only -o $T/a.o $SRC 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/a.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/a.co
/opt/rocm/lib/llvm/bin/llvm-readelf -sW $T/a.co | awk '$4=="FUNC" {print $3, $8}' | sort -n | uniq | while read sz nm; do
  d=$(echo $nm | c++filt | cut -c1-110); if echo "$d" | grep -qE "$PAT"; then echo "$sz $d"; fi
done
rm -rf $T

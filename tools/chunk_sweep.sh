#!/bin/bash
# Launch-round experiment: iteration time for forced chunk counts of the two all-pairs passes.  The counts are BUILD-time knobs
# (-DGINGR_COLSUM_CHUNKS / -DGINGR_ROWSTATS_CHUNKS, affinity.hip): every count is its own library, built here (needs hipcc on the box).
# usage: tools/chunk_sweep.sh "<colsum counts>" "<rowstats counts>" [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
CS=$1; RS=$2; shift 2
one() {  # name defs kernel
  make -s -C $R/gingr_amd/csrc variant NAME=$1 DEFS="$2" >/dev/null || { echo "build of $1 failed"; return; }
  GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_$1.so python3 $R/bench.py --no-cpu-baseline --no-parity-check "${@:4}" 2>/dev/null | tail -1 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k={x['kernel']:x['avg_ms'] for x in d['kernels']};print('$1: $3 %.4f ms  step %.4f' % (k['$3'], d['ms_per_step']))"
}
for c in $CS; do one colsum$c "-DGINGR_COLSUM_CHUNKS=$c" cpd_colsum_kernel "$@"; done
for c in $RS; do one rowstats$c "-DGINGR_ROWSTATS_CHUNKS=$c" cpd_rowstats_kernel "$@"; done

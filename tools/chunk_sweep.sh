#!/bin/bash
# Launch-round experiment: iteration time for forced chunk counts of the two all-pairs passes (developer knobs
# GINGR_COLSUM_CHUNKS / GINGR_ROWSTATS_CHUNKS); usage: tools/chunk_sweep.sh "<colsum counts>" "<rowstats counts>" [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
CS=$1; RS=$2; shift 2
for c in $CS; do
  GINGR_COLSUM_CHUNKS=$c python3 $R/bench.py --no-cpu-baseline --no-parity-check "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k={x['kernel']:x['avg_ms'] for x in d['kernels']};print('colsum chunks $c: colsum %.4f ms  step %.4f' % (k['cpd_colsum_kernel'], d['ms_per_step']))"
done
for c in $RS; do
  GINGR_ROWSTATS_CHUNKS=$c python3 $R/bench.py --no-cpu-baseline --no-parity-check "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k={x['kernel']:x['avg_ms'] for x in d['kernels']};print('rowstats chunks $c: rowstats %.4f ms  step %.4f' % (k['cpd_rowstats_kernel'], d['ms_per_step']))"
done

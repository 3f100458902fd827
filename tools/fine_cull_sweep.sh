#!/bin/bash
# late-regime (sigma2 small against the clouds) cost of the two culling variants of the all-pairs kernels over problem sizes
for P in ${SIZES:-15000 30000 50000}; do for F in 0 1; do
  r=$(python bench.py --ctx-option fine_cull=$F --points $P --sigma2 ${S2:-4} --steps 20 --warmup 3 --roofline-steps 3 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k['kernel'][:12]: round(k['avg_ms']*1e3,1) for k in d['kernels'][:2]})")
  echo "points=$P fine=$F sigma2_0=${S2:-4}: $r"
done; done

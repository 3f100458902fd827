#!/bin/bash
# femur-sized workload: where do the all-pairs kernels spend their time?  (developer sweep)
run() { python bench.py --points ${P:-1622} --steps 300 --warmup 10 --roofline-steps 3 --no-cpu-baseline --no-parity-check "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k['kernel'][:12]: round(k['avg_ms']*1e3,1) for k in d['kernels'][:2]}, 'sigma2', round(d['sigma2_after_timed_steps'],3))"; }
echo "default            : $(run)"
echo "GINGR_FINE_CULL=0  : $(GINGR_FINE_CULL=0 run)"
echo "GINGR_FINE_CULL=1  : $(GINGR_FINE_CULL=1 run)"
echo "GINGR_CULL=0       : $(GINGR_CULL=0 run)"
echo "steps=5 (early)    : $(python bench.py --points ${P:-1622} --steps 5 --warmup 0 --roofline-steps 3 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k['kernel'][:12]: round(k['avg_ms']*1e3,1) for k in d['kernels'][:2]}, 'sigma2', round(d['sigma2_after_timed_steps'],3))")"

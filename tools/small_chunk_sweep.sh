#!/bin/bash
# femur-sized workload: where do the all-pairs kernels spend their time?  (developer sweep)
run() { python bench.py --points ${P:-1622} --steps 300 --warmup 10 --roofline-steps 3 --no-cpu-baseline --no-parity-check "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k['kernel'][:12]: round(k['avg_ms']*1e3,1) for k in d['kernels'][:2]}, 'sigma2', round(d['sigma2_after_timed_steps'],3))"; }
echo "default            : $(run)"
echo "fine_cull=0        : $(run --ctx-option fine_cull=0)"
echo "fine_cull=1        : $(run --ctx-option fine_cull=1)"
echo "cull=0             : $(run --ctx-option cull=0)"
echo "steps=5 (early)    : $(python bench.py --points ${P:-1622} --steps 5 --warmup 0 --roofline-steps 3 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k['kernel'][:12]: round(k['avg_ms']*1e3,1) for k in d['kernels'][:2]}, 'sigma2', round(d['sigma2_after_timed_steps'],3))")"

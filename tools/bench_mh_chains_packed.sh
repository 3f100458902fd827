#!/bin/bash
# BASELINE config 5 ("8 chains"): independent Metropolis-Hastings chains are replicas.  One chain keeps an MI355X busy for ~0.54 ms of
# small kernels per 0.78 ms step, so several chains can SHARE one GPU: this runs N chain processes concurrently on device 0 and
# prints the aggregate rate.   usage: tools/bench_mh_chains_packed.sh "1 2 4 8" [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
STEPS=${2:-600}
for n in $1; do
  rm -f /tmp/chain_*.json
  t0=$(date +%s.%N)
  for i in $(seq 1 $n); do python3 $R/tools/bench_mh_chain.py $STEPS 2>/dev/null | tail -1 > /tmp/chain_$i.json & done
  wait
  t1=$(date +%s.%N)
  python3 - "$n" "$STEPS" "$t0" "$t1" <<'PY'
import json, sys, glob
n, steps, t0, t1 = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
rates = [json.load(open(f))["steps_per_s"] for f in sorted(glob.glob("/tmp/chain_*.json"))]
print(json.dumps({"chains_on_one_gpu": n, "steps_per_chain": steps, "per_chain_steps_per_s": [round(r, 1) for r in rates],
                  "aggregate_steps_per_s_in_chain_loops": round(sum(rates), 1), "wall_s_including_start_up": round(t1 - t0, 2)}))
PY
done

#!/bin/bash
# Power / clock read-outs (rocm-smi, read-only) while the metric workload runs for a few seconds: is the pair loops' clock a power limit?
R=$GRAFT_REPO_ROOT
cd $R
rocm-smi --showpower --showclocks --showperflevel 2>&1 | grep -v "^$" | head -30
echo "=== running"
python3 bench.py --steps ${1:-3000} --warmup 20 --no-cpu-baseline --no-parity-check --roofline-steps 0 --sustained-steps 0 ${@:2} > gpurun_out/power_bench.json 2> /dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo
  sleep 0.7
done
wait $BP
tail -1 gpurun_out/power_bench.json | cut -c1-200
rocm-smi --showmaxpower 2>&1 | grep -i -E "max|power" | head -5

#!/bin/bash
# Round-end evidence in one gpurun call: the default bench line, the rocprofv3 kernel statistics of the same command, the PMC
# traffic (separate FETCH_SIZE / WRITE_SIZE passes), the SQ counters and the emulated per-rank shard times.  Everything lands in
# gpurun_out/final/; copy what is to be judged to profiles/rNN_* (tools/collect_profiles.py).  The PMC passes come first and leave their
# artefacts under profiles/ of the box's copy, so the bench line of THIS run quotes the traffic and matrix-pipe counters of THIS box.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final
rm -rf $F; mkdir -p $F
cd $R
bash tools/pmc_traffic.sh > $F/pmc_traffic.txt 2>&1 && cp gpurun_out/pmc_traffic.json $F/pmc_traffic.json && cp gpurun_out/pmc_traffic.json profiles/r04_pmc_traffic.json
bash tools/pmc_sq.sh > $F/pmc_sq.txt 2>&1; cp gpurun_out/pmc_mfma.json $F/pmc_mfma.json; cp gpurun_out/pmc_mfma.json profiles/r04_pmc_mfma.json
python3 bench.py 2> $F/bench.err | tail -1 > $F/bench.json
bash tools/prof_stats.sh final > $F/kernel_stats.txt 2>&1; cp gpurun_out/prof_final/kernel_stats.csv $F/kernel_stats.csv
for n in 1 2 4 8; do python3 bench.py --emulate-world $n --no-cpu-baseline --no-parity-check --steps 100 --warmup 10 --roofline-steps 0 2>/dev/null | tail -1 > $F/emu$n.json; done
python3 bench.py --sigma2 4 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 > $F/bench_sigma2_4.json
python3 bench.py --points 15000 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_15k.json
python3 bench.py --points 100000 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_100k.json
for n in 2 4 8; do python3 bench.py --points 100000 --emulate-world $n --no-cpu-baseline --no-parity-check --steps 40 --warmup 5 --roofline-steps 0 2>/dev/null | tail -1 > $F/emu100k_$n.json; done
python3 bench.py --group --logical-shards 2 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_group_logical2.json
timeout 1500 python3 tools/bench_configs.py > $F/configs.txt 2> $F/configs.err; cp gpurun_out/r04_configs.json $F/configs.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench.json"))
print("bench", d["value"], d["ms_per_step"], d["valid"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print({k["kernel"]: round(k["avg_ms"], 4) for k in d["kernels"]})
for n in (1, 2, 4, 8):
    print("emulated", n, json.load(open(f"gpurun_out/final/emu{n}.json"))["ms_per_step"])
print("sigma2=4", json.load(open("gpurun_out/final/bench_sigma2_4.json"))["ms_per_step"])
print("group x2 logical", json.load(open("gpurun_out/final/bench_group_logical2.json"))["ms_per_step"])
for t in ("15k", "100k"):
    d = json.load(open(f"gpurun_out/final/bench_{t}.json"))
    print(t, d["ms_per_step"], d["valid"], d["parity_check"] and d["parity_check"]["ok"])
for n in (2, 4, 8):
    print("emulated 100k", n, json.load(open(f"gpurun_out/final/emu100k_{n}.json"))["ms_per_step"])
PY

#!/bin/bash
# Round-end evidence in one gpurun call: the default bench line, the rocprofv3 kernel statistics of the same command, the PMC
# traffic (separate FETCH_SIZE / WRITE_SIZE passes), the SQ counters and the emulated per-rank shard times.  Everything lands in
# gpurun_out/final/; copy what is to be judged to profiles/rNN_* (tools/collect_profiles.py).  The PMC passes come first and leave their
# artefacts under profiles/ of the box's copy, so the bench line of THIS run quotes the traffic and matrix-pipe counters of THIS box.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final
rm -rf $F; mkdir -p $F
cd $R
bash tools/pmc_traffic.sh > $F/pmc_traffic.txt 2>&1 && cp gpurun_out/pmc_traffic.json $F/pmc_traffic.json && cp gpurun_out/pmc_traffic.json profiles/r06_pmc_traffic.json
bash tools/pmc_sq.sh > $F/pmc_sq.txt 2>&1; cp gpurun_out/pmc_mfma.json $F/pmc_mfma.json; cp gpurun_out/pmc_mfma.json profiles/r06_pmc_mfma.json
python3 bench.py 2> $F/bench.err | tail -1 > $F/bench.json
bash tools/prof_stats.sh final > $F/kernel_stats.txt 2>&1; cp gpurun_out/prof_final/kernel_stats.csv $F/kernel_stats.csv
# model ranks above 112 (the wide Gram / super-panel solve / rank-256 fit pass): bench lines and the kernel statistics of the rank-256 run
for r in 128 200 256 512; do python3 bench.py --rank $r --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_rank$r.json; done
bash tools/pmc_wide.sh > $F/pmc_wide_r256.txt 2>&1; cp gpurun_out/pmc_wide_r256.json $F/pmc_wide_r256.json
bash tools/prof_stats.sh final_r256 --rank 256 > $F/kernel_stats_r256.txt 2>&1; cp gpurun_out/prof_final_r256/kernel_stats.csv $F/kernel_stats_r256.csv
for n in 1 2 4 8; do python3 bench.py --emulate-world $n --no-cpu-baseline --no-parity-check --steps 100 --warmup 10 --roofline-steps 0 --sustained-steps 0 2>/dev/null | tail -1 > $F/emu$n.json; done
python3 bench.py --sigma2 4 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 > $F/bench_sigma2_4.json
python3 bench.py --points 15000 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_15k.json
python3 bench.py --points 100000 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_100k.json
for n in 2 4 8; do python3 bench.py --points 100000 --emulate-world $n --no-cpu-baseline --no-parity-check --steps 40 --warmup 5 --roofline-steps 0 --sustained-steps 0 2>/dev/null | tail -1 > $F/emu100k_$n.json; done
python3 bench.py --group --logical-shards 2 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_group_logical2.json
timeout 1500 python3 tools/bench_configs.py > $F/configs.txt 2> $F/configs.err; cp gpurun_out/r06_configs.json $F/configs.json
# the 8-rank shard's kernels (the Amdahl table of DESIGN.md section 7), the Metropolis-Hastings chain and the two ICP flavours
bash tools/prof_stats.sh final_emu8 --emulate-world 8 --steps 100 --warmup 10 --roofline-steps 0 > $F/emu8_kernels.txt 2>&1; cp gpurun_out/prof_final_emu8/kernel_stats.csv $F/emu8_kernel_stats.csv
python3 bench.py --points 1622 --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $F/bench_1622.json
for i in 1 2 3 4 5; do python3 tools/bench_mh_chain.py 300 0 2>/dev/null | tail -1 > $F/chain_$i.json; done
python3 tools/bench_mh_chain.py 300 0 nofuse 2>/dev/null | tail -1 > $F/chain_call_by_call.json
bash tools/prof_mh_chain.sh final_chain 300 0 > $F/chain_kernels.txt 2>&1; cp gpurun_out/prof_final_chain/kernel_stats.csv $F/chain_kernel_stats.csv
python3 tools/bench_icp.py 2>/dev/null | tail -1 > $F/icp50k.json
python3 tools/bench_icp_surface.py 2>/dev/null | tail -1 > $F/icp_surface.json
python3 tools/bench_icp_surface.py 6 n=50 2>/dev/null | tail -1 > $F/icp_surface_n50.json
for fl in 2 1; do python3 tools/bench_reversed_shard.py 8 $fl 2>/dev/null | grep "^{" | tail -1 > $F/reversed_shard8_flavour$fl.json; done
# model set-up (one-off): on-device GPMM builds, the whole-registration timeline, the landmark iteration, the register eigen kernel
python3 tools/bench_gpmm.py 2>/dev/null | tail -1 > $F/gpmm_build.json
bash tools/experiments/prof_tool.sh final_gpmm tools/bench_gpmm.py > $F/gpmm_kernels.txt 2>&1; cp gpurun_out/prof_final_gpmm/kernel_stats.csv $F/gpmm_kernel_stats.csv
python3 tools/experiments/registration_timeline.py 2>/dev/null | grep "^rep" > $F/registration_timeline.txt
bash tools/experiments/prof_landmarks.sh > $F/landmarks_kernels.txt 2>&1; cp gpurun_out/prof_landmarks/kernel_stats.csv $F/landmarks_kernel_stats.csv
if [ -x tools/bin/ubench_sym_eig ]; then tools/bin/ubench_sym_eig > $F/ubench_sym_eig.txt 2>&1; fi
# per-wave stamps of the two pair loops on the 8-rank shard (diagnostic build: make -C gingr_amd/csrc variant NAME=stamps DEFS=-DGINGR_STAMPS)
if [ -f gingr_amd/libgingr_hip_stamps.so ]; then GINGR_HIP_LIB=$R/gingr_amd/libgingr_hip_stamps.so python3 tools/stamps_shard.py 8 50000 2>/dev/null | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostn\|^Librccl" > $F/stamps_emu8.txt; fi
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench.json"))
print("bench", d["value"], d["ms_per_step"], d["valid"], d["reason"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print("sustained", d["sustained"])
for r in (128, 200, 256, 512):
    try:
        dr = json.load(open(f"gpurun_out/final/bench_rank{r}.json"))
        print("rank", r, dr["ms_per_step"], dr["valid"], {k["kernel"]: round(k["avg_ms"], 4) for k in dr["kernels"]})
    except Exception as ex:
        print("rank", r, "failed", ex)
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
d16 = json.load(open("gpurun_out/final/bench_1622.json"))
print("1622", d16["ms_per_step"], d16["valid"], d16["reason"], d16["timing"])
print("chain steps/s", [round(json.load(open(f"gpurun_out/final/chain_{i}.json"))["steps_per_s"]) for i in range(1, 6)],
      "call by call", round(json.load(open("gpurun_out/final/chain_call_by_call.json"))["steps_per_s"]))
for t in ("icp50k", "icp_surface"):
    try:
        print(t, json.load(open(f"gpurun_out/final/{t}.json")).get("ms_per_iteration"))
    except Exception as ex:
        print(t, "failed", ex)
for c in json.load(open("gpurun_out/final/configs.json")):
    print("config", c["config"], c["value"], c["unit"], c["ms_per_step"], c["valid"])
PY

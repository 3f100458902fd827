"""The contract's launch line for N > 1 -- `python bench.py --gpus N` starting one rank per GPU under torch.distributed.run -- run
end to end on a ONE-GPU box: GINGR_BENCH_SHARED_DEVICE=1 puts every rank on device 0 and sends the exchange through gloo on host
copies (RCCL refuses two ranks on one device).  Everything else is the code the driver's scaling run executes: rank environment,
shard split, phase / exchange ordering, max-over-ranks clock, only rank 0 printing, the JSON line last on stdout, and the
`shard_consistency` check (the sharded run must end in the state ONE shard reaches from the same start)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,points", [(2, 6000), (3, 5003)])
def test_bench_launch_line_with_ranks_sharing_the_device(world, points):
    env = dict(os.environ, GINGR_BENCH_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "1",
                        "--points", str(points), "--rank", "40", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    out = json.loads(lines[-1])                      # the JSON line is the LAST line, printed once (rank 0 only)
    assert sum(1 for ln in lines if ln.startswith('{"metric"')) == 1
    assert out["n_gpus"] == world and out["config"]["parallelism"] == f"row-shard x{world}"
    sc = out["shard_consistency"]
    assert sc is not None and sc["ok"], sc
    assert sc["iterations"] == 1 + 3                # the state both sides are compared in: warm-up + roofline steps from sigma2_0
    assert sc["sigma2_rel"] < 1e-10 and sc["alpha_max_abs"] < 1e-9, sc
    assert out["valid"] is True
    # who took part and what the exchanges cost: the fields a first run on real multi-GPU hardware is read by
    rr = out["rccl_ranks"]
    assert rr["world_size"] == world and len(rr["ranks"]) == world and sorted(g["rank"] for g in rr["ranks"]) == list(range(world))
    assert rr["backend"] == "gloo" and rr["distinct_device_uuids"] == 1        # this test: every rank on device 0
    ex = out["exchange"]
    assert ex["segment0_column_sums_ms"] > 0.0 and ex["segment1_gram_bundle_ms"] > 0.0
    assert ex["bytes"]["segment0"] == 8 * points
    assert ex["path"].startswith("gloo")            # two ranks on ONE device: RCCL cannot run here, the host-driven exchange does
    # the same workload once more through the in-library device group, started by rank 0 after the ranks are done (one SCALE record
    # then carries RCCL and the group side by side)
    gm = out["group_mode"]
    assert "error" not in gm, gm
    assert gm["valid"] is True and gm["n_gpus"] == world and gm["ms_per_step"] > 0.0
    assert gm["exchange"]["segment0_column_sums_ms"] > 0.0 and gm["exchange"]["path"].startswith("in-library device group")
    # round 6: every rank's own exchange times next to the two pair loops it ran, and the two variant runs behind the headline (the
    # emulated shard without exchange; the split column-sum exchange -- here through gloo, where the option changes nothing but must run)
    pr = ex["per_rank"]
    assert sorted(p["rank"] for p in pr) == list(range(world))
    assert all(p["segment0_column_sums_ms"] > 0.0 and p["cpd_colsum_kernel_ms"] > 0.0 and p["cpd_rowstats_kernel_ms"] > 0.0 for p in pr), pr
    assert ex["segment0_column_sums_ms"] == pytest.approx(max(p["segment0_column_sums_ms"] for p in pr))
    va = out["variants"]
    assert "error" not in va["emulated_shard_no_exchange"], va
    assert va["emulated_shard_no_exchange"]["ms_per_step"] > 0.0 and va["emulated_shard_no_exchange"]["valid"] is False   # (never a registration)
    assert "error" not in va["split_exchange"], va
    assert va["split_exchange"]["valid"] is True and va["split_exchange"]["n_gpus"] == world


def test_bench_native_rccl_exchange_with_a_one_rank_communicator():
    """`bench.py --force-dist` on one GPU: the N > 1 code path with a ONE-rank RCCL communicator owned by the library's context --
    unique id, ncclCommInitRank, the one-off moment all-reduce and gingr_fitter_update_cpd_rccl_async (ncclAllReduce enqueued by the
    library on the kernels' stream between the phases).  The measured state must pass the oracle parity check of the line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GINGR_BENCH_SHARED_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "4", "--warmup", "1", "--points", "6000",
                        "--rank", "40", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    ex = out["exchange"]
    assert ex["path"] == "rccl-native"
    assert ex["rccl"]["world"] == 1 and ex["rccl"]["rank"] == 0 and ex["rccl"]["version"] > 0 and "rccl" in ex["rccl"]["library"]
    assert out["valid"] is True and out["parity_check"]["ok"], out["parity_check"]
    assert "one process per GPU, rccl-native" in out["config"]["exchange"]


def test_bench_group_mode_reports_the_exchange():
    """`bench.py --group --logical-shards 2`: the in-library device group prints the same two fields."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--group", "--logical-shards", "2", "--steps", "4", "--warmup", "1",
                        "--points", "6000", "--rank", "40", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    rr = out["rccl_ranks"]
    assert rr["world_size"] == 2 and rr["distinct_devices"] == 1 and rr["distinct_device_uuids"] == 1
    ex = out["exchange"]
    assert ex["segment0_column_sums_ms"] > 0.0 and ex["segment1_gram_bundle_ms"] > 0.0

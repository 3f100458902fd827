"""Self-validating multi-GPU tests: they SKIP on the one-GPU boxes of the pool and make the first run on a multi-GPU node diagnose
itself -- real RCCL ranks on distinct devices through bench.py's launch line, a two-process RCCL registration from plain C (the
ncclUniqueId handed over through a file, as a JVM host would), and the in-library device group over two physical devices
(tests/test_gpu_group.py::test_two_physical_devices_equal_one_device).  The one-rank form of the C program runs everywhere.
Reference shape: one GingrAlgorithm.update per iteration (G/api/GingrAlgorithm.scala:192-254); the sharding is this repository's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices() -> int:
    from gingr_amd import _native as nat
    return int(nat.load().gingr_device_count())


def _build_rank_program(tmp_path) -> str:
    exe = str(tmp_path / "cabi_rccl_rank")
    libdir = os.path.join(ROOT, "gingr_amd")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "cabi_rccl_rank.c"),
                           "-o", exe, "-L", libdir, "-lgingr_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _case(tmp_path, n_iter=4):
    rng = np.random.default_rng(5)
    ref = rng.normal(0, 30, (900, 3))
    mo = go.build_gaussian_gpmm(ref, 40.0, 20.0, rel_tol=1e-9, max_rank=20)
    target = mo.instance(rng.normal(0, 0.7, mo.rank)) @ go.euler_to_rot(0.02, -0.01, 0.015).T + np.array([0.5, -0.3, 0.2])
    target = target[rng.permutation(target.shape[0])[:850]] + rng.normal(0, 0.3, (850, 3))
    sigma2, w = go.cpd_initial_sigma2(mo.ref + mo.mean, target), 0.1
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        np.array([mo.M, target.shape[0], mo.rank, 0, 0, n_iter], dtype=np.int64).tofile(f)
        for a in (mo.ref, mo.mean, np.asfortranarray(mo.U).ravel(order="F"), mo.lam, target, np.array([sigma2, w])):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    st = go.initial_state(mo, sigma2)
    for _ in range(n_iter):
        st = go.cpd_update(mo, target, st, w=w)
    return mo, str(inp), st


def _parse(path):
    res = {}
    for line in open(path):
        name, *vals = line.split()
        res[name] = vals
    return res


def _run_ranks(exe, world, devices, inp, tmp_path):
    idfile = str(tmp_path / f"nccl_id_{world}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, str(r), str(world), str(devices[r]), idfile, inp, str(tmp_path / f"out_{world}_{r}.txt")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (_, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    return [_parse(tmp_path / f"out_{world}_{r}.txt") for r in range(world)]


def test_plain_c_rank_program_with_a_one_rank_communicator(tmp_path):
    """The C rank program end to end on ONE GPU (world 1): id through a file, communicator, moment all-reduce, fused RCCL update."""
    exe = _build_rank_program(tmp_path)
    mo, inp, st = _case(tmp_path)
    (got,) = _run_ranks(exe, 1, [0], inp, tmp_path)
    assert got["rccl"][0] == "1" and got["rccl"][1] == "0" and int(got["rccl"][2]) > 0
    fit = np.array([float(v) for v in got["fit"]]).reshape(-1, 3)
    assert np.linalg.norm(fit - st.fit) / np.linalg.norm(st.fit) < 1e-5
    assert abs(float(got["scalars"][7]) - st.sigma2) < 1e-8 * st.sigma2 and int(got["scalars"][8]) == 4 and int(got["scalars"][9]) == 0


def test_two_processes_two_devices_rccl_from_plain_c(tmp_path):
    """Two PROCESSES, one GPU each, no Python and no torch inside them: the library's own RCCL exchange over xGMI."""
    if _devices() < 2:
        pytest.skip("one GPU visible: two RCCL ranks need two devices")
    exe = _build_rank_program(tmp_path)
    mo, inp, st = _case(tmp_path)
    got = _run_ranks(exe, 2, [0, 1], inp, tmp_path)
    fits = []
    for r, g in enumerate(got):
        assert g["rccl"][0] == "2" and int(g["rccl"][1]) == r
        fits.append(np.array([float(v) for v in g["fit"]]).reshape(-1, 3))
        assert int(g["scalars"][8]) == 4 and int(g["scalars"][9]) == 0
    # the replicated state is bit-identical on both ranks (same sums in the same order), the rows together are the oracle's fit
    assert got[0]["alpha"] == got[1]["alpha"] and got[0]["scalars"] == got[1]["scalars"]
    fit = np.concatenate(fits)
    assert np.linalg.norm(fit - st.fit) / np.linalg.norm(st.fit) < 1e-5


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_launch_line_on_distinct_devices(world):
    """`python bench.py --gpus N` as the driver's scaling run launches it, on real devices: the library's native RCCL exchange, one
    distinct device per rank, the sharded state equal to the single-shard replay, and the device-group child run valid."""
    if _devices() < world:
        pytest.skip(f"{_devices()} GPU(s) visible: needs {world}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GINGR_BENCH_SHARED_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--points", "6000",
                        "--rank", "40", "--sustained-steps", "0", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == world and out["valid"] is True, out.get("reason")
    assert out["rccl_ranks"]["world_size"] == world and out["rccl_ranks"]["distinct_device_uuids"] == world
    assert out["exchange"]["path"] == "rccl-native" and out["exchange"]["rccl"]["world"] == world
    assert out["shard_consistency"]["ok"], out["shard_consistency"]
    gm = out["group_mode"]
    assert "error" not in gm and gm["valid"] is True and gm["distinct_device_uuids"] == world, gm


def test_config5_one_chain_per_gpu():
    """BASELINE config 5 as stated: N chain processes pinned one per GPU (`bench.py --config 5 --gpus N`)."""
    n = _devices()
    if n < 2:
        pytest.skip("one GPU visible: the one-chain-per-GPU launcher needs two (the packed form runs in tools/bench_configs.py)")
    n = min(n, 8)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--gpus", str(n)], cwd=ROOT, capture_output=True, text=True,
                       timeout=3000)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][-1])
    ch = out["chains_one_per_gpu"]
    assert ch["chains"] == n and ch["distinct_device_uuids"] == n and len(ch["per_chain_steps_per_s"]) == n
    assert all(s != 3 for s in ch["statuses"]) and ch["loops_overlap_fraction"] > 0.5

#!/usr/bin/env python3
"""bench.py -- GiNGR update iterations/second, CPD 50k <-> 50k, on N MI355X (one process per GPU).

A "step" is ONE full update iteration (GingrAlgorithm.update + the fit refresh) of the CPD configuration on the
synthetic 50k <-> 50k workload of SURVEY.md section 8d: both all-pairs passes over the 50 000 x 50 000 affinity,
the weighted Gram + posterior solve, two coefficient projections, Umeyama and the new fit.  Model, target and state
are resident in HBM before the timed region.  With --gpus N the reference rows are sharded over N ranks and the
partial sums are all-reduced over RCCL (strong scaling: the problem is fixed).

Launch forms:
  python bench.py                      one GPU
  torchrun ... bench.py --gpus N       one process per GPU, exchange = torch.distributed all-reduce (RCCL over xGMI)
  python bench.py --gpus N             no torchrun around it: spawns exactly that torchrun command as a child process (before
                                       anything touches the GPU) and passes its JSON line through
  python bench.py --gpus N --group     ONE process, the in-library device group (gingr_group_*: worker thread per device,
                                       one-shot all-reduce over peer pointers) -- the path a C / JVM host uses
  python bench.py --group --logical-shards N    N logical shards on device 0 (protocol test / timing on a one-GPU box)

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the kernels' own stream in extra
iterations after the timed region; `parity_check` compares the affinity statistics and the sigma2 update of the state the
timed region ended in with the strict C oracle (outside the timed region; folded into `valid`); `cpu_baseline` times the
optimised C restatement (oracle/cpd_baseline.c, built here with -O3 -march=native -fopenmp) plus the numpy GP part on the
host cores on a bounded sample of the same workload.  oracle/ is the checker and the reported baseline -- never the product.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the oracle's OpenMP threads must sleep, not spin, once a CPU leg is over: spinning threads slow the thread that feeds the GPU
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

F64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector float64 peak (AMD datasheet)
F64_MFMA_PEAK_TFLOPS = 78.6   # matrix float64 peak
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md


# ------------------------------------------------------------------------------------------------ synthetic workload
def synth_clouds(n_points: int):
    """x_j ~ N(0, 50^2 I) (seed 1234, float32-rounded like a mesh file); y_i = x_pi(i) + N(0, 2^2 I) (seed 1235)."""
    x = np.random.default_rng(1234).normal(0.0, 50.0, (n_points, 3)).astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(1235)
    y = x[rng.permutation(n_points)] + rng.normal(0.0, 2.0, (n_points, 3))
    return y, x


def synth_gpmm(ref: np.ndarray, rank: int, sigma: float = 70.0, scaling: float = 50.0):
    """Rank-`rank` Gaussian-kernel GPMM over `ref` (kernel defaults of the femur demo,
    examples/DemoHelper/DemoDatasetLoader.scala:113-114): pivoted Cholesky of the scalar kernel + eigendecomposition,
    replicated per coordinate (DiagonalKernel).  Workload synthesis only (numpy, not timed)."""
    M = ref.shape[0]
    k = (rank + 2) // 3
    diag = np.full(M, scaling)
    cols = []
    for _ in range(k):
        p = int(np.argmax(diag))
        d = ref - ref[p]
        col = scaling * np.exp(-(d * d).sum(1) / (sigma * sigma))
        for c in cols:
            col = col - c * c[p]
        col = col / math.sqrt(diag[p])
        cols.append(col)
        diag = np.maximum(diag - col * col, 0.0)
    L = np.stack(cols, 1)
    ev, V = np.linalg.eigh(L.T @ L)
    order = np.argsort(ev)[::-1]
    ev, V = ev[order], V[:, order]
    Us = L @ V / np.sqrt(ev)[None, :]
    U = np.zeros((3 * M, 3 * k))
    lam = np.zeros(3 * k)
    for d in range(3):
        U[d::3, d::3] = Us
        lam[d::3] = ev
    return U[:, :rank].copy(order="F"), lam[:rank].copy()


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_gp_part(Q, ref_flat, P1, PX, fit, sigma2, lam_cpd, Binv_eps, rank):
    """The GP part of one update on the CPU with numpy / the host BLAS (identity pose, zero mean -- the first iteration of the
    workload): weighted Gram, posterior solve, posterior mean, two coefficient projections (the ridge system factored once per
    model: the optimised form), three instances, Umeyama.  Returns nothing: only its time is of interest."""
    M = P1.shape[0]
    wgt = P1 / (sigma2 * lam_cpd)
    yhat = PX / P1[:, None]
    e = (yhat.reshape(-1) - ref_flat)
    w3 = np.repeat(wgt, 3)
    Qw = Q * w3[:, None]
    G = Qw.T @ Q
    rhs = Qw.T @ e
    a = np.linalg.solve(np.eye(rank) + G, rhs)
    shape = ref_flat + Q @ a                                   # posterior mean
    alpha1 = Binv_eps @ (Q.T @ (shape - ref_flat))             # coefficients(), ridge system pre-factored
    newshape = (ref_flat + Q @ alpha1).reshape(M, 3)           # instance
    cur = ref_flat.reshape(M, 3)
    mu_a, mu_b = cur.mean(0), newshape.mean(0)                 # Umeyama (rigid)
    U_, _, Vt = np.linalg.svd((newshape - mu_b).T @ (cur - mu_a))
    R = U_ @ np.diag([1.0, 1.0, np.sign(np.linalg.det(U_ @ Vt))]) @ Vt
    t = mu_b - R @ mu_a
    alpha = Binv_eps @ (Q.T @ (((newshape - t) @ R).reshape(-1) - ref_flat))
    return (ref_flat + Q @ alpha).reshape(M, 3) @ R.T + t      # the new fit


def cpu_baseline(y, x, sigma2, w, model=None, budget_s=16.0):
    """CPU baseline of ONE update iteration on the host cores: the optimised C restatement of the two all-pairs passes
    (oracle/cpd_baseline.c: -O3 -march=native -fopenmp, SIMD exponential, built on this host) timed on a row sample of the SAME
    workload and scaled to all rows, PLUS the GP part (numpy / host BLAS) when the model is given.  Also reports the strict
    checker build (oracle/cpd_oracle.c: -O2 -ffp-contract=off, scalar libm exp) on a smaller sample, for reference."""
    from oracle import c_baseline as cb
    from oracle import c_oracle as co
    M, N = y.shape[0], x.shape[0]
    ys, xs = cb.soa(y), cb.soa(x)
    visible = os.cpu_count()
    cores = cb.calibrate_threads(ys, xs, sigma2)      # the thread count with the best pair rate on this host (CPU quota!)
    m0 = max(64, min(M, 8 * cores))
    cb.colsum(np.ascontiguousarray(ys[:, :m0]), xs, sigma2)                      # warm-up: thread pool, page faults
    t0 = time.perf_counter()
    cb.colsum(np.ascontiguousarray(ys[:, :m0]), xs, sigma2)
    per_row = (time.perf_counter() - t0) / m0
    ms = int(min(M, max(m0, budget_s / 2.2 / max(per_row, 1e-9))))
    ysub = np.ascontiguousarray(ys[:, :ms])
    t0 = time.perf_counter()
    den = cb.colsum(ysub, xs, sigma2) * (M / ms) + co.outlier_constant(M, N, sigma2, w)
    P1s, PXs = cb.rowstats(ysub, xs, sigma2, 1.0 / den)
    t_pairs = (time.perf_counter() - t0) * (M / ms)
    quota = cb.cpu_quota()
    sample = (f"both all-pairs passes (optimised C, {cores} OpenMP threads = best of 1,2,4,... up to the container's CPU quota "
              f"({'%.0f cores' % quota if quota else 'none'}; the host shows {visible} hardware threads)) on rows [0,{ms}) of {M} x {N} targets, scaled x{M / ms:.2f}: "
              f"{t_pairs:.3f} s per iteration")
    t_gp = None
    if model is not None:
        host = model.to_host() if hasattr(model, "to_host") else model
        rank = host.variance.shape[0]
        Q = np.ascontiguousarray(host.basis) * np.sqrt(host.variance)[None, :]
        ref_flat = np.ascontiguousarray(host.reference).reshape(-1)
        Binv_eps = np.linalg.inv(Q.T @ Q / 1e-5 + np.eye(rank)) / 1e-5           # once per model (not timed)
        P1 = np.resize(P1s, M)
        PX = np.resize(PXs.T, (M, 3))
        cpu_gp_part(Q, ref_flat, P1, PX, y, sigma2, 1.0, Binv_eps, rank)        # warm-up
        t0 = time.perf_counter()
        cpu_gp_part(Q, ref_flat, P1, PX, y, sigma2, 1.0, Binv_eps, rank)
        t_gp = time.perf_counter() - t0
        sample += f"; GP part (numpy / host BLAS, full size, rank {rank}): {t_gp:.3f} s"
    # the strict checker build, for reference (bounded to ~4 s)
    mc = int(min(ms, max(16, 2.0 / max(per_row * 8.0, 1e-9))))
    t0 = time.perf_counter()
    dc = co.cpd_colsum_partial(y, x, sigma2, 0, mc) + co.outlier_constant(M, N, sigma2, w) + 1e-300
    co.cpd_rowstats_partial(y, x, sigma2, dc, 0, mc)
    t_strict = (time.perf_counter() - t0) * (M / mc)
    total = t_pairs + (t_gp or 0.0)
    return {"value": 1.0 / total, "unit": "iterations/s", "cores": cores, "kind": "port", "sample": sample,
            "build": "gcc " + " ".join(cb.CFLAGS[:3]) + " (oracle/cpd_baseline.c), SIMD exponential checked <= 1e-12 against libm",
            "seconds_per_iteration": {"all_pairs": t_pairs, "gp_part": t_gp},
            "thread_calibration_gpairs_per_s": {str(k): round(v, 3) for k, v in cb.LAST_CALIBRATION.items()},
            "cpu_quota_cores": cb.cpu_quota(),
            "strict_checker_build": {"value": 1.0 / t_strict, "unit": "iterations/s (all-pairs passes only)",
                                     "build": "gcc -O2 -fopenmp -ffp-contract=off, scalar libm exp (oracle/cpd_oracle.c)",
                                     "sample_rows": mc}}


def cpu_baseline_stock_structure(with_gpu: bool):
    """`--config1-stock-structure` (mode B): config 1 (femur, 1 622 <-> 1 622) as an EMULATION of the stock
    plugin's cost structure with the numpy oracle -- not a JVM measurement (no JVM exists on either box).  One stock
    iteration materialises the M x N matrix P four times (every case-class copy re-runs the constructor, CPD.scala:54-77
    via GingrAlgorithm.scala:244,246 and the generator wrappers) and recomputes all row sums of P inside every one of
    the M getUncertainty calls (CPD.scala:120-128: O(M^2 N)).  Prints its own JSON object (not the bench line)."""
    from oracle import gingr_oracle as go
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "inputs.npz"))
    ref, target = d["femur"], d["femur_target"]
    M = ref.shape[0]
    model = go.build_gaussian_gpmm(ref, 70.0, 50.0, rel_tol=0.0, max_rank=100)   # femur demo kernel, truncate(100)
    st = go.initial_state(model, go.cpd_initial_sigma2(ref, target))
    w, lam = 0.0, 1.0
    t0 = time.perf_counter()
    for _ in range(4):                                       # four constructor runs per update
        P = go.cpd_P(st.fit, target, st.sigma2, w)
    t_p = time.perf_counter() - t0
    t0 = time.perf_counter()
    var = np.empty(M)
    for i in range(M):                                       # getUncertainty(id): sum(P, Axis._1) recomputed per id
        P1 = P.sum(axis=1)
        var[i] = st.sigma2 * lam / P1[i]
    t_u = time.perf_counter() - t0
    t0 = time.perf_counter()
    go.cpd_update(model, target, st, w=w, lam=lam)           # correspondences, posterior, projections, Umeyama, sigma2
    t_r = time.perf_counter() - t0
    out = {"config": "femur DemoCPD, 1622 <-> 1622 vertices, rank-100 Gaussian GPMM (sigma 70, scaling 50), w = 0",
           "kind": "emulation of the stock plugin's cost structure with the numpy oracle (NOT a JVM measurement)",
           "host_cores": os.cpu_count(),
           "stock_structure_s_per_iteration": t_p + t_u + t_r,
           "breakdown_s": {"four_P_materialisations": t_p, "M_getUncertainty_calls_each_recomputing_row_sums": t_u,
                           "one_algorithm_faithful_iteration": t_r}}
    if with_gpu:
        import torch  # noqa: F401  (first: one HIP runtime per process)
        import gingr_amd as ga
        ctx = ga.Context(0)
        algo = ga.CpdRegistration(ctx)
        s = algo.createInitialState(ga.PointDistributionModel(model.ref, model.mean, model.U, model.lam), target,
                                    ga.CpdConfiguration(maxIterations=200, w=w))
        s = algo.update(s)                                   # warm-up (upload, first launches)
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            s = algo.update(s)                               # host-boundary call: push state, one iteration, pull the fit
        dt = (time.perf_counter() - t0) / n
        out["hip_update_s_per_iteration_host_boundary"] = dt
        out["hip_vs_stock_structure"] = (t_p + t_u + t_r) / dt
        algo.close()
    print(json.dumps(out))


# ------------------------------------------------------------------------------------------------ main
def spawn_ranks(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher around it: start the contract's launch line as a CHILD process -- one rank
    per GPU under torch.distributed.run -- and pass its stdout through.  Nothing in this process has touched the GPU (torch is
    not even imported yet), and nothing is exec'ed: the child is an ordinary subprocess whose exit code we return."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def _child_env():
    """The environment of a child run started by rank 0: nothing of THIS launch's rendezvous may leak into it (with
    TORCHELASTIC_USE_AGENT_STORE left set, a child's own init_process_group waits for the parent's store until its time-out)."""
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE",
            "GROUP_WORLD_SIZE", "ROLE_NAME")
    return {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}


_CHILD_DEADLINE = [None]


def _run_child(cmd, env, cap_s: float):
    """One child run behind the headline, bounded whatever it does: its own session (so that a time-out takes the launcher AND the
    ranks it started down -- killing only the launcher would leave the ranks holding our pipes and this process waiting for their
    end-of-file), output into temporary files instead of pipes, and one budget for all children of this run
    (GINGR_BENCH_CHILD_BUDGET_S, default 420 s) so that a hung child can only delay the headline line by that much.  Returns
    (return code, stdout, stderr); raises TimeoutError."""
    import signal
    import subprocess
    import tempfile
    import time
    if _CHILD_DEADLINE[0] is None:
        _CHILD_DEADLINE[0] = time.monotonic() + float(os.environ.get("GINGR_BENCH_CHILD_BUDGET_S", "420"))
    left = min(cap_s, _CHILD_DEADLINE[0] - time.monotonic())
    if left < 20.0:
        raise TimeoutError("the child runs' time budget is spent")
    with tempfile.TemporaryFile("w+") as so, tempfile.TemporaryFile("w+") as se:
        p = subprocess.Popen(cmd, env=env, stdout=so, stderr=se, stdin=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = p.wait(timeout=left)
        except subprocess.TimeoutExpired:
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(p.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    p.wait(timeout=10)
                    break
                except subprocess.TimeoutExpired:
                    continue
            raise TimeoutError(f"child run exceeded {left:.0f} s and was stopped (process group {p.pid})")
        so.seek(0), se.seek(0)
        return rc, so.read(), se.read()


def run_group_mode(args, world: int, shared_device: bool):
    """The same workload once more through the in-library device group (ONE process, one worker thread per GPU, peer-pointer
    all-reduce) so that one multi-GPU run carries RCCL and the group side by side: a CHILD process (this one has initialised the
    GPU and must not exec), started when the ranks are done with their devices.  Returns the digest of the child's line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--group", "--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--points", str(args.points), "--rank", str(args.rank), "--w", str(args.w), "--roofline-steps", str(args.roofline_steps),
           "--sustained-steps", str(args.sustained_steps), "--no-cpu-baseline", "--no-parity-check"]
    if shared_device:
        cmd += ["--logical-shards", str(world)]   # the one-GPU test mode: the shards share device 0
    env = _child_env()
    try:
        _, stdout, _ = _run_child(cmd, env, 240.0)
        line = [l for l in stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"ms_per_step": d["ms_per_step"], "value": d["value"], "valid": d["valid"], "n_gpus": d["n_gpus"],
                "exchange": d.get("exchange"), "devices": (d.get("rccl_ranks") or {}).get("devices"),
                "distinct_device_uuids": (d.get("rccl_ranks") or {}).get("distinct_device_uuids"),
                "how": "child process `bench.py --group` on the same devices after the headline measurement"}
    except Exception as e:  # the headline line must come out whatever happens here
        return {"error": f"{type(e).__name__}: {e}"[:400]}


def run_variants(args, world: int, shared_device: bool):
    """Two more runs of the same workload as CHILD processes once the ranks are done with their devices (this process has initialised
    the GPU and must not exec): (a) `--emulate-world N` on one GPU -- the per-rank compute without any exchange; (b) the N-rank run
    again with `--ctx-option split_exchange=1`.  Returns digests; never raises (the headline line must come out)."""
    import subprocess
    env = _child_env()
    common = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--points", str(args.points), "--rank", str(args.rank), "--w", str(args.w),
              "--roofline-steps", str(args.roofline_steps), "--sustained-steps", "0", "--no-cpu-baseline", "--no-parity-check", "--no-group-mode",
              "--no-variants"]
    out = {}

    def digest(cmd, what):
        stderr = None
        try:
            _, stdout, stderr = _run_child(cmd, env, 180.0)
            d = json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])
            return {"ms_per_step": d["ms_per_step"], "value": d["value"], "valid": d["valid"], "reason": d.get("reason"), "n_gpus": d["n_gpus"],
                    "exchange": d.get("exchange"), "how": what}
        except Exception as e:
            return {"error": f"{type(e).__name__}: {e}"[:400], "stderr_tail": (stderr[-600:] if stderr is not None else None), "how": what}
    out["emulated_shard_no_exchange"] = digest([sys.executable, os.path.abspath(__file__), "--emulate-world", str(world)] + common,
                                               f"child `bench.py --emulate-world {world}` on GPU 0: rank 0's shard, the other ranks' partial sums missing "
                                               "(a per-rank cost figure, not a registration)")
    out["split_exchange"] = digest([sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--ctx-option", "split_exchange=1"] + common,
                                   f"child `bench.py --gpus {world} --ctx-option split_exchange=1`: the column-sum all-reduce in two halves, the first "
                                   "on the context's second stream beside the second half of the pass")
    return out


def _latest_profile(suffix: str):
    """profiles/rNN_<suffix> of the highest round present (tracked artefacts of tools/final_profile.sh), or None"""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)):
        m = re.match(r"r(\d\d)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def load_pmc_traffic(kernel: str, points: int, rank: int):
    """HBM bytes per launch of `kernel` from the tracked PMC artefact tools/pmc_traffic.sh produced for THIS workload
    (profiles/rNN_pmc_traffic.json, newest round: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes, as the
    microarchitecture guide prescribes).  None when no artefact matches -- bench.py never invents the number."""
    for suffix in ("pmc_traffic.json", "pmc_wide_r256.json"):   # (the metric workload; the same workload at model rank 256: tools/pmc_wide.sh)
        path = _latest_profile(suffix)
        try:
            d = json.load(open(path))
        except Exception:
            continue
        wl = d.get("workload", {})
        if wl.get("points") != points or wl.get("rank") != rank or wl.get("gpus", 1) != 1:
            continue
        k = d.get("kernels", {}).get(kernel)
        if not k or "hbm_bytes_per_launch" not in k:
            continue
        rel = os.path.relpath(path, ROOT)
        return float(k["hbm_bytes_per_launch"]), f"{rel} ({k.get('note', 'FETCH_SIZE x 2 + WRITE_SIZE per launch')})"
    return None, None


def load_pmc_mfma(kernel: str, points: int, rank: int):
    """Matrix-pipe counters of `kernel` (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, ... from tools/pmc_sq.sh's third pass) out of the
    tracked artefact profiles/rNN_pmc_mfma.json for THIS workload; None when there is none."""
    for suffix in ("pmc_mfma.json", "pmc_wide_r256.json"):
        path = _latest_profile(suffix)
        try:
            d = json.load(open(path))
        except Exception:
            continue
        wl = d.get("workload", {})
        if wl.get("points") != points or wl.get("rank") != rank:
            continue
        k = d.get("kernels", {}).get(kernel)
        if not k or "counters" not in k:
            continue
        return dict(k, source=os.path.relpath(path, ROOT))
    return None


def load_rocprof_avg_ms(kernel: str, points: int, rank: int):
    """Average duration of `kernel` in the tracked rocprofv3 --kernel-trace --stats summary of THIS workload (profiles/rNN_bench50k_
    kernel_stats_final.csv for the metric workload, rNN_bench50k_r256_kernel_stats.csv for model rank 256), next to the live HIP-event
    figure: an event pair around a launch adds 2-3 us, which is 5 % of a 50 us kernel and nothing of a 1.2 ms one.  None when no summary
    of this workload is tracked."""
    import csv
    if points != 50000 or rank not in (100, 256):
        return None, None
    path = _latest_profile("bench50k_kernel_stats_final.csv" if rank == 100 else "bench50k_r256_kernel_stats.csv")
    try:
        best = None
        for row in csv.DictReader(open(path)):
            if kernel in row["Name"] and (best is None or int(row["Calls"]) > best[1]):
                best = (float(row["AverageNs"]) * 1e-6, int(row["Calls"]))
        return (best[0], os.path.relpath(path, ROOT)) if best else (None, None)
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--rank", type=int, default=100)
    ap.add_argument("--w", type=float, default=0.1)
    ap.add_argument("--host-gpmm", action="store_true", help="synthesise the GPMM with numpy on the host and upload it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=3)
    ap.add_argument("--sustained-steps", type=int, default=200,
                    help="a second, longer timed run of the same workload reported as `sustained` next to the contract's K steps (0: skip)")
    ap.add_argument("--sigma2", type=float, default=0.0,
                    help="experiment: start from this sigma2 instead of the CPD initial value (late-iteration regime)")
    ap.add_argument("--group", action="store_true",
                    help="ONE process drives all GPUs through the in-library device group (gingr_group_*) instead of one process per "
                         "GPU with torch.distributed")
    ap.add_argument("--logical-shards", type=int, default=0,
                    help="with --group: this many LOGICAL shards on device 0 (protocol test on a one-GPU box; the shards share the GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: run the N>1 code path (process group + phase/all-reduce driver) with the given world")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="timing experiment on ONE GPU: own only rows of rank 0 of a world of this size (results are not a "
                         "valid registration: the other shards' partial sums are missing); shows the per-rank cost at N GPUs")
    ap.add_argument("--config1-stock-structure", choices=["cpu", "gpu"], default=None,
                    help="instead of the benchmark: config 1 (femur) stock-structure emulation (SURVEY 8d mode B); 'gpu' "
                         "also times the HIP path on the same inputs")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="one measurement line for BASELINE.json config N (tools/bench_configs.py) instead of the headline benchmark")
    ap.add_argument("--exchange", choices=["rccl-native", "torch"], default="rccl-native",
                    help="N > 1 (one process per GPU): who runs the two all-reduces of an iteration -- the library itself (ncclAllReduce on "
                         "its own stream, gingr_fitter_update_cpd_rccl_async; default) or torch.distributed called back once per exchange")
    ap.add_argument("--no-group-mode", action="store_true",
                    help="N > 1: skip the second measurement through the in-library device group (a child process started by rank 0 after "
                         "the headline measurement; its line is folded into the output as `group_mode`)")
    ap.add_argument("--no-variants", action="store_true",
                    help="N > 1: skip the two child runs behind the headline (emulated shard without exchange, split column-sum exchange)")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="NAME=VALUE",
                    help="same-box comparisons: gingr_ctx_set_option on the context, NAME in cull | fine_cull | nn_grid | tri_grid | "
                         "split_exchange | gram_downdate (all select between code paths with the same results)")
    args = ap.parse_args()
    if args.config:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs
        # config 5 = independent chains: `--gpus N` starts N chain processes, process i with HIP_VISIBLE_DEVICES=i (this process
        # stays a launcher and touches no GPU)
        line = bench_configs.CONFIGS[5](gpus=args.gpus) if args.config == 5 else bench_configs.CONFIGS[args.config]()
        print(json.dumps(line), flush=True)
        return 0
    if args.config1_stock_structure:
        cpu_baseline_stock_structure(args.config1_stock_structure == "gpu")
        return 0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.group:
        return spawn_ranks(args.gpus, sys.argv[1:])          # before torch / HIP are touched
    if args.group:
        world, rank, local_rank = 1, 0, 0                     # one process whatever the launcher said
    elif world != args.gpus:
        args.gpus = world

    # must be in the environment before the HIP / HSA runtime initialises (the pool's driver only supports dmabuf IPC; RCCL and
    # cross-process tensor sharing fail with the legacy mode); normally already exported
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if rank != 0:
        # only rank 0 reports; RCCL prints a version banner through C stdio on every rank, which would otherwise be flushed into
        # the shared stdout at an arbitrary time (possibly after rank 0's JSON line)
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)

    import torch
    import torch.distributed as dist
    import gingr_amd as ga
    from gingr_amd.sharded import ShardedFitter

    # testing on a ONE-GPU box: every rank uses device 0 and the exchange goes through gloo on host copies, so the contract's
    # launch line (`--gpus 2` -> torch.distributed.run -> two ranks) runs end to end without a second GPU; never a measurement
    shared_device = os.environ.get("GINGR_BENCH_SHARED_DEVICE") == "1"
    if shared_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = (world > 1 or args.force_dist or args.emulate_world > 1) and not args.group
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:       # stand-alone test modes only; torchrun always provides it
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        import datetime
        patience = datetime.timedelta(seconds=300)  # a rank that never shows up fails the run in minutes, not after the default half hour
        if shared_device:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=patience)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, timeout=patience,
                                    device_id=torch.device(f"cuda:{local_rank}"))

    M = N = args.points
    y, x = synth_clouds(M)
    n_shards = world

    # ---------------------------------------------------------------------------------------------- set-up (untimed)
    stream = None
    if args.group:
        # one process, the in-library group: one shard per GPU, or logical shards on device 0
        n_shards = args.logical_shards if args.logical_shards > 0 else args.gpus
        devices = [0] * n_shards if args.logical_shards > 0 else list(range(args.gpus))
        group = ga.DeviceGroup(devices)
        if args.host_gpmm:
            basis, lam = synth_gpmm(y, args.rank)
            group.upload_model(y, np.zeros_like(y), basis, lam)
        else:
            group.build_gaussian_gpmm(y, [70.0], [50.0], 0.0, args.rank)
        group.set_target(x)
        group.set_options(ga.GlobalTranformationType.RigidTransforms, 1.0)
        ctx = ga.Context(0)                                   # stateless helpers (initial sigma2) + nothing else
        tctx_handle = group.ctx_handle(0)                     # timing hooks of shard 0
        model = None
        m_loc = group.shard_rows(0)[1] - group.shard_rows(0)[0]

        class Runner:
            def reset(self, s2):
                group.set_state(np.zeros(args.rank), s2)

            def update(self, n):
                group.update_cpd(args.w, 1.0, n)

            def sync(self):
                group.synchronize()

            def state(self):
                return group.get_state()

            def stats(self):
                return None                                   # per-shard statistics are not exposed through the group

            def close(self):
                group.close()
        runner = Runner()
        lib = group._lib

        def timing(which=None, enable=None, reset=False):
            import ctypes
            if enable is not None:
                lib.gingr_ctx_timing_enable(tctx_handle, 1 if enable else 0)
            if reset:
                lib.gingr_ctx_timing_reset(tctx_handle)
            if which is not None:
                ms, n = ctypes.c_double(), ctypes.c_int64()
                lib.gingr_ctx_timing_read(tctx_handle, which, ctypes.byref(ms), ctypes.byref(n))
                return ms.value, n.value
    else:
        ctx = ga.Context(local_rank)
        for kv in args.ctx_option:
            name, _, val = kv.partition("=")
            from gingr_amd import _native as _nat
            ctx.set_option(getattr(_nat, "OPT_" + name.upper()), int(val))   # cull | fine_cull | nn_grid | tri_grid
        stream = torch.cuda.Stream(device=local_rank)
        ctx.set_stream(stream.cuda_stream)
        if args.host_gpmm:
            basis, lam = synth_gpmm(y, args.rank)
            model = ga.PointDistributionModel(reference=y, mean=np.zeros_like(y), basis=basis, variance=lam)
        else:
            # GPMMTriangleMesh3D(reference, tol).Gaussian(sigma, scaling) built in HBM (femur demo kernel,
            # examples/DemoHelper/DemoDatasetLoader.scala:113-114), stopped at exactly `rank` columns; untimed set-up
            model = ga.GPMMTriangleMesh3D(ctx, y, relativeTolerance=0.0, maxRank=args.rank).Gaussian(70.0, 50.0)

        def all_reduce(t):
            if shared_device:
                h = t.cpu()                                   # waits for the producing kernels on the current stream
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)

        # The exchange of the N > 1 run.  Default: the library's own RCCL communicator (one per rank, bootstrapped through the process
        # group that torch.distributed.run set up: rank 0 creates the ncclUniqueId, everybody receives its 128 bytes) -- the two
        # all-reduces of an iteration are then ncclAllReduce calls enqueued by libgingr_hip on the kernels' stream.  The shared-device
        # test mode cannot have one (RCCL refuses two ranks on one GPU): there, and with --exchange torch, torch.distributed is
        # called back once per exchange.
        native = use_dist and args.exchange == "rccl-native" and not shared_device
        exchange_path = None
        if use_dist:
            exchange_path = ("rccl-native" if native else
                             ("gloo on host copies (shared-device test mode)" if shared_device else "torch.distributed (nccl) callback"))
        native_error = None
        if native:
            comm_world = world if not args.emulate_world else 1
            # ncclCommInitRank is collective: a rank that cannot even bind librccl would leave the others blocked inside it until a
            # time-out, so every rank loads the library first and all of them learn whether everybody could
            try:
                ctx.rccl_load()
                loaded = True
            except Exception as ex:
                loaded, native_error = False, repr(ex)
            if comm_world > 1:
                lf = torch.tensor([1 if loaded else 0], dtype=torch.int32, device=f"cuda:{local_rank}")
                dist.all_reduce(lf, op=dist.ReduceOp.MIN)
                loaded = bool(lf.item())
            uid = [None]
            if loaded:
                try:
                    uid = [ctx.rccl_unique_id() if rank == 0 else None]
                except Exception as ex:      # (everybody must learn it)
                    uid, native_error = [None], repr(ex)
                if comm_world > 1:
                    dist.broadcast_object_list(uid, src=0)
            ok = uid[0] is not None
            if ok:
                try:
                    ctx.rccl_init(uid[0], comm_world, rank if comm_world > 1 else 0)
                    # one all-reduce through the new communicator before anything depends on it: every rank contributes 1
                    probe = torch.ones(4, dtype=torch.float64, device=f"cuda:{local_rank}")
                    torch.cuda.synchronize(local_rank)
                    ctx.rccl_allreduce(probe.data_ptr(), 4)
                    ctx.synchronize()
                    ok = bool(torch.all(probe == float(comm_world)).item())
                    if not ok:
                        native_error = f"probe all-reduce returned {probe.tolist()} for {comm_world} ranks"
                except Exception as ex:
                    ok, native_error = False, repr(ex)
            if comm_world > 1:               # all ranks take the same path: native only if it works everywhere
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=f"cuda:{local_rank}")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(flag.item())
            if not ok:
                native = False
                exchange_path = f"torch.distributed (nccl) callback -- native RCCL exchange unavailable: {native_error or 'failed on another rank'}"
                print(f"[bench] rank {rank}: {exchange_path}", file=sys.stderr)
        with torch.cuda.stream(stream):
            shard_world = args.emulate_world if args.emulate_world > 1 else world
            fitter = ShardedFitter(ctx, model, x, rank=rank, world=shard_world, all_reduce=all_reduce if use_dist else None,
                                   global_transform=ga.GlobalTranformationType.RigidTransforms, step_length=1.0, rccl=native)
            if args.force_dist and shard_world == 1 and not native:      # exercise the phase + all-reduce driver with one rank
                from gingr_amd.sharded import as_torch, NUM_SEGMENTS
                import ctypes
                from ctypes import c_int64, c_void_p
                pp = c_void_p(); offs = (c_int64 * NUM_SEGMENTS)(); cnts = (c_int64 * NUM_SEGMENTS)()
                fitter._lib.gingr_fitter_exchange(fitter.handle, ctypes.byref(pp), offs, cnts)
                fitter.xch = as_torch(pp.value, offs[NUM_SEGMENTS - 1] + cnts[NUM_SEGMENTS - 1], local_rank)
                fitter.world = 2
        m_loc = fitter.end - fitter.begin

        class Runner:
            def reset(self, s2):
                with torch.cuda.stream(stream):
                    fitter.set_state(np.zeros(args.rank), s2)

            def update(self, n):
                with torch.cuda.stream(stream):
                    fitter.update_cpd(args.w, 1.0, n)

            def sync(self):
                torch.cuda.synchronize(local_rank)
                if use_dist:
                    dist.barrier()
                    torch.cuda.synchronize(local_rank)

            def state(self):
                return fitter.get_state()

            def stats(self):
                return fitter.get_cpd_stats()

            def close(self):
                fitter.close()
        runner = Runner()

        def timing(which=None, enable=None, reset=False):
            if enable is not None:
                ctx.timing_enable(enable)
            if reset:
                ctx.timing_reset()
            if which is not None:
                return ctx.timing_read(which)

    sigma2_0 = ctx.cpd_initial_sigma2(y, x)      # CpdRegistrationState.apply, CPD.scala:92-102 (mean == reference here)
    if args.sigma2 > 0:
        sigma2_0 = args.sigma2

    # ---------------------------------------------------------------------------------------------- the contract's timing
    # one-time costs (code-object loads, LDS-size attributes, clock ramp) are paid on a throw-away run of three updates; the
    # state is then reset, so the W warm-up and K timed steps below run the workload from sigma2_0 exactly as specified
    runner.reset(sigma2_0)
    runner.update(3)
    runner.sync()

    def max_over_ranks(v):
        if not use_dist:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device="cpu" if shared_device else f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def timed_updates(steps, warmup):
        """W warm-up + K timed updates from sigma2_0.  Every timed update must be one of a LIVE registration: an update of a state
        that has failed (ModelFlexibilityError: the state is frozen, GingrAlgorithm.scala:194-210) runs in a fully culled regime
        and times nothing of the workload.  A probe run of the same W + K updates says whether the registration survives them
        (the trajectory is deterministic); if it collapses after L < W + K updates the K timed updates are taken in blocks of
        `reset, w warm-up, n timed` with w + n < L, each block bracketed by a synchronisation, and the block times are added up.
        Returns (seconds of the K updates [max over ranks], host enqueue seconds, info dict, final state, why-invalid or None)."""
        runner.reset(sigma2_0)
        runner.update(warmup + steps)
        runner.sync()
        _, scp, _ = runner.state()
        info = {"steps": steps, "warmup": warmup, "blocks": 1, "live_iterations_from_sigma2_0": None}
        if scp.status == 0 and scp.iteration == warmup + steps:
            plan = [(warmup, steps)]
        else:
            live = int(scp.iteration)             # updates that succeeded before the failing one
            usable = live - 1                     # one update of margin to the collapse
            w = min(warmup, max(1, usable // 5))
            n = usable - w
            info.update({"live_iterations_from_sigma2_0": live, "status_at_collapse": int(scp.status)})
            if n < 1:
                return None, None, info, runner.state(), (f"the registration fails after {live} updates from sigma2_0 "
                                                          f"(status {int(scp.status)}): nothing live to time")
            plan = []
            left = steps
            while left > 0:
                plan.append((w, min(n, left)))
                left -= plan[-1][1]
            info.update({"blocks": len(plan), "warmup_per_block": w, "timed_per_block": n,
                         "note": f"collapse after {live} updates: timed in {len(plan)} block(s) of reset + {w} warm-up + <= {n} timed updates, "
                                 "every timed update is one of a live registration"})
        total, enq, bad = 0.0, 0.0, None
        st = None
        for (w, n) in plan:
            runner.reset(sigma2_0)
            runner.update(w)
            runner.sync()
            t0 = time.perf_counter()
            runner.update(n)
            enq += time.perf_counter() - t0       # the call returns when the n steps are ENQUEUED: host cost of the launches
            runner.sync()
            total += time.perf_counter() - t0
            st = runner.state()
            if st[1].status != 0 or st[1].iteration != w + n:
                bad = (f"a timed block ended in status {int(st[1].status)} after {int(st[1].iteration)} of {w + n} updates "
                       "(failed / frozen state: not a timing of the workload)")
        return max_over_ranks(total), enq, info, st, bad

    elapsed, enqueue_s, timing_info, (alpha, sc, fit), reason = timed_updates(args.steps, args.warmup)
    if reason is None and not np.all(np.isfinite(fit)):
        reason = "non-finite fit after the timed steps"
    if reason is None and args.emulate_world > 1:
        reason = ("emulated shard (--emulate-world): the other ranks' partial sums are missing, a per-rank cost experiment and not "
                  "a registration")
    ok = reason is None
    sigma2_timed = float(sc.sigma2)
    # the driver's contract fixes K = 20 (46 ms at 50k): a longer run of the same workload next to it, same rules
    sustained = None
    if elapsed is not None and args.sustained_steps > 0:
        s_el, _, s_info, s_state, s_bad = timed_updates(args.sustained_steps, args.warmup)
        sustained = {"steps": args.sustained_steps, "valid": s_bad is None and s_el is not None and ok, "timing": s_info, "reason": s_bad}
        if s_el is not None:
            sustained.update({"ms_per_step": s_el / args.sustained_steps * 1e3, "value": args.sustained_steps / s_el,
                              "unit": "iterations/s", "sigma2_after": float(s_state[1].sigma2)})
    if elapsed is None:                           # nothing could be timed: keep the line well-formed
        elapsed, enqueue_s = float("nan"), float("nan")

    # ---- live roofline of the dominant kernels (HIP events on the kernels' stream, extra iterations)
    roof = None
    kernels = []
    # (from a defined, live spot of the same trajectory: sigma2_0 + the warm-up of the timed blocks)
    w_roof = int(timing_info.get("warmup_per_block", args.warmup))
    runner.reset(sigma2_0)
    runner.update(w_roof)
    runner.sync()
    timing(enable=True, reset=True)
    runner.update(args.roofline_steps)
    runner.sync()
    pairs = float(m_loc) * float(N)
    names = {0: ("cpd_colsum_kernel", 11.0), 1: ("cpd_rowstats_kernel", 18.0)}
    for which, (name, flops_per_pair) in names.items():
        ms, n = timing(which)
        if n:
            avg = ms / n
            ach = flops_per_pair * pairs / (avg * 1e-3) / 1e12
            kernels.append({"kernel": name, "avg_ms": avg, "launches": n, "bound": "valu_f64",
                            "achieved": ach, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / F64_VALU_PEAK_TFLOPS, "algorithmic_flops_per_pair": flops_per_pair})
    # Secondary kernels, with SURVEY 8d's formulas at the MODEL rank r (not the padded rank rp the kernels stream and multiply):
    #   Gram: 3 M r^2 flops SYRK-halved -- what a triangle-only kernel has to issue, and `frac`; 6 M r^2 (full symmetric) as a note;
    #         its pass over the basis is 24 M r bytes, quoted against the HBM peak next to it (neither bound is near saturation
    #         at r = 100; the matrix pipe is the binding one from r = 256 on);
    #   sweeps: 24 M r bytes.
    r_model = args.rank
    rp = (args.rank + 15) // 16 * 16
    wide = rp >= 128
    ms, n = timing(2)
    if n:
        avg = ms / n
        gram_name = "gram_wide_kernel" if wide else "gram_tri_kernel"
        ach_half = 3.0 * m_loc * r_model * r_model / (avg * 1e-3) / 1e12
        hbm = 24.0 * m_loc * r_model / (avg * 1e-3) / 1e9
        kernels.append({"kernel": gram_name, "avg_ms": avg, "launches": n, "bound": "mfma",
                        "achieved": ach_half, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach_half / F64_MFMA_PEAK_TFLOPS,
                        "algorithmic_flops": 3.0 * m_loc * r_model * r_model,
                        "note": "SYRK-halved count 3 M r^2 at the model rank (the kernel multiplies the upper triangle only); "
                                "full_symmetric_* is SURVEY 8d's 6 M r^2 count of the same launch",
                        "full_symmetric_achieved": 2.0 * ach_half, "full_symmetric_frac": 2.0 * ach_half / F64_MFMA_PEAK_TFLOPS,
                        "hbm_achieved_GBs": hbm, "hbm_frac": hbm / HBM_PEAK_GBS, "algorithmic_bytes": 24.0 * m_loc * r_model,
                        "padded_rank": rp,
                        "traffic": (load_pmc_traffic(gram_name, M, args.rank) if n_shards == 1 and not args.emulate_world else (None, None))[0],
                        # busy cycles of the matrix pipe / (SIMDs x kernel cycles) from the SQ counters (tracked artefact), next to
                        # the time-derived fractions above
                        "mfma_counters": load_pmc_mfma(gram_name, M, args.rank) if n_shards == 1 and not args.emulate_world else None})
    ms, n = timing(4)
    if n:
        avg = ms / n
        gbs = 24.0 * m_loc * r_model / (avg * 1e-3) / 1e9
        tr, src = load_pmc_traffic("sweep_fit_boxes_kernel", M, args.rank) if n_shards == 1 else (None, None)
        kernels.append({"kernel": "sweep_fit_boxes_kernel", "avg_ms": avg, "launches": n, "bound": "hbm", "achieved": gbs,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes": 24.0 * m_loc * r_model, "padded_rank": rp, "traffic": tr, "traffic_source": src})
    ms, n = timing(5)
    if n:
        if rp > 240:    # the multi-workgroup blocked solve (gp.hip: launch_posterior_solve): one event pair around all of its launches
            kernels.append({"kernel": "posterior solve (solve_system + per 64-column panel chol_block64 / chol_panel / chol_trailing / "
                                      "chol_backward + solve_finish)", "avg_ms": ms / n, "launches": n, "bound": "latency",
                            "note": "r^3 / 3 flops on the critical path of the iteration; ~6 us per launch, the 64 x 64 diagonal block "
                                    "(one workgroup, 17 us) per panel"})
        else:
            solve_name = ("posterior_solve_wide_kernel" if rp > 128 else "posterior_solve_lds_kernel")
            kernels.append({"kernel": solve_name, "avg_ms": ms / n, "launches": n, "bound": "latency",
                            "note": "one workgroup: r^3 / 3 flops on the critical path of the iteration"})
    if n_shards == 1 and not args.emulate_world:
        for k in kernels:   # the tracked rocprof summary of the same workload, where there is one
            ms_r, src_r = load_rocprof_avg_ms(k["kernel"], M, args.rank)
            if ms_r is not None:
                k["rocprof_avg_ms"] = ms_r
                k["rocprof_source"] = src_r
                if "achieved" in k and k.get("avg_ms"):
                    k["frac_from_rocprof"] = k["frac"] * k["avg_ms"] / ms_r
    # (the whole update as HIP events see it DURING these instrumented iterations: the per-kernel event pairs serialise the
    # launches, and the iterations sit at a later sigma2 than the timed ones -- so it is an upper bracket of ms_per_step, not a
    # second measurement of it)
    ms, n = timing(3)
    upd_ms = ms / n if n else None
    # ---- the two exchanges of an iteration as this rank / shard 0 sees them (HIP events on the kernels' stream around the collective
    # resp. around the group's wait + peer-sum kernel: includes waiting for the slowest peer); N > 1: maximum over the ranks
    exchange = None
    if n_shards > 1 or use_dist:
        ex = []
        for seg in (0, 1):
            ms, n = timing(6 + seg)
            ex.append(ms / n if n else 0.0)
        per_rank = None
        if use_dist:
            # every rank's own view first (events around each collective on ITS stream: its wait for the slowest peer included), with the
            # two pair loops it ran next to them -- the first 8-GPU line has to be readable rank by rank
            mine = {"rank": rank, "segment0_column_sums_ms": ex[0], "segment1_gram_bundle_ms": ex[1]}
            for which, name in ((0, "cpd_colsum_kernel_ms"), (1, "cpd_rowstats_kernel_ms"), (3, "instrumented_update_ms")):
                ms_k, n_k = timing(which)
                mine[name] = ms_k / n_k if n_k else None
            per_rank = [None] * dist.get_world_size()
            dist.all_gather_object(per_rank, mine)
            te = torch.tensor(ex, dtype=torch.float64, device="cpu" if shared_device else f"cuda:{local_rank}")
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            ex = [float(v) for v in te.tolist()]
        exchange = {"path": exchange_path if use_dist else "in-library device group (peer pointers)",
                    "per_rank": per_rank,
                    "rccl": ctx.rccl_info() if (use_dist and native) else None,
                    "segment0_column_sums_ms": ex[0], "segment1_gram_bundle_ms": ex[1],
                    "bytes": {"segment0": 8 * N, "segment1": 8 * (((args.rank + 15) // 16 * 16) ** 2 + 2 * ((args.rank + 15) // 16 * 16) + 8)},
                    "how": "HIP events around each all-reduce on the stream the kernels run on, averaged over the roofline "
                           "iterations, max over ranks; includes the wait for the slowest peer"}
    timing(enable=False)
    # ---- who took part: the world the collective library saw and the physical devices behind the ranks / shards
    ranks_info = None
    if use_dist:
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank, "device": props.name, "uuid": str(getattr(props, "uuid", "")),
                "pci_bus_id": getattr(props, "pci_bus_id", None)}
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, mine)
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "distinct_device_uuids": len({g["uuid"] for g in gathered}), "ranks": gathered}
    elif args.group:
        info = group.exchange_info()
        uu = [str(getattr(torch.cuda.get_device_properties(d), "uuid", d)) for d in devices]
        ranks_info = {"world_size": n_shards, "backend": "in-library device group (peer pointers, one-shot all-reduce)",
                      "distinct_device_uuids": len(set(uu)), "devices": devices, "uuids": uu, **info}
    dom = max((k for k in kernels if k["bound"] == "valu_f64"), key=lambda k: k["avg_ms"], default=None)
    if dom is not None:
        tr, src = (load_pmc_traffic(dom["kernel"], M, args.rank)
                   if (n_shards == 1 and not args.emulate_world and M == N) else (None, None))
        roof = {"bound": "valu_f64", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": dom["peak"],
                "unit": "TFLOP/s", "frac": dom["frac"],
                # HBM bytes per launch from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes), read from the tracked
                # artefact of tools/pmc_traffic.sh for this workload; null when there is none
                "traffic": tr, "traffic_source": src,
                "algorithmic_bytes": 24.0 * (m_loc + N) + 32.0 * m_loc,
                "nearest_contract_bound": "mfma",
                "note": "all-pairs kernel: O(M+N) bytes, O(M*N) float64 VALU flops (software exp counted as 1 flop); "
                        "HBM and MFMA are not the binding resource.  In the contract's hbm|mfma vocabulary this is the "
                        "compute side: the peak used, 78.6 TFLOP/s, is also the dense f64 MFMA peak -- on gfx950 the f64 "
                        "vector and matrix pipes share the issue slots (profiles/r01_ubench_mfma_valu_overlap.txt), so "
                        "moving the K=3 contraction to MFMA does not raise the ceiling (measured in rounds 1-2: tools/experiments/experiment_affinity_mfma.hip)"}

    # ---- parity of the state the measurements ended in (outside every timed region): ONE more update on the device, then the
    # affinity statistics of that evaluation (P1, PX of this rank's rows; den, Np) and the sigma2 it committed against the strict
    # C oracle evaluated at the pre-update state (full size: ~2.5e9 pair evaluations on the host cores)
    parity = None
    if not args.no_parity_check and args.emulate_world <= 1 and rank == 0 and not args.group and world == 1:
        from oracle import c_oracle as co
        _, sc, fit = runner.state()                            # the state after the warm-up + roofline iterations
        runner.update(1)
        runner.sync()
        _, sc_next, _ = runner.state()
        got = runner.stats()
        want = co.cpd_stats(fit, x, float(sc.sigma2), args.w)
        e_p1 = float(np.max(np.abs(got["P1"] - want.P1) / np.maximum(np.abs(want.P1), 1e-300)))
        e_px = float(np.linalg.norm(got["PX"] - want.PX) / np.linalg.norm(want.PX))
        e_den = float(np.max(np.abs(got["den"] - want.den) / want.den))
        e_s2 = float(abs(sc_next.sigma2 - want.sigma2_next) / want.sigma2_next)
        parity = {"against": f"oracle/cpd_oracle.c (strict C restatement, parity unpinned), one update from the state {w_roof} warm-up + {args.roofline_steps} roofline updates after sigma2_0",
                  "rows_checked": int(want.P1.shape[0]), "P1_max_rel": e_p1, "PX_rel_l2": e_px, "den_max_rel": e_den,
                  "sigma2_next_rel": e_s2, "tolerance": 1e-8}
        parity["ok"] = bool(max(e_p1, e_px, e_den, e_s2) < 1e-8) and sc_next.status == 0
        if not parity["ok"] and reason is None:
            reason = "parity check against the strict C oracle failed (see parity_check)"
        ok = ok and parity["ok"]

    # ---- N > 1: the sharded run must land on the state a single shard reaches from the same start (outside every timed region;
    # rank 0 replays the warm-up + roofline iterations the sharded state went through on its own GPU: milliseconds).  Same arithmetic, different order of
    # the partial sums, so agreement is ~1e-12; a wrong exchange shows up as O(1)
    shard_check = None
    if n_shards > 1 and rank == 0 and not args.emulate_world and not args.group and not args.no_parity_check:
        a_sh, sc_sh, _ = runner.state()
        with torch.cuda.stream(stream):
            single = ShardedFitter(ctx, model, x, rank=0, world=1, all_reduce=None,
                                   global_transform=ga.GlobalTranformationType.RigidTransforms, step_length=1.0)
            single.set_state(np.zeros(args.rank), sigma2_0)
            single.update_cpd(args.w, 1.0, w_roof + args.roofline_steps)
        torch.cuda.synchronize(local_rank)
        a_1, sc_1, _ = single.get_state()
        single.close()
        e_s2 = float(abs(sc_sh.sigma2 - sc_1.sigma2) / sc_1.sigma2)
        e_a = float(np.max(np.abs(a_sh - a_1)))
        e_t = float(np.max(np.abs(np.array(sc_sh.translation[:]) - np.array(sc_1.translation[:]))))
        shard_check = {"against": "the same iterations on ONE shard (rank 0's GPU)", "iterations": int(sc_1.iteration),
                       "sigma2_rel": e_s2, "alpha_max_abs": e_a, "translation_max_abs": e_t, "tolerance": 1e-8,
                       "ok": bool(max(e_s2, e_a, e_t) < 1e-8 and sc_sh.iteration == sc_1.iteration)}
        if not shard_check["ok"] and reason is None:
            reason = "the sharded run does not land on the single-shard state (see shard_consistency)"
        ok = ok and shard_check["ok"]

    cpu = None
    if rank == 0 and n_shards == 1 and not args.no_cpu_baseline and not args.emulate_world:
        cpu = cpu_baseline(y, x, sigma2_0, args.w, model=model)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mode = ("in-library device group, one process" + (" (logical shards on device 0)" if args.logical_shards > 0 else "")
                if args.group else ((f"one process per GPU, {exchange_path}") if (n_shards > 1 or use_dist) else "single shard"))
        out = {
            "metric": "GiNGR update iters/sec, 50k<->50k CPD",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": args.gpus if args.group else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"CPD update, synthetic Gaussian clouds {M}<->{N}, GPMM rank {args.rank}, w={args.w}, "
                                   f"rigid global transform, sigma2_0={sigma2_0:.3f}",
                       "points": M, "targets": N, "rank": args.rank, "parallelism": f"row-shard x{n_shards}",
                       "exchange": mode, "emulated_world": args.emulate_world or None},
            "valid": ok,
            "reason": reason,
            "timing": timing_info,
            "sustained": sustained,
            "parity_check": parity,
            "shard_consistency": shard_check,
            "sigma2_after_timed_steps": sigma2_timed,
            "instrumented_update_ms": {"value": upd_ms, "note": "mean of the roofline iterations with per-kernel HIP events on (they "
                                       "serialise the launches; later sigma2 than the timed steps): an upper bracket of ms_per_step"},
            # host time to enqueue one step (12 kernel launches, no synchronisation inside): what a HIP graph could save at most
            "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
            "rccl_ranks": ranks_info,
            "exchange": exchange,
            "roofline": roof,
            "kernels": kernels,
            "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
    if use_dist:
        # rank 0 may still be busy with its single-shard replay / parity work: everybody leaves the process group together
        dist.barrier()
    runner.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and world > 1 and not args.group and not args.no_group_mode and not args.emulate_world:
        out["group_mode"] = run_group_mode(args, world, shared_device)
    if rank == 0 and world > 1 and not args.group and not args.no_variants and not args.emulate_world:
        # next to the headline, from child processes on the same devices: what one rank's shard costs with the exchanges taken out
        # (an emulated shard on one GPU), and the same N-rank run with the column-sum exchange split in two halves
        # (GINGR_OPT_SPLIT_EXCHANGE = 1; off by default: 26-35 us slower in the single-GPU emulation, never seen on real links)
        out["variants"] = run_variants(args, world, shared_device)
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so that the JSON is the LAST line on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)

#!/usr/bin/env python3
"""bench.py -- GiNGR update iterations/second, CPD 50k <-> 50k, on N MI355X (one process per GPU).

A "step" is ONE full update iteration (GingrAlgorithm.update + the fit refresh) of the CPD configuration on the
synthetic 50k <-> 50k workload of SURVEY.md section 8d: both all-pairs passes over the 50 000 x 50 000 affinity,
the weighted Gram + posterior solve, two coefficient projections, Umeyama and the new fit.  Model, target and state
are resident in HBM before the timed region.  With --gpus N the reference rows are sharded over N ranks and the
partial sums are all-reduced over RCCL (strong scaling: the problem is fixed).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the kernels' own stream in extra
iterations after the timed region; `cpu_baseline` times the C restatement (oracle/, the checker -- never the product)
on the host cores on a bounded sample of the same workload.
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
def cpu_baseline(y, x, sigma2, w, budget_s=20.0):
    """Time the C restatement of the two all-pairs passes (oracle/cpd_oracle.c, OpenMP over all host cores) on a row
    sample of the SAME workload and scale to one iteration.  The GP part (O(M r^2)) is not included, which favours the
    CPU number.  Returns the cpu_baseline JSON object."""
    from oracle import c_oracle as co
    cores = co.num_threads()
    M, N = y.shape[0], x.shape[0]
    # calibrate on a small slice, then pick the sample so that both passes take about budget_s
    m0 = max(64, min(M, 16 * cores))
    t0 = time.perf_counter()
    co.cpd_colsum_partial(y, x, sigma2, 0, m0)
    dt = time.perf_counter() - t0
    per_row = dt / m0
    ms = int(min(M, max(m0, budget_s / 2.0 / max(per_row, 1e-9))))
    t0 = time.perf_counter()
    den = co.cpd_colsum_partial(y, x, sigma2, 0, ms) + co.outlier_constant(M, N, sigma2, w) + 1e-300
    co.cpd_rowstats_partial(y, x, sigma2, den, 0, ms)
    dt = time.perf_counter() - t0
    it_per_s = 1.0 / (dt * (M / ms))
    return {"value": it_per_s, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"both all-pairs passes on rows [0,{ms}) of {M} x {N} targets, scaled x{M / ms:.2f}; "
                      f"{dt:.1f} s measured; GP solve excluded"}


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
    ap.add_argument("--roofline-steps", type=int, default=3)
    ap.add_argument("--sigma2", type=float, default=0.0,
                    help="experiment: start from this sigma2 instead of the CPD initial value (late-iteration regime)")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: run the N>1 code path (process group + phase/all-reduce driver) with the given world")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="timing experiment on ONE GPU: own only rows of rank 0 of a world of this size (results are not a "
                         "valid registration: the other shards' partial sums are missing); shows the per-rank cost at N GPUs")
    ap.add_argument("--config1-stock-structure", choices=["cpu", "gpu"], default=None,
                    help="instead of the benchmark: config 1 (femur) stock-structure emulation (SURVEY 8d mode B); 'gpu' "
                         "also times the HIP path on the same inputs")
    args = ap.parse_args()
    if args.config1_stock_structure:
        cpu_baseline_stock_structure(args.config1_stock_structure == "gpu")
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
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

    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist or args.emulate_world > 1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:       # stand-alone test modes only; torchrun always provides it
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    M = N = args.points
    y, x = synth_clouds(M)

    ctx = ga.Context(local_rank)
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
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    with torch.cuda.stream(stream):
        shard_world = args.emulate_world if args.emulate_world > 1 else world
        fitter = ShardedFitter(ctx, model, x, rank=rank, world=shard_world, all_reduce=all_reduce if use_dist else None,
                               global_transform=ga.GlobalTranformationType.RigidTransforms, step_length=1.0)
        if args.force_dist and shard_world == 1:      # exercise the phase + all-reduce driver with one rank
            from gingr_amd.sharded import as_torch, NUM_SEGMENTS
            import ctypes
            from ctypes import c_int64, c_void_p
            pp = c_void_p(); offs = (c_int64 * NUM_SEGMENTS)(); cnts = (c_int64 * NUM_SEGMENTS)()
            fitter._lib.gingr_fitter_exchange(fitter.handle, ctypes.byref(pp), offs, cnts)
            fitter.xch = as_torch(pp.value, offs[NUM_SEGMENTS - 1] + cnts[NUM_SEGMENTS - 1], local_rank)
            fitter.world = 2
        sigma2_0 = ctx.cpd_initial_sigma2(y, x)      # CpdRegistrationState.apply, CPD.scala:92-102 (mean == reference here)

        if args.sigma2 > 0:
            sigma2_0 = args.sigma2

        def reset():
            fitter.set_state(np.zeros(args.rank), sigma2_0)

        def sync():
            torch.cuda.synchronize(local_rank)
            if use_dist:
                dist.barrier()
                torch.cuda.synchronize(local_rank)

        # one-time costs (code-object loads, LDS-size attributes, clock ramp) are paid on a throw-away run of three updates; the
        # state is then reset, so the W warm-up and K timed steps below run the workload from sigma2_0 exactly as specified
        reset()
        fitter.update_cpd(args.w, 1.0, 3)
        sync()
        reset()
        fitter.update_cpd(args.w, 1.0, args.warmup)
        sync()
        t0 = time.perf_counter()
        fitter.update_cpd(args.w, 1.0, args.steps)
        sync()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        alpha, sc, fit = fitter.get_state()
        ok = bool(np.all(np.isfinite(fit)) and sc.status == 0 and sc.iteration == args.warmup + args.steps
                  and args.emulate_world <= 1)

        # ---- live roofline of the dominant kernels (HIP events on the kernels' stream, extra iterations)
        roof = None
        kernels = []
        ctx.timing_enable(True)
        ctx.timing_reset()
        fitter.update_cpd(args.w, 1.0, args.roofline_steps)
        sync()
        m_loc = fitter.end - fitter.begin
        pairs = float(m_loc) * float(N)
        names = {0: ("cpd_colsum_kernel", 11.0), 1: ("cpd_rowstats_kernel", 18.0)}
        for which, (name, flops_per_pair) in names.items():
            ms, n = ctx.timing_read(which)
            if n:
                avg = ms / n
                ach = flops_per_pair * pairs / (avg * 1e-3) / 1e12
                kernels.append({"kernel": name, "avg_ms": avg, "launches": n, "bound": "valu_f64",
                                "achieved": ach, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / F64_VALU_PEAK_TFLOPS, "algorithmic_flops_per_pair": flops_per_pair})
        ms, n = ctx.timing_read(2)
        if n:
            avg = ms / n
            rp = (args.rank + 15) // 16 * 16
            ach = 6.0 * m_loc * rp * rp / (avg * 1e-3) / 1e12
            kernels.append({"kernel": "gram_kernel", "avg_ms": avg, "launches": n, "bound": "mfma",
                            "achieved": ach, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / F64_MFMA_PEAK_TFLOPS})
        ms, n = ctx.timing_read(4)
        if n:
            avg = ms / n
            rp = (args.rank + 15) // 16 * 16
            gbs = 24.0 * m_loc * rp / (avg * 1e-3) / 1e9
            kernels.append({"kernel": "sweep_kernel", "avg_ms": avg, "launches": n, "bound": "hbm", "achieved": gbs,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "algorithmic_bytes": 24.0 * m_loc * rp,
                            "traffic": "FETCH_SIZE x2 (gfx950 correction) = 134 MB at 50k, r=100: profiles/r01_pmc_traffic.md"})
        ms, n = ctx.timing_read(5)
        if n:
            kernels.append({"kernel": "posterior_solve_lds_kernel", "avg_ms": ms / n, "launches": n, "bound": "latency"})
        ms, n = ctx.timing_read(3)
        upd_ms = ms / n if n else None
        ctx.timing_enable(False)
        dom = max((k for k in kernels if k["bound"] == "valu_f64"), key=lambda k: k["avg_ms"], default=None)
        if dom is not None:
            roof = {"bound": "valu_f64", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": dom["peak"],
                    "unit": "TFLOP/s", "frac": dom["frac"],
                    # PMC FETCH_SIZE + WRITE_SIZE per launch (separate passes, profiles/r01_pmc_traffic.md), default workload only:
                    # 13 MB read + 157 MB of per-chunk partial sums written (98 chunks x 4 planes x 50k rows), hidden under 1.3 ms of
                    # VALU work and read back once by rowstats_reduce_kernel
                    "traffic": 170.0e6 if (M == 50000 and N == 50000 and world == 1 and not args.emulate_world
                                           and dom["kernel"] == "cpd_rowstats_kernel") else None,
                    "nearest_contract_bound": "mfma",
                    "note": "all-pairs kernel: O(M+N) bytes, O(M*N) float64 VALU flops (software exp counted as 1 flop); "
                            "HBM and MFMA are not the binding resource.  In the contract's hbm|mfma vocabulary this is the "
                            "compute side: the peak used, 78.6 TFLOP/s, is also the dense f64 MFMA peak -- on gfx950 the f64 "
                            "vector and matrix pipes share the issue slots (profiles/r01_ubench_mfma_valu_overlap.txt), so "
                            "moving the K=3 contraction to MFMA does not raise the ceiling (GINGR_AFFINITY=mfma measures it)"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.emulate_world:
        cpu = cpu_baseline(y, x, sigma2_0, args.w)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "GiNGR update iters/sec, 50k<->50k CPD",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
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
                       "points": M, "targets": N, "rank": args.rank, "parallelism": f"row-shard x{world}",
                       "emulated_world": args.emulate_world or None},
            "valid": ok,
            "sigma2_after_timed_steps": float(sc.sigma2),
            "update_ms_device": upd_ms,
            "roofline": roof,
            "kernels": kernels,
            "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
    fitter.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so that the JSON is the LAST line on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    return


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One reproducible measurement line per BASELINE.json config (`python bench.py --config N` prints line N; run as a script it
runs all five (+ the metric workload at SURVEY 8d's second model rank, 256) and writes gpurun_out/r06_configs.json -- copy it to profiles/).

  1  femur CPD, 1 622 <-> 1 622 vertices of the reference's own demo data (tests/golden/inputs.npz), Gaussian GPMM (70, 50) built on the
     device, DemoCPD settings (examples/DemoCPD.scala:11-25: CpdConfiguration defaults, NoTransforms) -- the reference's CPU-runnable case
  2  bunny closest point, 5 000 vertices, distance + argmin KERNEL ONLY (ClosestPointRegistrator.scala:139-145), with the number of
     distance tests the kernel really executed, so that a roofline fraction exists
  3  15 000 <-> 15 000 CPD update (soft assignment + GP posterior), synthetic clouds of SURVEY 8d
  4  100 000 <-> 100 000 CPD update: one GPU, plus the per-rank cost of an 8-rank row shard emulated on this GPU
  5  Metropolis-Hastings chain on the femur pair (DemoICP settings), steps/s of one chain (8 chains = replicas, one per GPU)

Every line has the shape of the headline bench line: metric / value / unit / config.workload / dtype / valid (parity against the
oracle) / roofline / cpu_baseline where one is meaningful.  oracle/ is the checker and the reported CPU baseline, never the product."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the oracle's OpenMP / BLAS threads must sleep, not spin, once a CPU leg is over: spinning threads slow the thread that feeds the GPU
# (measured: 0.32 instead of 0.11 ms per femur iteration when the numpy oracle had run in the same process before the timed loop)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
GOLD = os.path.join(ROOT, "tests", "golden")
F64_PEAK = 78.6  # TFLOP/s, vector = matrix float64 peak of MI355X


def _bench(*args, timeout=1800):
    """bench.py with the given arguments as a child process; its JSON line"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-2000:])
    return json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][-1])


def config1():
    import torch  # noqa: F401  (first: one HIP runtime per process)
    import gingr_amd as ga
    from gingr_amd.sharded import ShardedFitter
    from oracle import gingr_oracle as go
    d = np.load(os.path.join(GOLD, "inputs.npz"))
    ref, target = d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)
    ctx = ga.Context(0)
    model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.01).Gaussian(sigma=70.0, scaling=50.0)
    rank = int(model.rank)
    s2 = ctx.cpd_initial_sigma2(ref, target)
    w, lam = 0.0, 1.0                                               # CpdConfiguration defaults (CPD.scala:21-29)
    f = ShardedFitter(ctx, model, target, global_transform=ga.GlobalTranformationType.NoTransforms, step_length=1.0)
    # timing first (device-resident iterations), the oracle afterwards
    n = 300
    f.set_state(np.zeros(rank), s2)
    f.update_cpd(w, lam, 10)
    ctx.synchronize()
    f.set_state(np.zeros(rank), s2)
    ctx.synchronize()
    t0 = time.perf_counter()
    f.update_cpd(w, lam, n)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    _, sc, _ = f.get_state()
    ctx.timing_enable(True)
    ctx.timing_reset()
    f.update_cpd(w, lam, 5)
    ctx.synchronize()
    kern = {}
    for which, name in ((0, "cpd_colsum_kernel"), (1, "cpd_rowstats_kernel"), (2, "gram_kernel"), (5, "posterior_solve_lds_kernel")):
        ms, k = ctx.timing_read(which)
        if k:
            kern[name] = ms / k
    ctx.timing_enable(False)
    # parity: five updates against the oracle's trajectory from the same start, on the model the device built (downloaded)
    host = model.to_host(basis=True)
    mo = go.PDM(ref, np.zeros_like(ref), np.asarray(host.basis, dtype=np.float64), np.asarray(host.variance, dtype=np.float64))
    f.set_state(np.zeros(rank), s2)
    f.update_cpd(w, lam, 5)
    a5, sc5, fit5 = f.get_state()
    st = go.initial_state(mo, s2, global_transformation=go.NO_TRANSFORMS)
    t0 = time.perf_counter()
    for _ in range(5):
        st = go.cpd_update(mo, target, st, w=w, lam=lam)
    cpu_s = (time.perf_counter() - t0) / 5
    err = float(np.linalg.norm(fit5 - st.fit) / np.linalg.norm(st.fit))
    parity = {"against": "oracle/gingr_oracle.py (numpy restatement, parity unpinned), 5 updates from the same start",
              "fit_rel_l2": err, "sigma2_rel": float(abs(sc5.sigma2 - st.sigma2) / st.sigma2), "tolerance": 1e-5, "ok": bool(err < 1e-5)}
    M = N = ref.shape[0]
    roof = None
    if "cpd_rowstats_kernel" in kern:
        ach = 18.0 * M * N / (kern["cpd_rowstats_kernel"] * 1e-3) / 1e12
        roof = {"bound": "valu_f64", "kernel": "cpd_rowstats_kernel", "achieved": ach, "peak": F64_PEAK, "unit": "TFLOP/s",
                "frac": ach / F64_PEAK, "traffic": None,
                "note": "at this size the step is a chain of latency-bound launches (posterior solve %.0f us of %.0f us); 2.6e6 pairs "
                        "occupy the chip for ~1 us" % (kern.get("posterior_solve_lds_kernel", 0) * 1e3, dt / n * 1e6)}
    out = {"config": 1, "metric": "GiNGR update iters/sec, femur CPD 1622<->1622", "value": n / dt, "unit": "iterations/s", "n_gpus": 1,
           "steps": n, "ms_per_step": dt / n * 1e3, "higher_is_better": True, "dtype": "f64", "data": "reference demo data (femur STL pair)",
           "config_detail": {"workload": f"CPD update, femur {M}<->{N}, Gaussian GPMM (70, 50) rank {rank} built on the device, w=0, "
                                          f"NoTransforms (DemoCPD), sigma2_0={s2:.3f}, device-resident iterations"},
           "valid": bool(sc.status == 0 and parity["ok"]), "parity_check": parity, "roofline": roof, "kernels_ms": kern,
           "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "iterations/s", "cores": 1, "kind": "port",
                            "sample": "5 full updates of the numpy oracle (algorithm-faithful: one affinity evaluation per update, not the "
                                      "four P materialisations of the stock plugin; that structure: bench.py --config1-stock-structure)"}}
    f.close()
    ctx.close()
    return out


def config2():
    import torch  # noqa: F401
    import gingr_amd as ga
    from oracle import c_oracle as co
    d, e = np.load(os.path.join(GOLD, "inputs.npz")), np.load(os.path.join(GOLD, "expected.npz"))
    target, query = d["bunny5k"].astype(np.float64), e["nn_query"].astype(np.float64)
    M, N = query.shape[0], target.shape[0]
    ctx = ga.Context(0)
    idx, d2, md = ctx.nn(query, target)                              # warm-up + parity
    exact = bool(np.array_equal(idx, e["nn_idx"]))
    # timing and counting in separate passes: the counting instantiation of the kernel ends every wave with an atomic on ONE word
    # (8 000 serialised atomics are ~90 us, ten times the kernel itself)
    ctx.timing_enable(True)
    ctx.timing_reset()
    reps = 50
    for _ in range(reps):
        ctx.nn(query, target)
    ms, k = ctx.timing_read(8)
    ctx.timing_enable(False)
    avg_ms = ms / reps                                                # per CALL (all search launches of a call)
    variants = {}
    from gingr_amd import _native as nat
    for name, opt, val in (("all_pairs_scan_unordered", nat.OPT_CULL, 0), ("per_call_grid_cold_start", nat.OPT_NN_GRID, 2)):
        c2 = ga.Context(0)
        c2.set_option(opt, val)
        i2, _, _ = c2.nn(query, target)
        c2.timing_enable(True)
        c2.timing_reset()
        for _ in range(reps):
            c2.nn(query, target)
        ms2, k2 = c2.timing_read(8)
        variants[name] = {"us_per_call": ms2 / reps * 1e3, "launches_per_call": int(k2 // reps), "indices_equal": bool(np.array_equal(i2, idx))}
        c2.close()
    ctx.nn_counting(True)
    ctx.nn(query, target)
    tests = float(ctx.nn_tests())
    ctx.nn_counting(False)
    ach = 9.0 * tests / (avg_ms * 1e-3) / 1e12
    t0 = time.perf_counter()
    co.nn(query, target)
    cpu_s = time.perf_counter() - t0
    # for context only (not the metric): the search an ICP registration runs on the same pair -- grid search over the fixed target +
    # masked tile scan (nn_grid.hip), warm-started from the previous iteration; the queries are the fit of a small model of the bunny
    reg = None
    try:
        rng = np.random.default_rng(0)
        r = 8
        U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, r)))
        model = ga.PointDistributionModel(query, np.zeros_like(query), U, np.linspace(4.0, 0.5, r))
        algo = ga.IcpRegistration(ctx)
        cfg = ga.IcpConfiguration(maxIterations=50, initialSigma=1.0, endSigma=0.5, correspondenceMethod="PointcloudClosestPoint")
        state = algo.createInitialState(model, target, cfg)
        for _ in range(3):
            state = algo.update(state)
        ctx.timing_enable(True)
        ctx.timing_reset()
        n_it = 20
        for _ in range(n_it):
            state = algo.update(state)
        ms_r, k_r = ctx.timing_read(8)
        ctx.timing_enable(False)
        ctx.nn_counting(True)
        state = algo.update(state)
        ctx.synchronize()
        tests_r = float(ctx.nn_tests())
        ctx.nn_counting(False)
        algo.close()
        reg = {"what": "closest-point search inside IcpRegistration.update on the same pair (grid search + masked tile scan, both launches)",
               "us_per_search": ms_r / n_it * 1e3, "queries_per_s": M / (ms_r / n_it * 1e-3), "distance_tests_per_search": tests_r,
               "launches_timed": int(k_r)}
    except Exception as ex:  # context only: never fails the config line
        reg = {"error": repr(ex)}
    out = {"config": 2, "metric": "closest-point queries/sec, bunny 5k (distance + argmin kernel only)", "value": M / (avg_ms * 1e-3),
           "unit": "queries/s", "n_gpus": 1, "steps": reps, "ms_per_step": avg_ms, "higher_is_better": True, "dtype": "f64",
           "data": "reference demo data (bunny PLY, 5 000 vertices sub-sampled, seed 7) + perturbed copy as queries",
           "config_detail": {"workload": f"gingr_nn kernels, {M} queries x {N} targets, exact f64 distances (separately rounded products), "
                                          "lowest index on ties; stateless entry point (round 4): all pairs straight from the caller's order in "
                                          "~770 workgroups (nn_small_kernel: targets through LDS, 9 instructions per pair) + the combination of "
                                          "their slices (nn_small_reduce_kernel); BOTH launches are inside the timed region",
                             "launches_per_call": 2, "other_variants": variants},
           "valid": exact, "parity_check": {"against": "tests/golden/expected.npz nn_idx (oracle brute force)", "indices_bit_exact": exact,
                                            "mean_distance": md},
           "roofline": {"bound": "valu_f64", "kernel": "nn_small_kernel + nn_small_reduce_kernel", "achieved": ach, "peak": F64_PEAK, "unit": "TFLOP/s",
                        "frac": ach / F64_PEAK, "traffic": None, "distance_tests_per_call": tests, "all_pairs": float(M) * N,
                        "algorithmic_flops_per_test": 9.0, "all_pairs_equivalent_tflops": 9.0 * float(M) * N / (avg_ms * 1e-3) / 1e12,
                        "note": "all pairs, 9 flop each in 9 vector instructions (separately rounded products and sums + one minimum): the "
                                "instruction floor of 25 M pairs is ~7 us on 1 024 SIMDs; the first launch takes ~12 us (LDS fill and "
                                "dispatch of a one-round launch exposed), the combination ~5 us; rounds 1-3 timed the first launch of the "
                                "tile scan only (30 us; its reduction was outside the timed region)"},
           "cpu_baseline": {"value": M / cpu_s, "unit": "queries/s", "cores": 1, "kind": "port", "sample": "all 5 000 queries, oracle/cpd_oracle.c"},
           "registration_path": reg}
    ctx.close()
    return out


def config3():
    o = _bench("--points", "15000", "--steps", "100", "--warmup", "10")
    o["config"] = 3
    o["metric"] = "GiNGR update iters/sec, 15k<->15k CPD"
    return o


def config4():
    o = _bench("--points", "100000", "--steps", "20", "--warmup", "3")
    o["config"] = 4
    o["metric"] = "GiNGR update iters/sec, 100k<->100k CPD"
    e = _bench("--points", "100000", "--emulate-world", "8", "--no-cpu-baseline", "--no-parity-check", "--steps", "40", "--warmup", "5",
               "--roofline-steps", "0")
    o["emulated_rank_of_8"] = {"ms_per_step": e["ms_per_step"], "note": "ONE GPU runs rank 0's row shard of an 8-rank job (1-rank RCCL "
                               "exchange): per-rank cost before real xGMI latency; no 8-GPU node was available to this round"}
    return o


def run_chains(n: int, devices, steps: int = 3000):
    """n Metropolis-Hastings chains side by side, chain i in its own process with HIP_VISIBLE_DEVICES=devices[i] (the parent never
    touches a GPU), seeds 0..n-1, chain loops released together through a file barrier.  GingrAlgorithm.run is one sequential loop
    per chain (G/api/GingrAlgorithm.scala:115-175), so 8 chains are 8 independent replicas -- nothing is exchanged."""
    import tempfile
    bdir = tempfile.mkdtemp(prefix="gingr_chains_")
    procs = []
    for i in range(n):
        env = dict(os.environ)
        env["HIP_VISIBLE_DEVICES"] = str(devices[i])
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "bench_mh_chain.py"), str(steps), str(i), f"barrier={bdir}",
                                       f"index={i}"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    t_wait = time.time()
    while not all(os.path.exists(os.path.join(bdir, f"ready_{i}")) for i in range(n)):
        if any(p.poll() is not None for p in procs) or time.time() - t_wait > 900:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            errs = [p.communicate()[1][-600:] for p in procs]
            raise RuntimeError("a chain process ended before the start barrier: " + " | ".join(errs))
        time.sleep(0.01)
    t0 = time.time()
    open(os.path.join(bdir, "go"), "w").close()
    outs = [p.communicate(timeout=1800) for p in procs]
    wall = time.time() - t0
    lines = []
    for p, (so, se) in zip(procs, outs):
        if p.returncode != 0:
            raise RuntimeError(se[-2000:])
        lines.append(json.loads([ln for ln in so.splitlines() if ln.strip().startswith("{")][-1]))
    starts = [c["loop_started_unix"] for c in lines]
    ends = [c["loop_started_unix"] + c["loop_seconds"] for c in lines]
    overlap = max(0.0, min(ends) - max(starts)) / max(1e-9, max(ends) - min(starts))
    return {"chains": n, "devices": [str(d) for d in devices[:n]], "steps_per_chain": steps,
            "per_chain_steps_per_s": [round(c["steps_per_s"], 1) for c in lines],
            "aggregate_steps_per_s": sum(c["steps_per_s"] for c in lines),
            "aggregate_steps_per_s_wall": n * steps / (max(ends) - min(starts)),
            "loops_overlap_fraction": overlap, "wall_s_release_to_exit": wall,
            "device_uuids": [c["device"]["uuid"] for c in lines], "distinct_device_uuids": len({c["device"]["uuid"] for c in lines}),
            "statuses": [c["status"] for c in lines], "log_value_best": [c["log_value_best"] for c in lines],
            "log_value_initial": lines[0]["log_value_initial"]}


def config5(gpus: int = 1):
    # how many devices are there?  (device_count does not initialise the GPU: this process stays a pure launcher)
    import torch
    visible = torch.cuda.device_count()
    multi = None
    if gpus > 1 and visible >= gpus:
        multi = dict(run_chains(gpus, list(range(gpus))), mode="one chain per GPU (BASELINE config 5 as stated)")
    elif gpus > 1:
        multi = {"mode": f"asked for {gpus} GPUs, {visible} visible: the chains are PACKED on device 0 (they share the GPU: not config 5 as "
                         "stated, a protocol run of the same launcher)",
                 **run_chains(gpus, [0] * gpus)}
    packed = None
    if visible == 1:    # one-GPU box: what several replicas sharing the device give (a chain leaves most of the GPU idle)
        packed = [run_chains(n, [0] * n) for n in (2, 4, 8)]

    def chain(*extra):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_mh_chain.py"), "300", "0", *extra], capture_output=True,
                             text=True, timeout=1800, cwd=ROOT)
        if out.returncode != 0:
            raise RuntimeError(out.stderr[-2000:])
        return json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][-1])
    c = chain()                 # one native call per step (gingr_fitter_mh_step), the default of GingrAlgorithm.run
    slow = chain("nofuse")      # the call-by-call path of rounds 1-3: same draws, same decisions
    return {"config": 5, "metric": "MH-in-GiNGR chain steps/sec, femur (one chain)", "value": c["steps_per_s"], "unit": "steps/s", "n_gpus": 1,
            "steps": c["steps"], "ms_per_step": c["ms_per_step"], "higher_is_better": True, "dtype": "f64",
            "data": "reference demo data (femur STL pair)",
            "config_detail": {"workload": "Metropolis-Hastings chain, surface-ICP proposals (posterior sample) + random walks, DemoICP settings; "
                                          "one native call per step (proposal, likelihood, transition density; round 4); "
                                          "8 chains = 8 independent replicas, one context per GPU, no communication",
                              "call_by_call_steps_per_s": slow["steps_per_s"],
                              "same_best_sample_as_call_by_call": bool(slow["log_value_best"] == c["log_value_best"])},
            # the chain ends with MaxIteration (2) or Converged (1); 3 = ModelFlexibilityError
            "valid": bool(c["status"] != 3 and c["log_value_best"] >= c["log_value_initial"]
                          and slow["log_value_best"] == c["log_value_best"]),
            "parity_check": {"against": "tests/test_gpu_sampling.py: a 25-step chain reproduces the oracle's accept / reject sequence; "
                                        "tests/test_gpu_mh_step.py: fused steps = call-by-call steps (decisions, states <= 1e-9)",
                             "log_value_initial": c["log_value_initial"], "log_value_best": c["log_value_best"]},
            "roofline": None, "cpu_baseline": None, "detail": c,
            # `--config 5 --gpus N`: N chain processes, one per GPU; on a one-GPU box also 2 / 4 / 8 replicas packed on the device
            "chains_one_per_gpu": multi, "chains_packed_on_one_gpu": packed}


def metric_rank256():
    """The metric workload (50k <-> 50k CPD) at SURVEY 8d's second model rank, r = 256: bench.py's own line (child process: bench.py
    owns its context), with the kernels of the wide path (gram_wide_kernel, posterior_solve_wide_kernel, sweep_fit_boxes_kernel<16, 1>)."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rank", "256", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=1200, cwd=ROOT)
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    d["config"] = "metric workload at model rank 256 (SURVEY 8d: r in {100, 256})"
    return d


CONFIGS = {1: config1, 2: config2, 3: config3, 4: config4, 5: config5, 6: metric_rank256}


def main():
    which = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 6]
    lines = []
    for c in which:
        try:
            o = CONFIGS[c]()   # (config 5 with --gpus N: python bench.py --config 5 --gpus N)
        except Exception as ex:  # a failing config must not hide the others
            o = {"config": c, "error": repr(ex)[:2000], "valid": False}
        lines.append(o)
        print(json.dumps(o), flush=True)
    if len(which) > 1:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(lines, open(os.path.join(ROOT, "gpurun_out", "r06_configs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

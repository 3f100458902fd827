"""GPU tests against the committed fixtures, logical row shards on one device, and BASELINE-size property checks.

The fixtures under tests/golden/*.npz are REGRESSION PINS: inputs taken from the reference's demo data, expected values produced
by this repository's own oracle (tests/golden/make_golden*.py) -- they catch drift of the HIP path and of the oracle, not a
misreading shared by both.  Truth derived from the real reference would come from tests/golden/reference/ (see
tests/test_reference_golden.py); none exists yet: parity unpinned."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "inputs.npz")), np.load(os.path.join(GOLD, "expected.npz"))


# ------------------------------------------------------------------------------------------ committed fixtures
def test_golden_femur_cpd_stats(ctx, gold):
    inp, exp = gold
    y, x = inp["femur"].astype(np.float64), inp["femur_target"].astype(np.float64)
    assert abs(ctx.cpd_initial_sigma2(y, x) - float(exp["femur_sigma2_init"])) < 1e-9
    for tag in ("s1_w01", "sinit_w0", "s25_w0"):
        s2, w = exp[f"cpd_{tag}_args"]
        got = ctx.cpd_stats(y, x, float(s2), float(w))
        assert np.allclose(got["den"], exp[f"cpd_{tag}_den"], rtol=1e-11)
        assert np.allclose(got["P1"], exp[f"cpd_{tag}_P1"], rtol=1e-10)
        assert rel(got["PX"], exp[f"cpd_{tag}_PX"]) < 1e-11
        assert abs(got["Np"] - exp[f"cpd_{tag}_scalars"][0]) < 1e-9 * abs(got["Np"])
        assert abs(got["sigma2_next"] - exp[f"cpd_{tag}_scalars"][1]) < 1e-8 * abs(got["sigma2_next"])


def test_golden_bunny_nn_bit_exact(ctx, gold):
    inp, exp = gold
    idx, _, md = ctx.nn(exp["nn_query"].astype(np.float64), inp["bunny5k"].astype(np.float64))
    assert np.array_equal(idx, exp["nn_idx"])
    assert abs(md - float(exp["nn_mean_distance"])) < 1e-12


def _femur_model(gold):
    import gingr_amd as ga
    inp, exp = gold
    y = inp["femur"].astype(np.float64)
    return ga.PointDistributionModel(y, np.zeros_like(y), exp["gpmm_basis"], exp["gpmm_variance"])


@pytest.mark.parametrize("tag,w,use_lm", [("cpd_rigid", 0.0, False), ("cpd_rigid_lm_w", 0.1, True)])
def test_golden_femur_cpd_trajectory(ctx, gold, tag, w, use_lm):
    import gingr_amd as ga
    inp, exp = gold
    x = inp["femur_target"].astype(np.float64)
    model = _femur_model(gold)
    lms = None
    if use_lm:
        lms = ga.LandmarkCorrespondences(exp["lm_pids"], inp["femur_target_lm"], np.tile(np.eye(3), (len(exp["lm_pids"]), 1, 1)))
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(model, x, ga.CpdConfiguration(maxIterations=50, w=w), landmarks=lms)
    assert abs(state.general.sigma2 - float(exp["femur_sigma2_init"])) < 1e-9
    for it in range(1, 6):
        state = algo.update(state)
        if it in (1, 2, 5):
            g = state.general
            pose = exp[f"{tag}_it{it}_pose"]
            assert g.status == int(pose[8]) == 0
            assert rel(g.modelParameters.shape, exp[f"{tag}_it{it}_alpha"]) < 1e-5
            assert np.allclose([g.modelParameters.rotation.phi, g.modelParameters.rotation.theta, g.modelParameters.rotation.psi],
                               pose[0:3], atol=1e-9)
            assert np.allclose(g.modelParameters.translation, pose[3:6], atol=1e-7)
            assert abs(g.sigma2 - pose[7]) < 1e-8 * pose[7]
    assert rel(state.general.fit, exp[f"{tag}_it5_fit"]) < 1e-5      # north_star tolerance on vertex positions
    algo.close()


def test_golden_femur_icp(ctx, gold):
    import gingr_amd as ga
    inp, exp = gold
    x = inp["femur_target"].astype(np.float64)
    algo = ga.IcpRegistration(ctx)
    state = algo.createInitialState(_femur_model(gold), x, ga.IcpConfiguration(maxIterations=10, initialSigma=100.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint"),
                                    transform=ga.GlobalTranformationType.NoTransforms)
    for _ in range(3):
        state = algo.update(state)
    assert np.array_equal(algo.last_correspondence_indices(), exp["icp_it3_idx"])     # bit-exact correspondence indices
    assert rel(state.general.fit, exp["icp_it3_fit"]) < 1e-5
    assert state.general.sigma2 == float(exp["icp_it3_sigma2"])
    algo.close()


# ------------------------------------------------------------------------------------------ logical shards, one device
@pytest.mark.parametrize("nshards", [2, 3])
def test_logical_row_shards_equal_single_shard(ctx, nshards):
    import torch
    import gingr_amd as ga
    from gingr_amd.sharded import NUM_PHASES, NUM_SEGMENTS, ShardedFitter
    import ctypes
    from gingr_amd import _native as nat
    rng = np.random.default_rng(77)
    ref = rng.normal(0, 40, (901, 3))
    mo = go.build_gaussian_gpmm(ref, 60.0, 30.0, rel_tol=1e-9, max_rank=40)
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
    target = mo.instance(rng.normal(0, 1, mo.rank))[:800] + rng.normal(0, 0.3, (800, 3)) + 1.0
    s2 = ctx.cpd_initial_sigma2(mo.ref, target)

    single = ShardedFitter(ctx, model, target)
    single.set_state(np.zeros(mo.rank), s2)
    single.update_cpd(0.1, 1.0, 3)
    a1, sc1, fit1 = single.get_state()

    shards = [ShardedFitter(ctx, model, target, rank=r, world=nshards, all_reduce=None, defer_setup=True) for r in range(nshards)]

    def allreduce(tensors):
        ctx.synchronize()
        tot = torch.stack(tensors).sum(0)
        for t in tensors:
            t.copy_(tot)
        torch.cuda.synchronize()

    allreduce([s.gram_tensor() for s in shards])
    for s in shards:
        s.finish_setup()
        s.set_state(np.zeros(mo.rank), s2)
    p = nat.CpdParams(0.1, 1.0)
    for _ in range(3):
        for ph in range(NUM_PHASES):
            for s in shards:
                rc = s._lib.gingr_fitter_cpd_phase_async(s.handle, ctypes.byref(p), ph)
                assert rc == 0
            if ph < NUM_SEGMENTS:
                allreduce([s._segment(ph) for s in shards])
    fits, alphas = [], []
    for s in shards:
        a, sc, fit = s.get_state()
        fits.append(fit)
        alphas.append(a)
        assert sc.iteration == 3 and sc.status == 0
        assert abs(sc.sigma2 - sc1.sigma2) < 1e-10 * sc1.sigma2
    assert rel(np.concatenate(fits), fit1) < 1e-9
    for a in alphas:
        assert rel(a, a1) < 1e-7
    for s in shards + [single]:
        s.close()


def test_single_rank_nccl_process_group_path(ctx):
    """The exact plumbing bench.py uses for N > 1 (torch.distributed all_reduce on tensors aliasing the exchange buffer,
    library kernels on a torch stream), exercised with a world of one rank."""
    import torch
    import torch.distributed as dist
    import gingr_amd as ga
    from gingr_amd.sharded import ShardedFitter
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        rng = np.random.default_rng(3)
        ref = rng.normal(0, 40, (500, 3))
        mo = go.build_gaussian_gpmm(ref, 60.0, 30.0, rel_tol=1e-9, max_rank=24)
        model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
        target = mo.instance(rng.normal(0, 1, mo.rank)) + rng.normal(0, 0.3, (500, 3))
        c2 = ga.Context(0)
        stream = torch.cuda.Stream()
        c2.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            f = ShardedFitter(c2, model, target, rank=0, world=1)
            f.world = 2                      # force the phase + all_reduce path
            f.all_reduce = lambda t: dist.all_reduce(t)
            from gingr_amd.sharded import as_torch
            import ctypes
            from ctypes import c_int64, c_void_p
            from gingr_amd.sharded import NUM_SEGMENTS
            p = c_void_p(); offs = (c_int64 * NUM_SEGMENTS)(); cnts = (c_int64 * NUM_SEGMENTS)()
            assert f._lib.gingr_fitter_exchange(f.handle, ctypes.byref(p), offs, cnts) == 0
            f.xch = as_torch(p.value, offs[NUM_SEGMENTS - 1] + cnts[NUM_SEGMENTS - 1], 0)
            s2 = c2.cpd_initial_sigma2(mo.ref, target)
            f.set_state(np.zeros(mo.rank), s2)
            f.update_cpd(0.0, 1.0, 2)
            torch.cuda.synchronize()
            a, sc, fit = f.get_state()
        st = go.initial_state(mo, s2)
        for _ in range(2):
            st = go.cpd_update(mo, target, st)
        assert rel(fit, st.fit) < 1e-5 and sc.iteration == 2
        f.close()
        c2.close()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ BASELINE-size properties
def test_50k_properties_and_sampled_oracle(ctx):
    """50k <-> 50k (the metric size): size-independent properties plus a row/column sample against the C oracle."""
    rng = np.random.default_rng(1234)
    N = M = 50000
    x = rng.normal(0, 50, (N, 3)).astype(np.float32).astype(np.float64)
    y = x[rng.permutation(N)] + rng.normal(0, 2, (M, 3))
    s2 = 40.0
    got0 = ctx.cpd_stats(y, x, s2, 0.0)
    assert np.max(np.abs(got0["Pt1"] - 1.0)) < 1e-12 and abs(got0["Np"] - N) < 1e-7        # w = 0: columns sum to one
    got = ctx.cpd_stats(y, x, s2, 0.1)
    # column sample: den_j against the oracle over ALL rows
    cols = rng.choice(N, 48, replace=False)
    den_s = co.cpd_colsum_partial(y, x[cols], s2, 0, M) + co.outlier_constant(M, N, s2, 0.1)
    assert np.allclose(got["den"][cols], den_s, rtol=1e-11)
    # row sample: P1_i / PX_i against the oracle over ALL columns (using the verified den)
    rows = rng.choice(M, 48, replace=False)
    P1s, PXs = co.cpd_rowstats_partial(y[rows], x, s2, got["den"], 0, len(rows))
    assert np.allclose(got["P1"][rows], P1s, rtol=1e-10) and np.allclose(got["PX"][rows], PXs, rtol=1e-9, atol=1e-12)
    # translation invariance of P (hence of P1) and equivariance of PX
    shift = np.array([7.0, -3.0, 2.0])
    got_t = ctx.cpd_stats(y + shift, x + shift, s2, 0.1)
    assert np.allclose(got_t["P1"], got["P1"], rtol=1e-9)
    assert np.allclose(got_t["PX"], got["PX"] + got["P1"][:, None] * shift, rtol=1e-8, atol=1e-9)
    # Np = sum P1 = sum Pt1 (checksum of checksums)
    assert abs(got["P1"].sum() - got["Pt1"].sum()) < 1e-8 * got["Np"]
    # nearest neighbour at full size: sampled rows against the oracle, and idempotence on the target itself
    idx, d2, _ = ctx.nn(y, x)
    si, sd2, _ = co.nn(y[rows], x)
    assert np.array_equal(idx[rows], si) and np.array_equal(d2[rows], sd2)
    idx_self, d2_self, _ = ctx.nn(x[:5000], x)
    assert np.array_equal(idx_self, np.arange(5000)) and np.all(d2_self == 0.0)


def test_two_logical_shards_through_sharded_fitter_update_itself(ctx):
    """ShardedFitter.update_cpd / update_icp -- the driver loop bench.py uses for N > 1, including its stream contract (the
    contexts run on their own streams here, torch's collective stand-in on the current stream: every exchange must be
    bracketed by synchronisation) -- on two logical shards in two threads with a barrier all-reduce."""
    import threading
    import torch
    import gingr_amd as ga
    from gingr_amd.sharded import ShardedFitter
    rng = np.random.default_rng(78)
    ref = rng.normal(0, 40, (777, 3))
    mo = go.build_gaussian_gpmm(ref, 60.0, 30.0, rel_tol=1e-9, max_rank=32)
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
    target = mo.instance(rng.normal(0, 1, mo.rank))[:700] + rng.normal(0, 0.3, (700, 3)) + 1.0
    s2 = ctx.cpd_initial_sigma2(mo.ref, target)
    single = ShardedFitter(ctx, model, target)
    single.set_state(np.zeros(mo.rank), s2)
    single.update_cpd(0.1, 1.0, 3)
    single.update_icp(20.0, 1.0, 10, 2)
    a1, sc1, fit1 = single.get_state()
    single.close()

    world = 2
    ctxs = [ga.Context(0) for _ in range(world)]              # own non-blocking streams: NOT torch's current stream
    barrier = threading.Barrier(world)
    slots = [None] * world
    results, errors = [None] * world, []

    def make_all_reduce(r):
        def all_reduce(t):
            slots[r] = t
            barrier.wait()
            tot = torch.stack([slots[q] for q in range(world)]).sum(0)   # same order on both shards: bit-identical sums
            barrier.wait()                                               # everybody has read every partial
            t.copy_(tot)
            torch.cuda.synchronize()
        return all_reduce

    def work(r):
        try:
            f = ShardedFitter(ctxs[r], model, target, rank=r, world=world, all_reduce=make_all_reduce(r))
            assert not f._same_stream()
            f.set_state(np.zeros(mo.rank), s2)
            f.update_cpd(0.1, 1.0, 3)
            f.update_icp(20.0, 1.0, 10, 2)
            results[r] = f.get_state()
            f.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    fit = np.concatenate([results[r][2] for r in range(world)])
    for r in range(world):
        a, sc, _ = results[r]
        assert sc.iteration == 5 and sc.status == 0 and abs(sc.sigma2 - sc1.sigma2) < 1e-10 * sc1.sigma2
        assert rel(a, a1) < 1e-7
    assert results[0][1].sigma2 == results[1][1].sigma2          # the replicated state is bit-identical across shards
    assert rel(fit, fit1) < 1e-9
    for c in ctxs:
        c.close()


def test_native_rccl_exchange_equals_the_single_shard():
    """The library's own RCCL exchange (Context.rccl_init + gingr_fitter_update_{cpd,icp}_rccl_async) with a one-rank communicator:
    unique id, communicator, the one-off moment all-reduce and the two per-iteration ncclAllReduce calls on the context's stream must
    leave exactly the state the plain single-shard update reaches (a one-rank in-place sum is the identity)."""
    import gingr_amd as ga
    from gingr_amd.sharded import ShardedFitter
    rng = np.random.default_rng(5)
    ref = rng.normal(0, 50.0, (3000, 3)).astype(np.float32).astype(np.float64)
    mo = go.build_gaussian_gpmm(ref, 70.0, 50.0, rel_tol=1e-12, max_rank=24)
    to_ga = lambda m: ga.PointDistributionModel(m.ref, m.mean, m.U, m.lam)
    target = mo.ref[rng.permutation(mo.M)[:2800]] + rng.normal(0, 0.5, (2800, 3))
    states = {}
    for mode in ("plain", "rccl"):
        ctx = ga.Context(0)
        if mode == "rccl":
            ctx.rccl_init(ctx.rccl_unique_id(), 1, 0)
            info = ctx.rccl_info()
            assert info["world"] == 1 and info["rank"] == 0 and info["version"] > 0
        f = ShardedFitter(ctx, to_ga(mo), target, rank=0, world=1, rccl=(mode == "rccl"))
        f.set_state(np.zeros(mo.rank), 30.0)
        f.update_cpd(0.1, 1.0, 3)
        f.update_icp(2.0, 1.0, 10, 2)
        ctx.synchronize()
        a, sc, fit = f.get_state()
        states[mode] = (a.copy(), float(sc.sigma2), int(sc.iteration), int(sc.status), fit.copy())
        f.close()
        ctx.close()
    p, q = states["plain"], states["rccl"]
    assert p[2] == q[2] == 5 and p[3] == q[3] == 0
    assert np.max(np.abs(p[0] - q[0])) <= 1e-12 and abs(p[1] - q[1]) <= 1e-12 * abs(p[1])
    assert np.max(np.abs(p[4] - q[4])) <= 1e-12 * np.max(np.abs(p[4]))


def test_split_column_sum_exchange_gives_the_same_shard_state():
    """GINGR_OPT_SPLIT_EXCHANGE: pass 1 in two halves of the target tiles, the all-reduce of the first half on the context's second
    stream behind the second half (same communicator, event-ordered).  On the one GPU of the pool the exchange is a one-rank
    communicator and the fitter rank 0 of a two-rank row shard (the other rank's sums are missing: an emulated shard, as in
    bench.py --emulate-world) -- what must hold is that the split run lands on the state of the unsplit run of the same shard up to
    the order of the chunk partials, iteration after iteration, and that the event ordering does not lose a half."""
    import gingr_amd as ga
    from gingr_amd import _native as nat
    from gingr_amd.sharded import ShardedFitter
    rng = np.random.default_rng(11)
    ref = rng.normal(0, 50.0, (9000, 3)).astype(np.float32).astype(np.float64)
    mo = go.build_gaussian_gpmm(ref, 70.0, 50.0, rel_tol=1e-12, max_rank=24)
    target = mo.ref[rng.permutation(mo.M)[:8800]] + rng.normal(0, 0.5, (8800, 3))
    states = {}
    for split in (0, 1):
        ctx = ga.Context(0)
        ctx.set_option(nat.OPT_SPLIT_EXCHANGE, split)
        assert ctx.get_option(nat.OPT_SPLIT_EXCHANGE) == split
        ctx.rccl_init(ctx.rccl_unique_id(), 1, 0)
        f = ShardedFitter(ctx, ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam), target, rank=0, world=2, rccl=True)
        f.set_state(np.zeros(mo.rank), ctx.cpd_initial_sigma2(mo.ref, target))
        per_it = []
        for _ in range(4):
            f.update_cpd(0.1, 1.0, 1)
            ctx.synchronize()
            a, sc, fit = f.get_state()
            st = f.get_cpd_stats()
            per_it.append((a.copy(), float(sc.sigma2), int(sc.status), fit.copy(), st["den"].copy()))
        f.update_cpd(0.1, 1.0, 6)                                     # several iterations enqueued back to back
        ctx.synchronize()
        a, sc, fit = f.get_state()
        per_it.append((a.copy(), float(sc.sigma2), int(sc.status), fit.copy(), f.get_cpd_stats()["den"].copy()))
        states[split] = per_it
        f.close()
        ctx.close()
    for (a0, s0, st0, fit0, den0), (a1, s1, st1, fit1, den1) in zip(states[0], states[1]):
        assert st0 == st1 == 0
        assert np.max(np.abs(den1 - den0) / np.abs(den0)) < 1e-12      # both halves of the column sums arrived, in the right places
        assert abs(s1 - s0) <= 1e-11 * abs(s0) and np.max(np.abs(a1 - a0)) <= 1e-9
        assert np.max(np.abs(fit1 - fit0)) <= 1e-9 * np.max(np.abs(fit0))

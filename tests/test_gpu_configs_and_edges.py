"""GPU tests for the remaining BASELINE.json configs (as parity cases, not bench lines) and for edge cases of the C ABI."""
import ctypes

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def synth_model(M, rank, seed, sigma=70.0, scaling=50.0, spread=50.0):
    rng = np.random.default_rng(seed)
    ref = rng.normal(0, spread, (M, 3)).astype(np.float32).astype(np.float64)
    return go.build_gaussian_gpmm(ref, sigma, scaling, rel_tol=1e-12, max_rank=rank), rng


def to_ga(mo):
    import gingr_amd as ga
    return ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)


# ------------------------------------------------------------------------------------------ config 3: ~15k CPD + posterior
def test_config3_15k_cpd_update_against_oracle(ctx):
    """Armadillo-sized CPD (the mesh itself is a missing blob in the reference: synthetic stand-in), 15k <-> 15k, rank 100:
    two full updates against the oracle (C streaming statistics + numpy GP part)."""
    import gingr_amd as ga
    mo, rng = synth_model(15000, 100, seed=15)
    target = mo.instance(rng.normal(0, 1.0, mo.rank)) @ go.euler_to_rot(0.05, -0.04, 0.03).T + np.array([2.0, -1.0, 0.5])
    target = target[rng.permutation(mo.M)] + rng.normal(0, 0.5, (mo.M, 3))
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(to_ga(mo), target, ga.CpdConfiguration(maxIterations=30, w=0.1))
    st = go.initial_state(mo, co.initial_sigma2(mo.ref + mo.mean, target))
    assert abs(state.general.sigma2 - st.sigma2) < 1e-10 * st.sigma2
    for it in range(2):
        state = algo.update(state)
        st = go.cpd_update(mo, target, st, w=0.1, stats=co.cpd_stats(st.fit, target, st.sigma2, 0.1))
        assert state.general.status == st.status == 0
        assert rel(state.general.fit, st.fit) < 1e-5, (it, rel(state.general.fit, st.fit))
        assert abs(state.general.sigma2 - st.sigma2) < 1e-8 * st.sigma2
    algo.close()


# ------------------------------------------------------------------------------------------ config 4: 100k <-> 100k
def test_config4_100k_properties(ctx):
    """100k <-> 100k (the 8-GPU config; here on one GPU): closed-form properties and sampled oracle rows/columns."""
    rng = np.random.default_rng(4)
    N = M = 100000
    x = rng.normal(0, 50, (N, 3)).astype(np.float32).astype(np.float64)
    y = x[rng.permutation(N)] + rng.normal(0, 2, (M, 3))
    got = ctx.cpd_stats(y, x, 25.0, 0.0)
    assert np.max(np.abs(got["Pt1"] - 1.0)) < 1e-12 and abs(got["Np"] - N) < 1e-6
    cols = rng.choice(N, 16, replace=False)
    rows = rng.choice(M, 16, replace=False)
    assert np.allclose(got["den"][cols], co.cpd_colsum_partial(y, x[cols], 25.0, 0, M), rtol=1e-11)
    P1s, PXs = co.cpd_rowstats_partial(y[rows], x, 25.0, got["den"], 0, len(rows))
    assert np.allclose(got["P1"][rows], P1s, rtol=1e-10) and np.allclose(got["PX"][rows], PXs, rtol=1e-9, atol=1e-12)
    idx, d2, _ = ctx.nn(y[:20000], x)
    si, sd2, _ = co.nn(y[rows % 20000], x)
    assert np.array_equal(idx[rows % 20000], si) and np.array_equal(d2[rows % 20000], sd2)


# ------------------------------------------------------------------------------------------ config 5: independent replicas
def test_config5_replicas_are_independent_and_reproducible():
    """8 MH chains = 8 contexts pinned one per GPU ("replicas only").  Two contexts on one device must not interact, and the
    same inputs must give bit-identical results (fixed-order reductions, no float atomics)."""
    import gingr_amd as ga
    mo, rng = synth_model(800, 30, seed=5)
    target = mo.instance(rng.normal(0, 1.0, mo.rank)) + rng.normal(0, 0.3, (mo.M, 3))
    outs = []
    ctxs = [ga.Context(0), ga.Context(0)]
    algos = [ga.IcpRegistration(c) for c in ctxs]
    cfg = ga.IcpConfiguration(maxIterations=20, initialSigma=10.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    states = [a.createInitialState(to_ga(mo), target, cfg) for a in algos]
    for _ in range(3):                      # interleave the two chains
        states = [a.update(s) for a, s in zip(algos, states)]
    for s in states:
        outs.append((s.general.fit.copy(), s.general.modelParameters.shape.copy(), s.general.sigma2))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    for a in algos:
        a.close()
    for c in ctxs:
        c.close()


# ------------------------------------------------------------------------------------------ ranks and sizes off the tile grid
@pytest.mark.parametrize("M,N,rank", [(1, 1, 1), (3, 2, 1), (17, 5, 3), (130, 77, 17), (500, 450, 130), (260, 300, 64), (400, 380, 120),
                                      (700, 650, 512)])
def test_ragged_sizes_and_ranks(ctx, M, N, rank):
    """Sizes that are not multiples of any tile (1 point, rank 1, rank > 128 -> global-memory Cholesky + wide sweeps)."""
    import gingr_amd as ga
    rng = np.random.default_rng(M * 1000 + N + rank)
    ref = rng.normal(0, 30, (M, 3))
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, min(rank, 3 * M))))
    rank = U.shape[1]
    lam = np.sort(rng.uniform(1.0, 400.0, rank))[::-1].copy()
    mo = go.PDM(ref=ref, mean=rng.normal(0, 0.1, (M, 3)), U=U, lam=lam)
    target = rng.normal(0, 30, (N, 3))
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(to_ga(mo), target, ga.CpdConfiguration(maxIterations=10, w=0.2, initialSigma=400.0))
    st = go.initial_state(mo, 400.0)
    for _ in range(2):
        state = algo.update(state)
        st = go.cpd_update(mo, target, st, w=0.2)
        assert state.general.status == st.status
        if st.status == 0:
            assert rel(state.general.fit, st.fit) < 1e-5
            assert abs(state.general.sigma2 - st.sigma2) < 1e-8 * abs(st.sigma2)
    algo.close()


# ------------------------------------------------------------------------------------------ argument / state errors
def test_bad_arguments_and_call_order(ctx):
    import gingr_amd as ga
    from gingr_amd import _native as nat
    lib = nat.load()
    y = np.zeros((4, 3))
    with pytest.raises(ga.GingrNativeError) as e:
        ctx.cpd_stats(y, np.zeros((0, 3)), 1.0, 0.0)          # empty cloud
    assert e.value.code == nat.ERR_BAD_ARGUMENT
    with pytest.raises(ga.GingrNativeError) as e:
        ctx.cpd_stats(y, y, 1.0, 1.0)                         # w must be < 1 (w/(1-w))
    assert e.value.code == nat.ERR_BAD_ARGUMENT
    with pytest.raises(ga.GingrNativeError):
        ctx.nn(np.zeros((0, 3)), y)
    with pytest.raises(ga.GingrNativeError):
        ctx.gauss_block(y, y, -1.0, 1.0)
    mo, _ = synth_model(50, 6, seed=9)
    with pytest.raises(ga.GingrNativeError) as e:
        ga.DeviceModel(ctx, ga.PointDistributionModel(mo.ref, mo.mean, mo.U, -mo.lam))   # negative variance
    assert e.value.code == nat.ERR_BAD_ARGUMENT
    with pytest.raises(ga.GingrNativeError) as e:
        ga.DeviceModel(ctx, to_ga(mo), 10, 5)                 # empty shard
    assert e.value.code == nat.ERR_BAD_ARGUMENT
    dm = ga.DeviceModel(ctx, to_ga(mo))
    h = ctypes.c_void_p()
    assert lib.gingr_fitter_create(ctx.handle, dm.handle, ctypes.byref(h)) == 0
    p = nat.CpdParams(0.0, 1.0)
    assert lib.gingr_fitter_update_cpd_async(h, ctypes.byref(p), 1) == nat.ERR_STATE       # no target / state yet
    assert b"target" in lib.gingr_last_error(ctx.handle)
    lib.gingr_fitter_destroy(h)
    sharded = ga.DeviceModel(ctx, to_ga(mo), 0, 25)           # a row shard must be finalized before use
    assert lib.gingr_fitter_create(ctx.handle, sharded.handle, ctypes.byref(h)) == nat.ERR_STATE
    sharded.close()
    dm.close()


def test_nan_input_is_reported_not_hidden(ctx):
    """NaN coordinates: the reference's posterior would throw inside Try -> the update must report failure, never a
    silently 'repaired' result."""
    import gingr_amd as ga
    mo, rng = synth_model(120, 8, seed=11)
    target = mo.ref + rng.normal(0, 0.3, (mo.M, 3))
    target[7, 1] = np.nan
    algo = ga.CpdRegistration(ctx)
    s0 = algo.createInitialState(to_ga(mo), target, ga.CpdConfiguration(maxIterations=5, initialSigma=50.0))
    s1 = algo.update(s0)
    assert s1.general.status == ga.FittingStatuses.None_ and np.array_equal(s1.general.modelParameters.shape, np.zeros(mo.rank))
    s2 = algo.update(s1)
    assert s2.general.status == ga.FittingStatuses.ModelFlexibilityError
    algo.close()


# ------------------------------------------------------------------------------------------ exact-zero tile culling
@pytest.mark.parametrize("sigma2,w", [(1.0, 0.1), (0.25, 0.3), (30.0, 0.0)])
def test_tile_culling_is_bit_identical(sigma2, w):
    """Model rows and targets live in Morton order on the device and tile pairs whose every K underflows to exactly +0 are
    skipped.  Skipping must not change a single bit: compare against a context with GINGR_OPT_CULL = 0."""
    import gingr_amd as ga
    from gingr_amd import _native as nat
    mo, rng = synth_model(6000, 24, seed=77, spread=60.0)
    target = (mo.ref + rng.normal(0, 0.4, mo.ref.shape))[rng.permutation(mo.M)[:5500]]
    results = []
    # culling on with the kernel variant picked by the device-reported regime (possibly stale), culling off, and culling on with
    # each of the two variants pinned (tile-level only / quarter-tile x slot): all four must agree bit for bit
    for opt, val in ((nat.OPT_CULL, 1), (nat.OPT_CULL, 0), (nat.OPT_FINE_CULL, 0), (nat.OPT_FINE_CULL, 1)):
        c = ga.Context(0)
        c.set_option(opt, val)
        assert c.get_option(opt) == val
        algo = ga.CpdRegistration(c)
        state = algo.createInitialState(to_ga(mo), target, ga.CpdConfiguration(maxIterations=10, w=w, initialSigma=sigma2))
        for _ in range(3):
            state = algo.update(state)
        results.append((state.general.fit.copy(), state.general.modelParameters.shape.copy(), state.general.sigma2,
                        state.general.status))
        algo.close()
        c.close()
    a = results[0]
    for b in results[1:]:
        assert a[3] == b[3]
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)
        assert (a[2] == b[2]) or (np.isnan(a[2]) and np.isnan(b[2]))
    # and the culled run still matches the oracle
    st = go.initial_state(mo, sigma2)
    for _ in range(3):
        st = go.cpd_update(mo, target, st, w=w, stats=co.cpd_stats(st.fit, target, st.sigma2, w))
    assert st.status == a[3]
    if st.status == 0:
        assert rel(a[0], st.fit) < 1e-5


def test_culled_underflow_column_still_fails_like_the_reference(ctx):
    """w = 0 and a target far from every fit point: den_j = 0, 0/0 = NaN in the reference.  The far tile must not be culled
    away (that would silently 'repair' the failure)."""
    import gingr_amd as ga
    mo, rng = synth_model(3000, 12, seed=78, spread=40.0)
    far = np.array([[1e4, 1e4, 1e4]]) + rng.normal(0, 1.0, (300, 3))
    target = np.concatenate([mo.ref + rng.normal(0, 0.2, mo.ref.shape), far])
    algo = ga.CpdRegistration(ctx)
    s0 = algo.createInitialState(to_ga(mo), target, ga.CpdConfiguration(maxIterations=5, w=0.0, initialSigma=1.0))
    s1 = algo.update(s0)
    s2 = algo.update(s1)
    assert s1.general.status == ga.FittingStatuses.None_ and s2.general.status == ga.FittingStatuses.ModelFlexibilityError
    algo.close()


def test_long_run_parity_does_not_drift(ctx):
    """Forty CPD iterations from the initial sigma2 down to the noise floor (guarded fast form early, exact form and tile
    culling late, similarity transform): the trajectories of the HIP path and of the oracle stay together."""
    import gingr_amd as ga
    mo, rng = synth_model(420, 36, 77)
    target = 1.05 * (mo.instance(rng.normal(0, 1.0, mo.rank)) @ go.euler_to_rot(0.08, -0.05, 0.06).T) + np.array([3.0, -2.0, 1.0])
    target = target[rng.permutation(mo.M)[:390]] + rng.normal(0, 0.2, (390, 3))
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
    algo = ga.CpdRegistration(ctx)
    cfg = ga.CpdConfiguration(maxIterations=100, w=0.05)
    state = algo.createInitialState(model, target, cfg, transform=ga.GlobalTranformationType.SimilarityTransforms)
    st = go.initial_state(mo, state.general.sigma2, global_transformation=go.SIMILARITY_TRANSFORMS)
    s2_0 = st.sigma2
    for it in range(40):
        state = algo.update(state)
        st = go.cpd_update(mo, target, st, w=0.05)
        assert state.general.status == st.status == 0, it
    assert st.sigma2 < 1e-3 * s2_0                      # the run really went through the regimes
    assert abs(state.general.sigma2 - st.sigma2) < 1e-6 * st.sigma2
    assert rel(state.general.fit, st.fit) < 1e-6
    assert rel(state.general.modelParameters.shape, st.alpha) < 1e-5
    algo.close()


@pytest.mark.parametrize("rank", [1, 120, 512])
def test_point_cloud_icp_at_the_rank_limits(ctx, rank):
    """The uniform-weight posterior of point-cloud ICP comes from the model's eigen-decomposition of Q^T Q (cyclic Jacobi, at most
    512 x 512): ranks 1, 120 (padded to 128) and 512 against the oracle's Cholesky-free numpy regression."""
    import gingr_amd as ga
    rng = np.random.default_rng(rank)
    M = 600
    ref = rng.normal(0, 30, (M, 3))
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, rank)))
    lam = np.sort(rng.uniform(1.0, 400.0, rank))[::-1].copy()
    mo = go.PDM(ref=ref, mean=rng.normal(0, 0.1, (M, 3)), U=U, lam=lam)
    target = mo.instance(rng.normal(0, 0.5, rank)) + rng.normal(0, 0.2, (M, 3))
    algo = ga.IcpRegistration(ctx)
    cfg = ga.IcpConfiguration(maxIterations=10, initialSigma=20.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    state = algo.createInitialState(to_ga(mo), target, cfg)
    st = go.initial_state(mo, state.general.sigma2)
    for it in range(3):
        state = algo.update(state)
        st, _ = go.icp_update(mo, target, st, 20.0, 1.0, 10)
        assert state.general.status == st.status == 0
        assert rel(state.general.fit, st.fit) < 1e-6 and state.general.sigma2 == st.sigma2, (rank, it)
    algo.close()


@pytest.mark.parametrize("M,N", [(3000, 3000), (5600, 5600)])
def test_resident_cpd_run_in_blocks_equals_the_per_iteration_loop(ctx, M, N):
    """`run` without a call-back enqueues CPD updates in blocks (4 / 2 / 1 by problem size) under the device-side stopping rule
    (gingr_fitter_set_stop_threshold); these sizes take the 2-update and the 1-update block.  Same stopping iteration, same state and
    status as the per-iteration loop with a call-back -- whether the run converges inside a block or runs out of iterations."""
    import gingr_amd as ga
    mo, rng = synth_model(M, 16, seed=M + 5, spread=40.0)
    target = (mo.ref + rng.normal(0, 0.5, mo.ref.shape))[rng.permutation(M)[:N]]
    algo = ga.CpdRegistration(ctx)
    for cfg in (ga.CpdConfiguration(maxIterations=40, threshold=5e-3, w=0.05), ga.CpdConfiguration(maxIterations=6, threshold=1e-12, w=0.0)):
        init = algo.createInitialState(to_ga(mo), target, cfg)
        fast = algo.run(init)
        seen = []
        slow = algo.run(init, callBackLogger=lambda s: seen.append(s.general.iteration))
        assert fast.general.iteration == slow.general.iteration == seen[-1] and fast.general.status == slow.general.status
        assert fast.general.sigma2 == slow.general.sigma2
        assert np.array_equal(fast.general.modelParameters.shape, slow.general.modelParameters.shape)
        assert np.array_equal(fast.general.fit, slow.general.fit)
    algo.close()

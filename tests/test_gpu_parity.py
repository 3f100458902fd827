"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): nearest-neighbour indices bit-exact; posterior-mean vertex positions within 1e-5
relative.  The streaming statistics are held to a much tighter 1e-10 so that the 1e-5 budget is left for the
ill-conditioned solves.
"""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu

REL_STATS = 1e-10   # streaming sums vs the dense float64 restatement
REL_MESH = 1e-5     # north_star tolerance on vertex positions


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def maxrel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def clouds(M, N, seed, noise=2.0, spread=50.0):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, spread, (N, 3)).astype(np.float32).astype(np.float64)
    if M <= N:
        y = x[rng.permutation(N)[:M]] + rng.normal(0, noise, (M, 3))
    else:
        y = rng.normal(0, spread, (M, 3))
    return y, x


# ------------------------------------------------------------------------------------------ all-pairs operators
@pytest.mark.parametrize("M,N,sigma2,w", [
    (1, 1, 3.0, 0.0), (2, 3, 10.0, 0.0), (257, 255, 5000.0, 0.0), (300, 513, 4.0, 0.1), (1000, 777, 40.0, 0.3),
    (64, 2000, 900.0, 0.0),
])
def test_cpd_stats_vs_dense_oracle(ctx, M, N, sigma2, w):
    y, x = clouds(M, N, seed=M * 7 + N)
    got = ctx.cpd_stats(y, x, sigma2, w)
    want = go.cpd_stats_dense(y, x, sigma2, w)
    assert maxrel(got["den"], want.den) < REL_STATS
    assert maxrel(got["P1"], want.P1) < REL_STATS
    assert rel(got["PX"], want.PX) < REL_STATS
    assert maxrel(got["Pt1"], want.Pt1) < REL_STATS
    assert abs(got["Np"] - want.Np) <= REL_STATS * abs(want.Np)
    assert abs(got["sigma2_next"] - want.sigma2_next) <= 1e-9 * abs(want.sigma2_next)


def test_cpd_stats_w0_closed_form(ctx):
    # w = 0: every column of P sums to one, Np = N
    y, x = clouds(500, 400, seed=3)
    got = ctx.cpd_stats(y, x, 200.0, 0.0)
    assert np.allclose(got["Pt1"], 1.0, rtol=0, atol=1e-12)
    assert abs(got["Np"] - 400.0) < 1e-9
    assert got["c"] == 0.0


def test_cpd_stats_underflow_gives_nan_like_reference(ctx):
    # a target point ~40 sigma away from every fit point: den_j underflows to 0, 0/0 = NaN (CPD.scala:66,71-74)
    y = np.zeros((4, 3)); y[:, 0] = np.arange(4)
    x = np.array([[0.5, 0, 0], [1000.0, 0, 0]])
    got = ctx.cpd_stats(y, x, 1.0, 0.0)
    want = go.cpd_stats_dense(y, x, 1.0, 0.0)
    assert want.den[1] == 0.0 and got["den"][1] == 0.0
    assert np.all(np.isnan(got["P1"])) and np.all(np.isnan(want.P1))


def test_cpd_stats_medium_vs_c_oracle(ctx):
    y, x = clouds(5000, 5000, seed=11)
    for sigma2, w in [(5000.0, 0.0), (4.0, 0.1)]:
        got = ctx.cpd_stats(y, x, sigma2, w)
        want = co.cpd_stats(y, x, sigma2, w)
        assert maxrel(got["den"], want.den) < REL_STATS
        assert maxrel(got["P1"], want.P1) < REL_STATS
        assert rel(got["PX"], want.PX) < REL_STATS
        assert abs(got["sigma2_next"] - want.sigma2_next) <= 1e-9 * abs(want.sigma2_next)


def test_initial_sigma2(ctx):
    y, x = clouds(700, 900, seed=5)
    assert abs(ctx.cpd_initial_sigma2(y, x) - go.cpd_initial_sigma2(y, x)) < 1e-11 * go.cpd_initial_sigma2(y, x)


@pytest.mark.parametrize("offset", [0.0, 1e4])
def test_initial_sigma2_from_moments(ctx, offset):
    """From 2^20 pairs on the sum over all pairs comes from the clouds' moments about the first point (affinity.hip
    cloud_moments_kernel); clouds 10^4 away from the origin (200 times their extent) must not cost digits."""
    y, x = clouds(2100, 3000, seed=6)
    y, x = y + offset, x + offset + 3.0
    want = go.cpd_initial_sigma2(y, x)
    assert abs(ctx.cpd_initial_sigma2(y, x) - want) < 1e-11 * want


@pytest.mark.parametrize("M,N", [(1, 1), (5, 1), (300, 257), (2000, 5000), (5000, 333)])
def test_nn_bit_exact(ctx, M, N):
    rng = np.random.default_rng(M + 13 * N)
    x = rng.normal(0, 50, (N, 3))
    y = rng.normal(0, 50, (M, 3))
    idx, d2, md = ctx.nn(y, x)
    widx, wd2, wmd = co.nn(y, x)
    assert np.array_equal(idx, widx)
    assert np.array_equal(d2, wd2)          # squared distances are bit-identical too
    assert abs(md - wmd) <= 1e-14 * max(wmd, 1.0)


def test_nn_ties_lowest_index(ctx):
    # duplicated target points and an integer grid: exact ties everywhere, lowest index must win
    g = np.stack(np.meshgrid(np.arange(6.0), np.arange(6.0), np.arange(6.0), indexing="ij"), -1).reshape(-1, 3)
    x = np.concatenate([g, g, g[::-1]])
    y = g + 0.5
    idx, _, _ = ctx.nn(y, x)
    widx, _, _ = go.icp_closest_point(y, x)
    assert np.array_equal(idx, widx)
    assert idx.max() < g.shape[0]


def test_gauss_block(ctx):
    rng = np.random.default_rng(2)
    A, B = rng.normal(0, 40, (300, 3)), rng.normal(0, 40, (77, 3))
    got = ctx.gauss_block(A, B, 70.0, 50.0)
    want = go.gauss_block(A, B, 70.0, 50.0)
    assert np.max(np.abs(got - want)) < 1e-12 * 50.0
    assert got.shape == (300, 77)


# ------------------------------------------------------------------------------------------ model operators
def make_model(M=600, rank=40, seed=1, sigma=60.0, scaling=30.0):
    rng = np.random.default_rng(seed)
    ref = rng.normal(0, 40, (M, 3))
    mo = go.build_gaussian_gpmm(ref, sigma=sigma, scaling=scaling, rel_tol=1e-6, max_rank=rank)
    mo.mean = rng.normal(0, 0.3, (M, 3))        # non-zero mean displacement
    return mo


def to_ga(mo):
    import gingr_amd as ga
    return ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)


def test_model_instance_and_coefficients(ctx):
    import gingr_amd as ga
    mo = make_model()
    dm = ga.DeviceModel(ctx, to_ga(mo))
    rng = np.random.default_rng(4)
    alpha = rng.normal(0, 1, mo.rank)
    euler, t, c, s = (0.3, -0.2, 0.1), (5.0, -3.0, 2.0), (1.0, 2.0, 3.0), 1.1
    st = go.State(alpha=alpha, euler=euler, center=np.array(c), translation=np.array(t), scale=s, sigma2=1.0,
                  fit=np.zeros((mo.M, 3)))
    want = go.model_instance_shape_pose_scale(mo, st)
    got = dm.instance(alpha, euler, c, t, s)
    assert rel(got, want) < 1e-13
    posed = mo.transform(go.euler_to_rot(*euler), np.array(t), np.array(c))
    mesh = posed.instance(alpha) + rng.normal(0, 0.05, (mo.M, 3))
    want_a = posed.coefficients(mesh)
    got_a = dm.coefficients(mesh, euler, c, t)
    assert rel(got_a, want_a) < 1e-8
    dm.close()


def test_model_posterior_mean_with_landmarks(ctx):
    import gingr_amd as ga
    mo = make_model(M=500, rank=33, seed=9)
    dm = ga.DeviceModel(ctx, to_ga(mo))
    rng = np.random.default_rng(8)
    euler, t, c = (0.1, 0.2, -0.3), np.array([1.0, 2.0, -1.0]), np.zeros(3)
    posed = mo.transform(go.euler_to_rot(*euler), t, c)
    obs = posed.instance(rng.normal(0, 1, mo.rank)) + rng.normal(0, 0.2, (mo.M, 3))
    var = rng.uniform(0.5, 4.0, mo.M)
    lm_p = np.array([3, 77, 250])
    A = rng.normal(0, 1, (3, 3, 3))
    lm_cov = np.einsum("lab,lcb->lac", A, A) + 0.5 * np.eye(3)
    lm_xyz = obs[lm_p] + 1.0
    # oracle: landmark pids replace the dense observation of the same point
    keep = ~np.isin(np.arange(mo.M), lm_p)
    pids = np.concatenate([np.arange(mo.M)[keep], lm_p])
    pts = np.concatenate([obs[keep], lm_xyz])
    covs = np.concatenate([var[keep][:, None, None] * np.eye(3)[None], lm_cov])
    want_mesh, want_a = posed.posterior_mean(pids, pts, covs)
    got_mesh, got_a = dm.posterior_mean(obs, 1.0 / var, euler, c, t,
                                        ga.LandmarkCorrespondences(lm_p, lm_xyz, lm_cov))
    assert rel(got_mesh, want_mesh) < REL_MESH * 1e-3
    assert rel(got_a, want_a) < 1e-7
    dm.close()


# ------------------------------------------------------------------------------------------ the update
def run_pair(ctx, mo, target, algo_name, n_iter, transform, cfg_kwargs, landmarks=None, step=1.0):
    import gingr_amd as ga
    model = to_ga(mo)
    lm_ga = None if landmarks is None else ga.LandmarkCorrespondences(landmarks.pids, landmarks.points, landmarks.covs)
    if algo_name == "cpd":
        algo = ga.CpdRegistration(ctx)
        cfg = ga.CpdConfiguration(**cfg_kwargs)
    else:
        algo = ga.IcpRegistration(ctx)
        cfg = ga.IcpConfiguration(**{"correspondenceMethod": "PointcloudClosestPoint", **cfg_kwargs})
    state = algo.createInitialState(model, target, cfg, transform=transform, stepLength=step, landmarks=lm_ga)
    st = go.initial_state(mo, state.general.sigma2, global_transformation=transform, step_length=step)
    assert rel(state.general.fit, st.fit) < 1e-13
    traj = []
    for _ in range(n_iter):
        state = algo.update(state)
        if algo_name == "cpd":
            st = go.cpd_update(mo, target, st, w=cfg.w, lam=cfg.lambda_, landmarks=landmarks)
        else:
            st, idx = go.icp_update(mo, target, st, cfg.initialSigma, cfg.endSigma, cfg.maxIterations, landmarks=landmarks)
            assert np.array_equal(algo.last_correspondence_indices(), idx)
        traj.append((state, st))
    algo.close()
    return traj


@pytest.mark.parametrize("transform", [0, 1, 2])
def test_cpd_update_trajectory(ctx, transform):
    mo = make_model(M=700, rank=48, seed=21)
    rng = np.random.default_rng(22)
    R = go.euler_to_rot(0.15, -0.1, 0.2)
    target = (mo.instance(rng.normal(0, 1.0, mo.rank)) @ R.T) * (1.05 if transform == 2 else 1.0) + np.array([4.0, -2.0, 3.0])
    target = target[rng.permutation(mo.M)[:650]] + rng.normal(0, 0.3, (650, 3))
    traj = run_pair(ctx, mo, target, "cpd", 5, transform, dict(maxIterations=20, w=0.05))
    for k, (state, st) in enumerate(traj):
        g = state.general
        assert g.status == st.status == 0, k
        assert g.iteration == st.iteration
        assert rel(g.fit, st.fit) < REL_MESH, (k, rel(g.fit, st.fit))
        assert abs(g.sigma2 - st.sigma2) < 1e-7 * abs(st.sigma2), k
        assert rel(g.modelParameters.shape, st.alpha) < 1e-4, k
        assert np.allclose([g.modelParameters.rotation.phi, g.modelParameters.rotation.theta,
                            g.modelParameters.rotation.psi], st.euler, atol=1e-8), k
        assert np.allclose(g.modelParameters.translation, st.translation, atol=1e-6), k
        assert abs(g.modelParameters.scale - st.scale) < 1e-9, k


def test_cpd_update_with_landmarks_and_step(ctx):
    mo = make_model(M=400, rank=30, seed=31)
    rng = np.random.default_rng(32)
    target = mo.instance(rng.normal(0, 1.0, mo.rank)) + rng.normal(0, 0.2, (mo.M, 3)) + 2.0
    lm_idx = np.array([5, 100, 399])
    A = rng.normal(0, 1, (3, 3, 3))
    covs = np.einsum("lab,lcb->lac", A, A) + np.eye(3)
    lms = go.Landmarks(pids=lm_idx, points=target[lm_idx], covs=covs)
    traj = run_pair(ctx, mo, target, "cpd", 3, 1, dict(maxIterations=20, initialSigma=30.0, w=0.0, lambda_=2.0),
                    landmarks=lms, step=0.5)
    for k, (state, st) in enumerate(traj):
        assert state.general.status == st.status == 0
        assert rel(state.general.fit, st.fit) < REL_MESH, (k, rel(state.general.fit, st.fit))
        assert abs(state.general.sigma2 - st.sigma2) < 1e-7 * abs(st.sigma2)


def test_icp_update_trajectory(ctx):
    mo = make_model(M=500, rank=36, seed=41)
    rng = np.random.default_rng(42)
    target = mo.instance(rng.normal(0, 0.8, mo.rank)) + rng.normal(0, 0.1, (mo.M, 3))
    target = np.concatenate([target, rng.normal(0, 40, (200, 3))])
    traj = run_pair(ctx, mo, target, "icp", 4, 1, dict(maxIterations=10, initialSigma=20.0, endSigma=1.0))
    for k, (state, st) in enumerate(traj):
        assert state.general.status == st.status == 0
        assert rel(state.general.fit, st.fit) < REL_MESH, (k, rel(state.general.fit, st.fit))
        assert state.general.sigma2 == st.sigma2


@pytest.mark.parametrize("rank,sigma,duplicate", [(36, 60.0, True), (130, 22.0, False), (170, 18.0, False)])
def test_icp_update_through_the_moment_eigenbasis(ctx, rank, sigma, duplicate):
    """Point-cloud ICP without landmarks takes its posterior from the eigenbasis of the model's moment Q^T Q (eig.hip at finalize):
    ranks 130 / 170 are the two- and three-values-per-lane kernels; two identical basis columns make the moment singular, the
    Cholesky in front of the one-sided Jacobi fails and the two-sided kernel (complete V) serves."""
    mo = make_model(M=500, rank=rank, seed=43, sigma=sigma, scaling=20.0)
    assert mo.rank == rank
    if duplicate:
        mo.U[:, 5] = mo.U[:, 4]
    rng = np.random.default_rng(44)
    target = mo.instance(rng.normal(0, 0.8, mo.rank)) + rng.normal(0, 0.1, (mo.M, 3))
    traj = run_pair(ctx, mo, target, "icp", 3, 1, dict(maxIterations=10, initialSigma=20.0, endSigma=1.0))
    for k, (state, st) in enumerate(traj):
        assert state.general.status == st.status == 0
        assert rel(state.general.fit, st.fit) < REL_MESH, (k, rel(state.general.fit, st.fit))


def test_model_flexibility_error_status(ctx):
    # sigma2 tiny + w = 0 + a far-away target point: den underflows -> NaN -> posterior fails.
    # iteration 0: state returned unchanged; iteration > 0: ModelFlexibilityError (GingrAlgorithm.scala:194-208)
    import gingr_amd as ga
    mo = make_model(M=200, rank=12, seed=51)
    target = np.concatenate([mo.ref + mo.mean, [[5000.0, 0, 0]]])
    algo = ga.CpdRegistration(ctx)
    cfg = ga.CpdConfiguration(maxIterations=10, initialSigma=1.0, w=0.0)
    s0 = algo.createInitialState(to_ga(mo), target, cfg)
    s1 = algo.update(s0)
    assert s1.general.status == ga.FittingStatuses.None_ and s1.general.iteration == 1
    assert np.array_equal(s1.general.modelParameters.shape, s0.general.modelParameters.shape)
    assert s1.general.sigma2 == s0.general.sigma2
    s2 = algo.update(s1)
    assert s2.general.status == ga.FittingStatuses.ModelFlexibilityError
    st = go.initial_state(mo, 1.0)
    st = go.cpd_update(mo, target, st)
    st = go.cpd_update(mo, target, st)
    assert st.status == go.STATUS_MODEL_FLEXIBILITY_ERROR
    # run(): the failed iteration-0 update leaves sigma2 unchanged, so the CPD convergence test |d sigma2| < threshold
    # (CPD.scala:108-110) fires on the next element of the chain -- the reference reports "converged" here too
    final = algo.run(s0)
    assert final.general.status == ga.FittingStatuses.Converged and final.general.iteration == 1
    # started from a state that is past iteration 0 the same failure is a ModelFlexibilityError and stops the run
    final2 = algo.run(s1)
    assert final2.general.status == ga.FittingStatuses.ModelFlexibilityError
    algo.close()


def test_run_converges_and_matches_oracle_loop(ctx):
    import gingr_amd as ga
    mo = make_model(M=300, rank=20, seed=61)
    rng = np.random.default_rng(62)
    target = mo.instance(rng.normal(0, 1.0, mo.rank)) + rng.normal(0, 0.2, (mo.M, 3))
    algo = ga.CpdRegistration(ctx)
    cfg = ga.CpdConfiguration(maxIterations=8, w=0.0)
    s0 = algo.createInitialState(to_ga(mo), target, cfg, transform=0)
    final = algo.run(s0)
    # take(maxIterations) on a chain that yields the initial state first = maxIterations - 1 updates
    assert final.general.iteration == cfg.maxIterations - 1
    assert final.general.status == ga.FittingStatuses.MaxIteration
    st = go.initial_state(mo, s0.general.sigma2, global_transformation=0)
    for _ in range(cfg.maxIterations - 1):
        st = go.cpd_update(mo, target, st)
    assert rel(final.general.fit, st.fit) < REL_MESH
    algo.close()


# ------------------------------------------------------------------------------------------ weighted Gram kernel, every tile count
@pytest.mark.parametrize("rank,M", [(5, 37), (16, 129), (20, 700), (33, 1030), (48, 2049), (50, 333), (72, 1500), (80, 1111), (90, 2500),
                                    (96, 515), (100, 4099), (112, 3001)])
def test_weighted_gram_every_tile_count(ctx, rank, M):
    """gram_tri_kernel<NT, FULL> for NT = 1..7 (rank padded to 16 NT), row counts off the 16-row step grid and off the slab grid:
    (a) the weight-only launch behind posterior_mean (FULL = false), (b) the fused weight + right-hand-side launch of a CPD update
    (FULL = true), each against the oracle."""
    import gingr_amd as ga
    rng = np.random.default_rng(rank + M)
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, rank)))
    lam = np.sort(rng.uniform(1.0, 400.0, rank))[::-1].copy()
    mo = go.PDM(ref=rng.normal(0, 30, (M, 3)), mean=rng.normal(0, 0.1, (M, 3)), U=U, lam=lam)
    dm = ga.DeviceModel(ctx, to_ga(mo))
    obs = mo.instance(rng.normal(0, 1, mo.rank)) + rng.normal(0, 0.3, (mo.M, 3))
    var = rng.uniform(0.5, 4.0, mo.M)
    var[rng.integers(0, mo.M, max(1, mo.M // 50))] = 1e12     # a few rows with (nearly) zero weight
    want_mesh, want_a = mo.posterior_mean(np.arange(mo.M), obs, var[:, None, None] * np.eye(3)[None])
    got_mesh, got_a = dm.posterior_mean(obs, 1.0 / var)
    assert rel(got_a, want_a) < 1e-7
    assert rel(got_mesh, want_mesh) < 1e-8
    dm.close()
    target = mo.instance(rng.normal(0, 0.7, mo.rank)) + rng.normal(0, 0.5, (mo.M, 3))
    target = target[rng.permutation(mo.M)[: max(1, (3 * mo.M) // 4)]]
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(to_ga(mo), target, ga.CpdConfiguration(maxIterations=10, w=0.1, initialSigma=50.0))
    st = go.initial_state(mo, 50.0)
    for _ in range(2):
        state = algo.update(state)
        st = go.cpd_update(mo, target, st, w=0.1)
        assert state.general.status == st.status == 0
        assert rel(state.general.fit, st.fit) < 1e-7
        assert abs(state.general.sigma2 - st.sigma2) < 1e-8 * abs(st.sigma2)
    algo.close()

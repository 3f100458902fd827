"""Model ranks above 112: the wide Gram pass (gp_wide.hip: eight waves share the triangle; several
workgroups per slab past 136 tiles), the blocked posterior solve on the global workspace and the box-writing fit pass at every
padded rank class rp = 128 .. 512 -- against the numpy oracle of the GP part (scalismo's regression, reached from
G/api/GingrAlgorithm.scala:297-301) and, for whole updates, against the oracle's update map (GingrAlgorithm.scala:192-254).

Tolerances: posterior-mean vertices <= 1e-5 relative (BASELINE.json north_star); coefficients <= 1e-6 here because the systems
are well conditioned by construction.  The oracle is a restatement (parity unpinned)."""
import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def _model(M, rank, seed):
    rng = np.random.default_rng(seed)
    ref = rng.normal(0, 30, (M, 3))
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, rank)))
    lam = np.sort(rng.uniform(1.0, 400.0, rank))[::-1].copy()
    return go.PDM(ref=ref, mean=rng.normal(0, 0.1, (M, 3)), U=U, lam=lam), rng


def _to_ga(mo):
    import gingr_amd as ga
    return ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)


# every T / SUB / parts class of gp_wide.hip's plan: rp = 128 (SUB 4), 144 .. 256 (SUB 2, T = 6 .. 17), 272 .. 512 (two to four parts)
WIDE_RANKS = [113, 128, 129, 150, 176, 190, 200, 224, 240, 256, 257, 300, 336, 352, 384, 400, 448, 470, 496, 512]


@pytest.mark.parametrize("rank", WIDE_RANKS)
def test_posterior_mean_at_wide_ranks(ctx, rank):
    """Weighted Gram + right-hand side + solve + posed instance: one stateless posterior mean with ragged weights (zeros, a wide
    range) and a pose, against the oracle.  M is not a multiple of any step length, so the slack rows behind the basis are read."""
    import gingr_amd as ga
    M = 601 + (rank % 7)
    mo, rng = _model(M, rank, seed=rank)
    dm = ga.DeviceModel(ctx, _to_ga(mo))
    euler, t = (0.05, -0.03, 0.02), np.array([1.0, -2.0, 0.5])
    R = go.euler_to_rot(*euler)
    obs = (mo.ref + mo.mean) @ R.T + t + rng.normal(0, 1.0, (M, 3))
    w = rng.uniform(0.01, 5.0, M)
    w[rng.random(M) < 0.15] = 0.0
    mean, coeffs = dm.posterior_mean(obs, w, euler=euler, translation=tuple(t))
    # oracle: rows with weight 0 are not observed
    st = go.State(alpha=np.zeros(rank), euler=euler, center=np.zeros(3), translation=t, scale=1.0, sigma2=1.0, fit=mo.ref + mo.mean,
                  iteration=0, status=0, global_transformation=go.RIGID_TRANSFORMS, step_length=1.0)
    keep = np.nonzero(w > 0)[0]
    mean_ref, a_ref, _ = go.compute_posterior_mean(mo, st, keep, obs[keep], 1.0 / w[keep], None)
    assert rel(coeffs, a_ref) < 1e-6, (rank, rel(coeffs, a_ref))
    assert rel(mean, mean_ref) < 1e-8, (rank, rel(mean, mean_ref))
    dm.close()


@pytest.mark.parametrize("M,N,rank", [(700, 650, 128), (800, 700, 200), (900, 850, 256), (1000, 900, 330), (1200, 1100, 512),
                                      (48, 40, 130), (171, 160, 512)])    # (the last two: barely more basis rows than columns)
def test_cpd_updates_at_wide_ranks(ctx, M, N, rank):
    """Three fused CPD updates (fit pass with quarter boxes at rp > 128 included: the second update's culling reads them)."""
    import gingr_amd as ga
    mo, rng = _model(M, rank, seed=M + rank)
    target = rng.normal(0, 30, (N, 3))
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(_to_ga(mo), target, ga.CpdConfiguration(maxIterations=10, w=0.2, initialSigma=400.0))
    st = go.initial_state(mo, 400.0)
    for _ in range(3):
        state = algo.update(state)
        st = go.cpd_update(mo, target, st, w=0.2)
        assert state.general.status == st.status == 0
        assert rel(state.general.fit, st.fit) < 1e-5
        assert abs(state.general.sigma2 - st.sigma2) < 1e-8 * abs(st.sigma2)
        assert rel(state.general.modelParameters.shape, st.alpha) < 1e-4
    algo.close()


@pytest.mark.parametrize("rank", [200, 256])
def test_sampled_proposal_at_wide_ranks(ctx, rank):
    """update(probabilistic = true) at r > 128: a + L^-T z from the blocked solve equals the oracle's draw with the same z."""
    import gingr_amd as ga
    M, N = 500, 480
    mo, rng = _model(M, rank, seed=7 * rank)
    target = rng.normal(0, 30, (N, 3))
    algo = ga.CpdRegistration(ctx)
    state = algo.createInitialState(_to_ga(mo), target, ga.CpdConfiguration(maxIterations=10, w=0.1, initialSigma=300.0))
    st = go.initial_state(mo, 300.0)
    z = np.random.default_rng(5).standard_normal(rank)

    class _Fixed:
        def standard_normal(self, n):
            return z[:n].copy()

    state = algo.update(state, probabilistic=True, rnd=_Fixed())
    st = go.cpd_update(mo, target, st, w=0.1, z=z)
    assert state.general.status == st.status == 0
    assert rel(state.general.fit, st.fit) < 1e-5
    assert rel(state.general.modelParameters.shape, st.alpha) < 1e-4
    algo.close()


@pytest.mark.parametrize("rank", [128, 150, 256, 300, 512])
def test_log_transition_density_at_wide_ranks(ctx, rank):
    """posterior.gp.logpdf(posterior.coefficients(mesh)) (G/api/sampling/generators/GeneratorWrapperStochastic.scala:42-63) at r >= 128:
    the two factorisations side by side on the global workspaces (posterior_logpdf_wide_kernel), then the cached form for a second
    query about the same state -- both against the oracle; a state whose posterior fails reports -inf."""
    import gingr_amd as ga
    M = 450
    mo, rng = _model(M, rank, seed=11 * rank)
    target = mo.instance(rng.normal(0, 1.0, rank)) + rng.normal(0, 0.3, (M, 3))
    algo = ga.CpdRegistration(ctx)
    cfg = ga.CpdConfiguration(maxIterations=50, w=0.05)
    s0 = algo.createInitialState(_to_ga(mo), target, cfg)
    s1 = algo.update(s0)
    st = go.cpd_update(mo, target, go.initial_state(mo, s0.general.sigma2), w=0.05)
    assert rel(s1.general.fit, st.fit) < 1e-6
    s2 = algo.update(s1, probabilistic=True, rnd=np.random.default_rng(5))
    want = go.posterior_logpdf_of_mesh(mo, st, *go.cpd_observations(mo, target, st, w=0.05), mesh=st.fit)
    got = algo.logTransitionProbability(s1, s2)
    assert np.isfinite(got) and abs(got - want) < 1e-5 * abs(want), (rank, got, want)
    again = algo.logTransitionProbability(s1, s2)       # the memo / cached form answers the second query
    assert abs(again - want) < 1e-5 * abs(want), (rank, again, want)
    algo.close()

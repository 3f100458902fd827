"""GPU tests of the probabilistic proposal: update(probabilistic=True) = posterior.sample(), and the
log transition probability of GeneratorWrapperStochastic, against the oracle's restatement of scalismo's posterior model."""
import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def setup(M=500, rank=28, seed=3, step=1.0, transform=1):
    import gingr_amd as ga
    rng = np.random.default_rng(seed)
    ref = rng.normal(0, 40, (M, 3))
    mo = go.build_gaussian_gpmm(ref, 60.0, 30.0, rel_tol=1e-9, max_rank=rank)
    mo.mean = rng.normal(0, 0.2, (M, 3))
    target = mo.instance(rng.normal(0, 1.0, mo.rank)) @ go.euler_to_rot(0.1, 0.05, -0.08).T + np.array([1.0, 2.0, -1.0])
    nt = min(450, M - M // 10)
    target = target[rng.permutation(M)[:nt]] + rng.normal(0, 0.3, (nt, 3))
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
    return mo, model, target


@pytest.mark.parametrize("algo_name", ["cpd", "icp"])
def test_sample_update_matches_oracle_with_the_same_draws(ctx, algo_name):
    import gingr_amd as ga
    mo, model, target = setup()
    if algo_name == "cpd":
        algo, cfg = ga.CpdRegistration(ctx), ga.CpdConfiguration(maxIterations=50, w=0.1)
    else:
        algo, cfg = ga.IcpRegistration(ctx), ga.IcpConfiguration(maxIterations=50, initialSigma=30.0, endSigma=1.0, correspondenceMethod="PointcloudClosestPoint")
    state = algo.createInitialState(model, target, cfg)
    st = go.initial_state(mo, state.general.sigma2)
    r1, r2 = np.random.default_rng(99), np.random.default_rng(99)
    for it in range(3):
        probabilistic = it != 1                      # mix sampled and mean proposals
        state = algo.update(state, probabilistic=probabilistic, rnd=r1)
        z = r2.standard_normal(mo.rank) if probabilistic else None
        if algo_name == "cpd":
            st = go.cpd_update(mo, target, st, w=0.1, z=z)
        else:
            idx, _, _ = go.icp_closest_point(st.fit, target)
            s2n = go.icp_update_sigma2(st.sigma2, 30.0, 1.0, 50)
            st = go.update_from_observations(mo, st, np.arange(mo.M), target[idx], np.full(mo.M, st.sigma2), s2n, None, z)
        assert state.general.status == st.status == 0
        assert rel(state.general.fit, st.fit) < 1e-5, (it, rel(state.general.fit, st.fit))
        assert rel(state.general.modelParameters.shape, st.alpha) < 1e-4
    # the sampled proposal really differs from the mean proposal
    s_mean = algo.update(state, probabilistic=False)
    s_samp = algo.update(state, probabilistic=True, rnd=np.random.default_rng(1))
    assert rel(s_samp.general.fit, s_mean.general.fit) > 1e-6
    algo.close()


def test_sample_covariance_is_the_posterior_covariance(ctx):
    """Empirical check of the sampling distribution through the C ABI: with a one-step update map the coefficients of the
    sampled proposals scatter with covariance close to (I + G)^-1 (here: their projected shape parameters have the oracle's
    posterior spread); a cheap sanity check that z really enters as L^-T z."""
    import gingr_amd as ga
    mo, model, target = setup(M=300, rank=6, seed=8)
    algo = ga.CpdRegistration(ctx)
    s0 = algo.createInitialState(model, target[:250], ga.CpdConfiguration(maxIterations=50, w=0.1),
                                 transform=ga.GlobalTranformationType.NoTransforms)
    rnd = np.random.default_rng(5)
    samples = np.array([algo.update(s0, probabilistic=True, rnd=rnd).general.modelParameters.shape for _ in range(400)])
    mean_prop = algo.update(s0, probabilistic=False).general.modelParameters.shape
    assert np.allclose(samples.mean(0), mean_prop, atol=4 * samples.std(0).max() / np.sqrt(400) + 1e-6)
    # oracle covariance of alpha' for sampled proposals: alpha' ~ B S (a + L^-T z) / eps  =>  Cov = J Minv J^T
    st = go.initial_state(mo, s0.general.sigma2, global_transformation=go.NO_TRANSFORMS)
    pids, pts, var = go.cpd_observations(mo, target[:250], st, w=0.1)
    Q = mo.U * np.sqrt(mo.lam)
    Mm = np.eye(mo.rank) + Q.T @ (Q / np.repeat(var, 3)[:, None])
    S = Q.T @ Q
    J = np.linalg.solve(S / 1e-5 + np.eye(mo.rank), S / 1e-5)
    cov = J @ np.linalg.inv(Mm) @ J.T
    emp = np.cov(samples.T)
    assert np.allclose(np.sqrt(np.diag(emp)), np.sqrt(np.diag(cov)), rtol=0.2)
    algo.close()


@pytest.mark.parametrize("step", [1.0, 0.5])
def test_log_transition_probability(ctx, step):
    import gingr_amd as ga
    mo, model, target = setup(seed=13)
    algo = ga.CpdRegistration(ctx)
    cfg = ga.CpdConfiguration(maxIterations=50, w=0.05, lambda_=1.5)
    s0 = algo.createInitialState(model, target, cfg, stepLength=step)
    s1 = algo.update(s0)
    s2 = algo.update(s1, probabilistic=True, rnd=np.random.default_rng(2))
    got = algo.logTransitionProbability(s1, s2)
    # oracle: posterior of s1, mesh as in GeneratorWrapperStochastic.scala:48-53
    st = go.initial_state(mo, s0.general.sigma2, step_length=step)
    st = go.cpd_update(mo, target, st, w=0.05, lam=1.5)
    assert rel(s1.general.fit, st.fit) < 1e-6
    if step != 1.0:
        comp = st.alpha + (s2.general.modelParameters.shape - st.alpha) / step
        mesh = mo.instance(comp)
    else:
        mesh = st.fit
    want = go.posterior_logpdf_of_mesh(mo, st, *go.cpd_observations(mo, target, st, w=0.05, lam=1.5), mesh=mesh)
    assert np.isfinite(got) and abs(got - want) < 1e-5 * abs(want), (got, want)
    # querying must not disturb the chain: the next deterministic update is unchanged
    a = algo.update(s1).general.fit
    _ = algo.logTransitionProbability(s1, s2)
    b = algo.update(s1).general.fit
    assert np.array_equal(a, b)
    algo.close()


def test_log_transition_probability_of_failed_posterior_is_minus_infinity(ctx):
    import gingr_amd as ga
    mo, model, _ = setup(M=200, rank=8, seed=21)
    target = np.concatenate([mo.ref + mo.mean, [[5000.0, 0, 0]]])
    algo = ga.CpdRegistration(ctx)
    s0 = algo.createInitialState(model, target, ga.CpdConfiguration(maxIterations=10, initialSigma=1.0, w=0.0))
    assert algo.logTransitionProbability(s0, s0) == float("-inf")
    algo.close()


def test_rank_above_128_uses_the_global_workspace(ctx):
    """r > 128: the bordered solve and the log transition density run on the L2-resident workspace (same code as r <= 128)."""
    import gingr_amd as ga
    rng = np.random.default_rng(31)
    M = 400
    ref = rng.normal(0, 40, (M, 3))
    U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, 150)))
    lam = np.sort(rng.uniform(1.0, 300.0, 150))[::-1].copy()
    mo = go.PDM(ref=ref, mean=rng.normal(0, 0.2, (M, 3)), U=np.ascontiguousarray(U), lam=lam)
    target = mo.instance(rng.normal(0, 1.0, 150)) + rng.normal(0, 0.3, (M, 3))
    model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
    algo = ga.CpdRegistration(ctx)
    cfg = ga.CpdConfiguration(maxIterations=50, w=0.05)
    s0 = algo.createInitialState(model, target, cfg)
    s1 = algo.update(s0)
    st = go.cpd_update(mo, target, go.initial_state(mo, s0.general.sigma2), w=0.05)
    assert rel(s1.general.fit, st.fit) < 1e-6
    z = np.random.default_rng(5).standard_normal(150)
    s2 = algo.update(s1, probabilistic=True, rnd=np.random.default_rng(5))
    st2 = go.cpd_update(mo, target, st, w=0.05, z=z)
    assert rel(s2.general.fit, st2.fit) < 1e-5
    got = algo.logTransitionProbability(s1, s2)
    want = go.posterior_logpdf_of_mesh(mo, st, *go.cpd_observations(mo, target, st, w=0.05), mesh=st.fit)
    assert np.isfinite(got) and abs(got - want) < 1e-5 * abs(want), (got, want)
    algo.close()


def test_retry_counter_of_the_probabilistic_proposal(ctx):
    """GingrAlgorithm.scala:69-70,194-210 on the device: 11 consecutive sampled proposals with a failing posterior -> 10 unchanged
    states, then ModelFlexibilityError; a success gives one retry back; deterministic updates never retry; iteration 0 is
    forgiven.  Same sequence as the oracle (tests/test_oracle_kat.py::test_probabilistic_retry_counter_semantics)."""
    import dataclasses
    import gingr_amd as ga
    mo, model, target = setup(M=200, rank=12)
    bad_target = np.concatenate([mo.ref, [[9.0e5, 0.0, 0.0]]])     # sigma2 = 1: one column of K underflows to 0, P = 0/0
    algo = ga.CpdRegistration(ctx)
    cfg = ga.CpdConfiguration(maxIterations=50, w=0.0, initialSigma=1.0)
    st = algo.createInitialState(model, bad_target, cfg)
    assert abs(st.general.sigma2 - 1.0) < 1e-12 and algo.retryCounter == 10
    rnd = np.random.default_rng(5)
    s0 = algo.update(st, probabilistic=True, rnd=rnd)              # iteration 0: unchanged, no error, counter untouched
    assert s0.general.status == 0 and s0.general.iteration == 1 and algo.retryCounter == 10
    cur = s0
    for k in range(10):
        nxt = algo.update(cur, probabilistic=True, rnd=rnd)
        assert nxt.general.status == 0 and nxt.general.iteration == cur.general.iteration + 1
        assert np.array_equal(nxt.general.modelParameters.shape, cur.general.modelParameters.shape)
        assert nxt.general.sigma2 == cur.general.sigma2 and algo.retryCounter == 9 - k
        cur = nxt
    failed = algo.update(cur, probabilistic=True, rnd=rnd)
    assert failed.general.status == ga.FittingStatuses.ModelFlexibilityError and algo.retryCounter == 0
    # a good posterior (same algorithm instance, other target) gives one retry back
    good = dataclasses.replace(cur.general, target=target, sigma2=50.0)
    ok = algo.update(cur.updateGeneral(good), probabilistic=True, rnd=rnd)
    assert ok.general.status == 0 and algo.retryCounter == 1
    # deterministic: immediate error, counter untouched
    det = algo.update(cur, probabilistic=False)
    assert det.general.status == ga.FittingStatuses.ModelFlexibilityError and algo.retryCounter == 1
    algo.close()


def test_failed_projection_at_iteration_zero_is_an_error(ctx):
    """GingrAlgorithm.scala:248-251 vs :206-208: the iteration-0 exemption covers the posterior only."""
    import gingr_amd as ga
    mo, model, target = setup(M=200, rank=12)
    algo = ga.CpdRegistration(ctx)
    st = algo.createInitialState(model, target, ga.CpdConfiguration(maxIterations=50, w=0.1), stepLength=float("inf"))
    s1 = algo.update(st)
    st_o = go.initial_state(mo, st.general.sigma2, step_length=float("inf"))
    with np.errstate(all="ignore"):
        o1 = go.cpd_update(mo, target, st_o, w=0.1)
    assert st.general.iteration == 0
    assert s1.general.status == o1.status == ga.FittingStatuses.ModelFlexibilityError
    assert np.array_equal(s1.general.modelParameters.shape, st.general.modelParameters.shape)
    algo.close()


@pytest.mark.parametrize("rank", [28, 120, 150, 512])
def test_transition_density_memo_slots_and_cached_factors(ctx, rank):
    """A Metropolis-Hastings step alternates between two states (GingrAlgorithm.scala:68 memoises computePosterior): the second
    memo slot brings a state's [G, rhs] back instead of recomputing it, and a state whose factors are on the device answers
    further density queries from them.  Every route must give the value a fresh instance computes from scratch."""
    import gingr_amd as ga
    if rank <= 100:
        mo, model, target = setup(seed=17, rank=rank)
    else:
        rng = np.random.default_rng(33)
        M = 400
        ref = rng.normal(0, 40, (M, 3))
        U, _ = np.linalg.qr(rng.normal(0, 1, (3 * M, rank)))
        lam = np.sort(rng.uniform(1.0, 300.0, rank))[::-1].copy()
        mo = go.PDM(ref=ref, mean=rng.normal(0, 0.2, (M, 3)), U=np.ascontiguousarray(U), lam=lam)
        target = mo.instance(rng.normal(0, 1.0, rank)) + rng.normal(0, 0.3, (M, 3))
        model = ga.PointDistributionModel(mo.ref, mo.mean, mo.U, mo.lam)
    cfg = ga.CpdConfiguration(maxIterations=50, w=0.05)

    def fresh():
        a = ga.CpdRegistration(ctx)
        return a, a.createInitialState(model, target, cfg)

    algo, s0 = fresh()
    A = algo.update(s0)
    B = algo.update(A, probabilistic=True, rnd=np.random.default_rng(4))
    C = algo.update(B, probabilistic=True, rnd=np.random.default_rng(5))
    # reference values: one fresh instance per query (nothing memoised)
    want = {}
    for name, (frm, to) in {"AB": (A, B), "BA": (B, A), "AC": (A, C), "BC": (B, C)}.items():
        a2, _ = fresh()
        want[name] = a2.logTransitionProbability(frm, to)
        a2.close()
    # the sequence of a chain: q(B|A), q(A|B) [A goes to the second slot], q(C|A) [swap back, cached factors], q(C|B) [swap, cached]
    got = [algo.logTransitionProbability(A, B), algo.logTransitionProbability(B, A), algo.logTransitionProbability(A, C),
           algo.logTransitionProbability(B, C), algo.logTransitionProbability(A, B)]
    for g, k in zip(got, ["AB", "BA", "AC", "BC", "AB"]):
        assert np.isfinite(g) and abs(g - want[k]) <= 1e-9 * abs(want[k]), (k, g, want[k])
    # a sampled proposal from a state that comes back from the second slot = the proposal of a fresh instance (same draws)
    _ = algo.logTransitionProbability(B, A)                     # live: B, second slot: A
    p1 = algo.update(A, probabilistic=True, rnd=np.random.default_rng(8))
    a3, _ = fresh()
    p2 = a3.update(A, probabilistic=True, rnd=np.random.default_rng(8))
    assert np.array_equal(p1.general.fit, p2.general.fit) and p1.general.sigma2 == p2.general.sigma2
    # and the deterministic path never takes a posterior out of the second slot (its correspondence getters must stay valid)
    _ = algo.logTransitionProbability(B, A)
    d1 = algo.update(A)
    d2 = a3.update(A)
    assert np.array_equal(d1.general.fit, d2.general.fit)
    a3.close()
    algo.close()

"""CPU tests: the oracle's likelihood / metric restatements against closed forms, and the host-side chain logic of
gingr_amd.sampling (mixture selection and density, Metropolis-Hastings acceptance, best-sample logger) on stub generators."""
import math

import numpy as np
import scipy.stats

from oracle import gingr_oracle as go


def square_mesh():
    v = np.array([[0.0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
    t = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    return v, t


def test_gaussian_logpdf_and_prior_closed_forms():
    x = np.array([0.0, 0.3, -2.5, 7.0])
    assert np.allclose(go.gaussian_logpdf(x, 1.7), scipy.stats.norm.logpdf(x, 0.0, 1.7), rtol=1e-14, atol=1e-14)
    a = np.random.default_rng(0).normal(0, 1, 17)
    assert abs(go.model_evaluator_logvalue(a) - scipy.stats.multivariate_normal(np.zeros(17), np.eye(17)).logpdf(a)) < 1e-12


def test_surface_distance_stats_known_answers():
    v, t = square_mesh()
    pts = np.array([[0.5, 0.5, 2.0],      # above the interior: distance 2
                    [2.0, 0.5, 0.0],      # beside the edge x = 1: distance 1
                    [-3.0, -4.0, 0.0],    # diagonal from the corner (0,0): distance 5
                    [0.25, 0.75, 0.0]])   # on the surface: distance 0
    s, mx, n, ll = go.surface_distance_stats(pts, v, t, False, 2.0)
    assert (s, mx, n) == (8.0, 5.0, 4)
    assert abs(ll - scipy.stats.norm.logpdf([2.0, 1.0, 5.0, 0.0], 0, 2.0).sum()) < 1e-13
    # every vertex of the square is a boundary vertex: boundary-aware keeps nothing
    assert go.surface_distance_stats(pts, v, t, True)[2] == 0
    assert go.avg_distance(pts, v, t) == 2.0 and go.max_distance(pts, v, t) == 5.0
    v2 = v + np.array([0.0, 0.0, 3.0])
    assert go.hausdorff_distance(v, t, v2, t) == 3.0
    assert go.independent_point_distance_logvalue(v2, t, v, t, 1.0, "Symmetric") == \
        0.5 * go.independent_point_distance_logvalue(v2, t, v, t, 1.0, "ModelToTarget") + \
        0.5 * go.independent_point_distance_logvalue(v2, t, v, t, 1.0, "TargetToModel")


def test_degenerate_triangles_never_win():
    v = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 1, 0], [0, 0, 0.0]])
    t = np.array([[0, 1, 2], [0, 1, 3], [0, 4, 3]], dtype=np.int32)       # collinear cell, proper cell, cell with a repeated point
    p = np.array([[0.25, 0.25, 1.0], [3.0, 0.0, 0.0], [-1.0, -1.0, 0.0]])
    cp, d2 = go.mesh_closest_point(p, v, t)
    assert np.all(np.isfinite(cp)) and np.allclose(d2, [1.0, 1.0, 2.0])


# ---------------------------------------------------------------------------------------- host chain logic on stubs
class _State:
    def __init__(self, x):
        self.x = x


class _Walk:
    def __init__(self, sdev, gen):
        self.sdev, self.gen = sdev, gen

    def propose(self, s):
        return _State(s.x + self.sdev * float(self.gen.standard_normal()))

    def logTransitionProbability(self, f, t):
        return float(scipy.stats.norm.logpdf(t.x - f.x, 0.0, self.sdev))


class _Jump:                                   # asymmetric proposal: always +1
    def propose(self, s):
        return _State(s.x + 1.0)

    def logTransitionProbability(self, f, t):
        return 0.0 if t.x == f.x + 1.0 else -math.inf


class _Target:
    def logValue(self, s):
        return float(scipy.stats.norm.logpdf(s.x, 1.0, 0.5))


def test_mixture_picks_by_cumulative_weight_and_sums_densities():
    import torch  # noqa: F401
    from gingr_amd import sampling as sp
    rnd = sp.Random(3)
    a, b = _Walk(0.1, rnd.scalaRandom), _Walk(2.0, rnd.scalaRandom)
    mix = sp.MixtureProposal([(1.0, a), (3.0, b)], rnd)
    assert mix.factors == [0.25, 0.75] and mix.cumulative[-1] == 1.0
    f, t = _State(0.0), _State(0.7)
    want = math.log(0.25 * math.exp(a.logTransitionProbability(f, t)) + 0.75 * math.exp(b.logTransitionProbability(f, t)))
    assert abs(mix.logTransitionProbability(f, t) - want) < 1e-14
    only_inf = sp.MixtureProposal([(1.0, _Jump())], rnd)
    assert only_inf.logTransitionProbability(f, t) == -math.inf
    # component frequencies follow the weights
    picks = {0: 0, 1: 0}

    class Probe:
        def __init__(self, k):
            self.k = k

        def propose(self, s):
            picks[self.k] += 1
            return s

    m2 = sp.MixtureProposal([(1.0, Probe(0)), (3.0, Probe(1))], sp.Random(9))
    for _ in range(4000):
        m2.propose(f)
    assert abs(picks[1] / 4000.0 - 0.75) < 0.03


def test_metropolis_hastings_samples_the_target_and_handles_the_deterministic_case():
    import torch  # noqa: F401
    from gingr_amd import sampling as sp
    rnd = sp.Random(11)
    chain = sp.MetropolisHastings(sp.MixtureProposal([(0.7, _Walk(0.6, rnd.scalaRandom)), (0.3, _Walk(0.05, rnd.scalaRandom))], rnd),
                                  _Target(), rnd)
    s, xs = _State(0.0), []
    for k in range(20000):
        s = chain.next(s)
        if k >= 1000:
            xs.append(s.x)
    assert abs(np.mean(xs) - 1.0) < 0.05 and abs(np.std(xs) - 0.5) < 0.05

    class Det:                                   # GeneratorWrapperDeterministic: density -inf both ways -> ratio 0 -> accepted
        def propose(self, s):
            return _State(s.x + 1.0)

        def logTransitionProbability(self, f, t):
            return -math.inf

    class Flat:
        def logValue(self, s):
            return 0.0

    c2 = sp.MetropolisHastings(Det(), Flat(), sp.Random(0))
    s = _State(0.0)
    for _ in range(5):
        s = c2.next(s)
    assert s.x == 5.0
    # an irreversible jump (forward density 1, backward 0) is never accepted
    c3 = sp.MetropolisHastings(_Jump(), Flat(), sp.Random(0))
    s0 = _State(0.0)
    assert c3.next(s0) is s0


def test_best_and_current_sample_logger():
    import torch  # noqa: F401
    from gingr_amd import sampling as sp
    log = sp.BestAndCurrentSampleLogger(_Target())
    for x in (0.0, 0.9, 3.0, 1.05, 2.0):
        log.logState(_State(x))
    assert log.currentSample().x == 2.0 and log.currentBestSample().x == 1.05
    assert abs(log.currentBestValue() - scipy.stats.norm.logpdf(1.05, 1.0, 0.5)) < 1e-14


def test_json_state_logger_entries_and_file(tmp_path):
    """JSONStateLogger (JSONStateLogger.scala:103-146): accepted entries carry the parameters, rejected ones do not; the index is
    the running sample count; the file round-trips through gingr_amd.io and reconstructs the chain state at any position."""
    import dataclasses
    import torch  # noqa: F401
    import gingr_amd as ga
    from gingr_amd import sampling as sp

    @dataclasses.dataclass
    class G:
        modelParameters: object
        generatedBy: str

    @dataclasses.dataclass
    class S:
        general: G

    def state(a, name):
        mp = ga.ModelFittingParameters(1.0, (0.1 * a, 0.0, -a), ga.EulerAngles(0.01 * a, 0.0, 0.02), (0.0, 0.0, 0.0), np.array([a, -a, 2.0 * a]))
        return S(G(mp, name))

    class Ev(sp.AcceptAll):
        def evaluator(self):
            class E:
                def logValue(self_inner, s):
                    return -float(np.sum(np.asarray(s.general.modelParameters.shape) ** 2))
            return [sp.EvaluatorIdentifier("Prior", E()), sp.EvaluatorIdentifier("Distance", sp.AcceptAllEvaluator())]

    path = str(tmp_path / "chain.json")
    log = sp.JSONStateLogger(Ev(), path)
    s0, s1, s2, s3 = state(0.0, ""), state(1.0, "ICP"), state(2.0, "RandomShape-0.1"), state(3.0, "ICP")
    log.accept(s0, s0, None, None)
    log.accept(s0, s1, None, None)
    log.reject(s1, s2, None, None)
    log.accept(s1, s3, None, None)
    assert [e.index for e in log.log] == [0, 1, 2, 3] and [e.status for e in log.log] == [True, True, False, True]
    assert log.log[2].modelParameters == [] and log.log[2].name == "RandomShape-0.1"
    assert log.log[1].logvalue == {"Prior": -6.0, "Distance": 0.0, "product": -6.0}
    assert log.totalSamples == 4 and log.percentRejected == 0.25 and log.percentAcceptedOfType("ICP") == 1.0
    log.writeLog()
    back = ga.io.read_log(path)
    assert len(back) == 4
    at2 = ga.io.parameters_of_log_entry(back, 2)                 # the rejected entry: the state is still s1's
    assert np.array_equal(at2.shape, s1.general.modelParameters.shape) and at2.translation == s1.general.modelParameters.translation
    import pytest
    with pytest.raises(IOError):
        sp.JSONStateLogger(Ev(), str(tmp_path / "missing_dir" / "x.json"))

"""One native call per Metropolis-Hastings step (gingr_fitter_mh_step, VERDICT r03 task 4): the fused step must make the decisions
and produce the states of the call-by-call path (update / proposeParameters, surfaceDistanceStats, two logTransitionProbability
queries -- itself checked against the oracle's chain in test_gpu_sampling.py, which now also runs fused by default)."""
import ctypes
import math

import numpy as np
import pytest

from oracle import gingr_oracle as go
from tests.test_gpu_surface_icp import femur, make_state
from tests.test_gpu_sampling import _cpd_chain_setup
from gingr_amd.api import dptr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import gingr_amd as ga
    c = ga.Context(0)
    yield c
    c.close()


class _Log:
    def __init__(self):
        self.flags, self.kinds = [], []

    def accept(self, cur, prop, gen, ev):
        self.flags.append(True)
        self.kinds.append(prop.general.generatedBy)

    def reject(self, cur, prop, gen, ev):
        self.flags.append(False)
        self.kinds.append(prop.general.generatedBy)


def _run(algo, s0, settings, seed):
    from gingr_amd import sampling as sp
    states, log = [], _Log()
    best = algo.run(s0, callBackLogger=states.append, acceptRejectLogger=log, probabilisticSettings=settings, rnd=sp.Random(seed))
    return best, states, log


def _compare(a, b, tol):
    (best_a, states_a, log_a), (best_b, states_b, log_b) = a, b
    assert log_a.flags == log_b.flags and log_a.kinds == log_b.kinds
    assert len(states_a) == len(states_b)
    for k, (x, y) in enumerate(zip(states_a, states_b)):
        px, py = x.general.modelParameters, y.general.modelParameters
        assert np.abs(np.asarray(px.shape) - np.asarray(py.shape)).max() <= tol, k
        assert np.abs(np.asarray(px.translation) - np.asarray(py.translation)).max() <= tol, k
        assert abs(px.rotation.phi - py.rotation.phi) + abs(px.rotation.theta - py.rotation.theta) + abs(px.rotation.psi - py.rotation.psi) <= tol
        assert x.general.iteration == y.general.iteration and x.general.status == y.general.status
        assert x.general.sigma2 == y.general.sigma2
        assert np.abs(np.asarray(x.general.fit) - np.asarray(y.general.fit)).max() <= tol * max(1.0, np.abs(np.asarray(y.general.fit)).max()), k
    assert np.abs(np.asarray(best_a.general.modelParameters.shape) - np.asarray(best_b.general.modelParameters.shape)).max() <= tol


@pytest.mark.parametrize("points", [0, 700])
def test_fused_surface_icp_chain_equals_the_call_by_call_chain(ctx, points):
    """DemoICP's configuration on the femur (surface ICP proposals + the stock random walks, model-to-target likelihood over all /
    the first 700 vertices): 60 steps, same draws.  (Seed 17: five and seven rejected informed proposals in these 60 steps -- a chain's
    realisation changes with the last bits of the model's constants, because the surface correspondence's rejection rules are
    discontinuous, so the check that both outcomes occur wants a realisation with some margin;
    tools/experiments/chain_seeds.py.)"""
    import gingr_amd as ga
    from gingr_amd import sampling as sp
    ref, cells, target, tcells = femur()
    runs = []
    for fused in (True, False):
        mo, algo, s0 = make_state(ctx, ref, cells, target, tcells, rank=24, sigma=(1.0, 1.0), iters=61)
        settings = sp.ProbabilisticSettings(sp.IndependentPoints(algo, s0, 5.0, modelPointCount=points or None), randomMixture=0.5,
                                            fusedSteps=fused)
        runs.append(_run(algo, s0, settings, 17))
        assert algo._mh is None and algo._mh_last is None          # run() switches the mode off again
        algo.close()
    _compare(runs[0], runs[1], 1e-9)
    flags, kinds = runs[0][2].flags[1:], runs[0][2].kinds[1:]
    assert any(flags) and not all(flags)
    assert {"ICP"} < set(kinds)                                     # informed and random-walk proposals both occurred
    assert any(f for f, k in zip(flags, kinds) if k == "ICP") and any(not f for f, k in zip(flags, kinds) if k == "ICP")


def test_fused_cpd_and_pointcloud_icp_chains_equal_the_call_by_call_chains(ctx):
    import gingr_amd as ga
    from gingr_amd import sampling as sp
    mo, model, target, cells, tcells = _cpd_chain_setup(ctx)
    for make in (lambda: (ga.CpdRegistration(ctx), ga.CpdConfiguration(maxIterations=31, w=0.05)),
                 lambda: (ga.IcpRegistration(ctx), ga.IcpConfiguration(maxIterations=31, initialSigma=2.0, endSigma=0.5))):
        runs = []
        for fused in (True, False):
            algo, cfg = make()
            s0 = algo.createInitialState(model, target, cfg, targetCells=tcells)
            settings = sp.ProbabilisticSettings(sp.IndependentPoints(algo, s0, 1.0), randomMixture=0.4, fusedSteps=fused)
            runs.append(_run(algo, s0, settings, 5))
            algo.close()
        _compare(runs[0], runs[1], 1e-9)
        assert any(runs[0][2].flags[1:]) and not all(runs[0][2].flags[1:])


def test_mh_step_against_the_separate_entry_points_and_restore(ctx):
    """The raw C ABI: one step's numbers equal update_sample + distance stats + two posterior_logpdf calls; mh_restore brings the
    start state back bit for bit; argument errors."""
    import gingr_amd as ga
    from gingr_amd import _native as nat
    ref, cells, target, tcells = femur()
    mo, algo, s0 = make_state(ctx, ref, cells, target, tcells, rank=20, sigma=(1.0, 1.0), iters=10)
    lib, r, M = ctx._lib, s0.general.model.rank, ref.shape[0]
    z = np.random.default_rng(2).normal(0, 1, r)
    # call by call
    s1 = algo.update(s0, True, np.random.default_rng(2))
    stats = algo.surfaceDistanceStats(s1, 0, 0, None, False, 5.0)
    fw, bw = algo.logTransitionProbability(s0, s1), algo.logTransitionProbability(s1, s0)
    # fused, raw
    algo._push_state(s0.general)
    f = algo._fitter
    ip = nat.IcpParams(1.0, 1.0, 10)
    req = nat.MhRequest()
    req.flavour, req.kind, req.icp, req.z, req.eval_sdev, req.eval_points, req.need_forward = 2, 0, ctypes.pointer(ip), dptr(z), 5.0, 0, 1
    alpha, fit, res = np.empty(r), np.empty((M, 3)), nat.MhResult()
    assert lib.gingr_fitter_mh_step(f, ctypes.byref(req), dptr(alpha), dptr(fit), ctypes.byref(res)) == 0
    assert np.abs(alpha - np.asarray(s1.general.modelParameters.shape)).max() < 1e-12
    assert np.abs(fit - np.asarray(s1.general.fit)).max() < 1e-10
    assert res.scalars.iteration == s1.general.iteration and res.scalars.status == s1.general.status
    assert abs(res.log_value - stats[3]) <= 1e-10 * abs(stats[3]) and res.count == stats[2] and abs(res.dist_max - stats[1]) < 1e-12
    assert abs(res.log_q_forward - fw) <= 1e-9 * abs(fw) and abs(res.log_q_backward - bw) <= 1e-9 * abs(bw)
    assert res.forward_status == 0 and res.backward_status == 0
    # reject: the start state again, bit for bit
    assert lib.gingr_fitter_mh_restore(f) == 0
    back = algo._pull_state(s0.general)
    assert np.array_equal(np.asarray(back.modelParameters.shape), np.asarray(s0.general.modelParameters.shape))
    assert np.array_equal(np.asarray(back.fit), np.asarray(s0.general.fit)) and back.iteration == s0.general.iteration
    assert lib.gingr_fitter_mh_restore(f) != 0                     # nothing to restore twice
    # kind 1: parameters given, likelihood over the first 100 vertices
    a1 = np.asarray(s0.general.modelParameters.shape) + 0.01
    sc = nat.StateScalars()
    sc.euler[:] = [0.01, 0.0, -0.01]
    sc.center[:] = [0.0, 0.0, 0.0]
    sc.translation[:] = [0.1, 0.0, 0.2]
    sc.scale, sc.sigma2, sc.iteration, sc.status = 1.0, s0.general.sigma2, 1, 0
    req.kind, req.z, req.alpha, req.scalars, req.eval_points = 1, None, dptr(a1), ctypes.pointer(sc), 100
    assert lib.gingr_fitter_mh_step(f, ctypes.byref(req), dptr(alpha), None, ctypes.byref(res)) == 0
    assert np.array_equal(alpha, a1) and res.count == 100 and res.scalars.iteration == 1
    assert abs(res.log_q_forward - fw) <= 1e-9 * abs(fw)           # q(.|x) is a number of x alone (the reference projects from.fit)
    req.need_forward = 0
    assert lib.gingr_fitter_mh_restore(f) == 0
    assert lib.gingr_fitter_mh_step(f, ctypes.byref(req), dptr(alpha), None, ctypes.byref(res)) == 0
    assert math.isnan(res.log_q_forward) and res.forward_status == -1 and res.backward_status == 0
    inst = go.model_instance_shape_pose_scale(mo, go.State(alpha=a1, euler=(0.01, 0.0, -0.01), center=np.zeros(3), translation=np.array([0.1, 0.0, 0.2]),
                                                            scale=1.0, sigma2=1.0, fit=np.zeros((M, 3))))
    d, _ = go.surface_distances(inst[:100], target, tcells)
    assert abs(res.log_value - float(np.sum(go.gaussian_logpdf(d, 5.0)))) <= 1e-9 * abs(res.log_value)
    # errors
    req.eval_sdev = 0.0
    assert lib.gingr_fitter_mh_step(f, ctypes.byref(req), dptr(alpha), None, ctypes.byref(res)) == nat.ERR_BAD_ARGUMENT
    req.eval_sdev, req.flavour = 5.0, 7
    assert lib.gingr_fitter_mh_step(f, ctypes.byref(req), dptr(alpha), None, ctypes.byref(res)) == nat.ERR_BAD_ARGUMENT
    req.flavour, req.kind, req.z = 2, 0, None
    assert lib.gingr_fitter_mh_step(f, ctypes.byref(req), dptr(alpha), None, ctypes.byref(res)) == nat.ERR_BAD_ARGUMENT
    algo.close()

"""The reference's classic rigid / similarity ICP baseline (G/other/algorithms/icp/RigidICP.scala, ICPFactory.scala,
RigidICPRegistration.scala, G/other/utils/PoseRegistrator.scala) on the device against the oracle's restatement."""
import os

import numpy as np
import pytest

from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def femur_pair():
    d = np.load(os.path.join(HERE, "golden", "inputs.npz"))
    return d["femur"].astype(np.float64), d["femur_target"].astype(np.float64)


@pytest.mark.parametrize("kind", [0, 1])
def test_iterations_match_the_oracle(ctx, kind):
    from gingr_amd import classic
    tpl, tgt = femur_pair()
    tpl = 1.03 * (tpl @ go.euler_to_rot(0.06, -0.04, 0.05).T) + np.array([4.0, -3.0, 2.0])
    task = classic.ICPFactory(ctx, tpl, kind).registerRigidly(tgt)
    fit = tpl
    for it in range(4):
        got, dist = task.Iteration()
        want, wdist, (s, R, t) = go.rigid_icp_iteration(fit, tgt, similarity=bool(kind))
        assert abs(dist - wdist) < 1e-11 * wdist, (it, dist, wdist)
        assert np.abs(got - want).max() < 1e-9, (it, np.abs(got - want).max())
        gs, gR, gt = task.transform()
        assert abs(gs - s) < 1e-11 and np.abs(gR - R).max() < 1e-11 and np.abs(gt - t).max() < 1e-8
        if kind == 0:
            assert gs == 1.0
        fit = want
    # Iteration(template) restarts from the caller's points
    got, dist = task.Iteration(tpl)
    want, wdist, _ = go.rigid_icp_iteration(tpl, tgt, similarity=bool(kind))
    assert np.abs(got - want).max() < 1e-9 and abs(dist - wdist) < 1e-11 * wdist
    task.close()


def test_registration_loop_and_the_wrapper(ctx):
    from gingr_amd import classic
    tpl, tgt = femur_pair()
    moved = tpl @ go.euler_to_rot(0.1, 0.05, -0.08).T + np.array([6.0, 2.0, -4.0])
    task = classic.ICPFactory(ctx, moved).registerRigidly(tgt)
    got = task.Registration(60, tolerance=1e-3)
    want, iters, conv = go.rigid_icp_registration(moved, tgt, 60, 1e-3)
    assert task.iterations == iters and task.converged == conv
    assert np.abs(got - want).max() < 1e-7
    d_before = np.sqrt(go.icp_closest_point(moved, tgt)[1]).mean()
    d_after = np.sqrt(go.icp_closest_point(got, tgt)[1]).mean()
    assert d_after < 0.5 * d_before
    task.close()
    reg = classic.RigidICPRegistration(ctx, moved, max_iterations=60).register(tgt)
    assert np.abs(reg - want).max() < 1e-7          # the warp field carries the registered template (same point counts here)
    with pytest.raises(Exception):
        classic.ICPFactory(ctx, moved, registrator=7).registerRigidly(tgt)


def test_large_clouds_use_the_pruned_search(ctx):
    """20k <-> 20k: bit-exact closest points against the brute-force oracle are covered elsewhere; here the loop has to agree
    with the oracle's loop to rounding at a size where the nearest-first pruning is active."""
    from gingr_amd import classic
    rng = np.random.default_rng(9)
    tgt = rng.normal(0, 50, (20000, 3))
    tpl = tgt[rng.permutation(20000)] @ go.euler_to_rot(0.02, -0.01, 0.015).T + np.array([1.0, 0.5, -0.5])
    task = classic.ICPFactory(ctx, tpl).registerRigidly(tgt)
    got, dist = task.Iteration()
    want, wdist, _ = go.rigid_icp_iteration(tpl, tgt)
    assert abs(dist - wdist) < 1e-11 * wdist and np.abs(got - want).max() < 1e-9
    task.close()

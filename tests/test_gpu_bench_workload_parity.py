"""Parity of the FUSED update at the sizes the benchmark runs: the metric workload itself
(bench.py: synthetic 50k <-> 50k, rank-100 Gaussian GPMM built on the device, w = 0.1, rigid) and BASELINE config 4
(100k <-> 100k) -- full `update` iterations of the device-resident fitter against ONE oracle update from the same state
(C streaming statistics for the all-pairs part, numpy for the GP part), single shard and through two logical row shards.

Tolerances (BASELINE.json north_star): vertex positions <= 1e-5 relative; sigma2 <= 1e-8, shape coefficients <= 1e-4.
The oracle (`oracle/`) is a restatement of the reference (parity unpinned)."""
import ctypes

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import gingr_oracle as go

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def _oracle_state(mo, alpha, sc, fit):
    return go.State(alpha=np.array(alpha), euler=tuple(sc.euler), center=np.array(sc.center), translation=np.array(sc.translation),
                    scale=float(sc.scale), sigma2=float(sc.sigma2), fit=np.array(fit), iteration=int(sc.iteration),
                    status=int(sc.status), global_transformation=go.RIGID_TRANSFORMS, step_length=1.0)


def _check_one_update(fitter, mo, x, w, tag):
    """state_k (device) -> one oracle update and one device update from it; compare."""
    alpha, sc, fit = fitter.get_state()
    st = _oracle_state(mo, alpha, sc, fit)
    # the fit the device holds IS modelInstanceShapePoseScale of its state (the oracle's own instance agrees <= 1e-9)
    assert rel(go.model_instance_shape_pose_scale(mo, st), fit) < 1e-9, tag
    st1 = go.cpd_update(mo, x, st, w=w, stats=co.cpd_stats(st.fit, x, st.sigma2, w))
    fitter.update_cpd(w, 1.0, 1)
    a1, sc1, fit1 = fitter.get_state()
    assert sc1.status == st1.status == 0 and sc1.iteration == st1.iteration, tag
    e_fit, e_s2, e_a = rel(fit1, st1.fit), abs(sc1.sigma2 - st1.sigma2) / st1.sigma2, rel(a1, st1.alpha)
    assert e_fit < 1e-5, (tag, "fit", e_fit)
    assert e_s2 < 1e-8, (tag, "sigma2", e_s2)
    assert e_a < 1e-4, (tag, "alpha", e_a)
    assert np.allclose(sc1.euler, st1.euler, atol=1e-8) and np.allclose(sc1.translation, st1.translation, atol=1e-6), tag
    return e_fit, e_s2, e_a


def _workload(ctx, n, rank):
    import gingr_amd as ga
    from bench import synth_clouds
    y, x = synth_clouds(n)
    model = ga.GPMMTriangleMesh3D(ctx, y, relativeTolerance=0.0, maxRank=rank).Gaussian(70.0, 50.0)
    host = model.to_host()
    mo = go.PDM(host.reference, host.mean, np.ascontiguousarray(host.basis), host.variance)
    return y, x, model, mo


@pytest.mark.parametrize("n,steps", [(50000, (0, 10)), (100000, (3,))])
def test_fused_update_at_bench_size_against_oracle(ctx, n, steps):
    from gingr_amd.sharded import ShardedFitter
    y, x, model, mo = _workload(ctx, n, 100)
    f = ShardedFitter(ctx, model, x)
    s2 = ctx.cpd_initial_sigma2(y, x)
    f.set_state(np.zeros(mo.rank), s2)
    done = 0
    for k in steps:
        f.update_cpd(0.1, 1.0, k - done)
        done = k
        _check_one_update(f, mo, x, 0.1, f"n={n} after {k} steps")
        done += 1
    f.close()


def test_fused_update_at_bench_size_rank256_against_oracle(ctx):
    """SURVEY 8d's second model rank at the metric size: 50k <-> 50k, rank 256 -- the wide Gram pass
    (gp_wide.hip), the super-panel posterior solve and the rank-256 fit pass inside full `update` iterations, from the initial
    state and from the state after eight device iterations, against the oracle (GingrAlgorithm.scala:192-254, :297-301)."""
    from gingr_amd.sharded import ShardedFitter
    y, x, model, mo = _workload(ctx, 50000, 256)
    assert mo.rank == 256
    f = ShardedFitter(ctx, model, x)
    f.set_state(np.zeros(mo.rank), ctx.cpd_initial_sigma2(y, x))
    _check_one_update(f, mo, x, 0.1, "n=50000 rank 256, first update")
    f.update_cpd(0.1, 1.0, 7)
    _check_one_update(f, mo, x, 0.1, "n=50000 rank 256 after 8 steps")
    f.close()


def test_fused_update_50k_two_logical_shards_against_oracle(ctx):
    """The phase / exchange protocol on two row shards of the metric workload (hand-rolled all-reduce on one device), one
    iteration from the state after 5 single-shard steps, against the oracle's unsharded update."""
    import torch
    from gingr_amd import _native as nat
    from gingr_amd.sharded import NUM_PHASES, NUM_SEGMENTS, ShardedFitter
    y, x, model, mo = _workload(ctx, 50000, 100)
    single = ShardedFitter(ctx, model, x)
    single.set_state(np.zeros(mo.rank), ctx.cpd_initial_sigma2(y, x))
    single.update_cpd(0.1, 1.0, 5)
    alpha, sc, fit = single.get_state()
    single.close()
    st = _oracle_state(mo, alpha, sc, fit)
    st1 = go.cpd_update(mo, x, st, w=0.1, stats=co.cpd_stats(st.fit, x, st.sigma2, 0.1))

    shards = [ShardedFitter(ctx, model, x, rank=r, world=2, all_reduce=None, defer_setup=True) for r in range(2)]

    def allreduce(tensors):
        ctx.synchronize()
        tot = torch.stack(tensors).sum(0)
        for t in tensors:
            t.copy_(tot)
        torch.cuda.synchronize()

    allreduce([s.gram_tensor() for s in shards])
    for s in shards:
        s.finish_setup()
        s.set_state(alpha, sc.sigma2, euler=tuple(sc.euler), center=tuple(sc.center), translation=tuple(sc.translation),
                    scale=sc.scale, iteration=sc.iteration, status=sc.status)
    p = nat.CpdParams(0.1, 1.0)
    for ph in range(NUM_PHASES):
        for s in shards:
            assert s._lib.gingr_fitter_cpd_phase_async(s.handle, ctypes.byref(p), ph) == 0
        if ph < NUM_SEGMENTS:
            allreduce([s._segment(ph) for s in shards])
    fits = []
    for s in shards:
        a1, sc1, f1 = s.get_state()
        fits.append(f1)
        assert sc1.status == 0 and sc1.iteration == st1.iteration
        assert abs(sc1.sigma2 - st1.sigma2) < 1e-8 * st1.sigma2
        assert rel(a1, st1.alpha) < 1e-4
    assert rel(np.concatenate(fits), st1.fit) < 1e-5
    for s in shards:
        s.close()


def test_fused_icp_update_at_bench_size_against_oracle(ctx):
    """Point-cloud ICP at 50k <-> 50k (the secondary measurement of profiles/r02_icp_pointcloud_50k.json): one update from the state
    after three device iterations against the oracle -- correspondences by the C checker's exact nearest neighbour (bit-exact
    indices), GP part in numpy.  Covers the eigen-form posterior of the uniform-weight case at this size."""
    from gingr_amd.sharded import ShardedFitter
    y, x, model, mo = _workload(ctx, 50000, 100)
    f = ShardedFitter(ctx, model, x)
    f.set_state(np.zeros(mo.rank), 25.0)
    f.update_icp(25.0, 1.0, 20, 3)
    alpha, sc, fit = f.get_state()
    st = _oracle_state(mo, alpha, sc, fit)
    idx, d2, _ = co.nn(st.fit, x)
    s2n = go.icp_update_sigma2(st.sigma2, 25.0, 1.0, 20)
    st1 = go.update_from_observations(mo, st, np.arange(mo.M), x[idx], np.full(mo.M, st.sigma2), s2n, None)
    f.update_icp(25.0, 1.0, 20, 1)
    a1, sc1, fit1 = f.get_state()
    assert sc1.status == st1.status == 0 and sc1.iteration == st1.iteration
    assert rel(fit1, st1.fit) < 1e-5 and abs(sc1.sigma2 - st1.sigma2) <= 1e-12 * st1.sigma2 and rel(a1, st1.alpha) < 1e-4
    assert np.allclose(sc1.euler, st1.euler, atol=1e-8) and np.allclose(sc1.translation, st1.translation, atol=1e-6)
    f.close()

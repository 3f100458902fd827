"""The N > 1 path on CPU: world_size-2 and -3 `gloo` runs of the row-sharded update.

Each rank owns a contiguous share of the reference rows (uneven: 101 rows), runs the three phases (oracle-backed restatement of the native phases, same
exchange-segment layout) and all-reduces every segment with torch.distributed -- through the SAME driver loop
(gingr_amd.sharded.drive_update) and the SAME row partition (shard_rows) that bench.py uses with RCCL on the GPUs.
The result must equal the unsharded oracle update -- for CPD, for ICP with the point-cloud and with the surface correspondence
(gather of the fit = exchange segment 2), for one sampled proposal (replicated draw) and for the transition density.
"""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, transform, out_dir):
    import torch
    import torch.distributed as dist
    from gingr_amd.sharded import drive_update, shard_rows
    from oracle import gingr_oracle as go
    from tests.sharded_oracle import OracleShard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    ref = rng.normal(0, 30, (101, 3))                      # odd size: uneven shards
    mo = go.build_gaussian_gpmm(ref, 45.0, 25.0, rel_tol=1e-9, max_rank=14)
    target = (mo.instance(rng.normal(0, 1, mo.rank)) @ go.euler_to_rot(0.1, -0.05, 0.08).T + 1.5)[:90] + rng.normal(0, 0.2, (90, 3))
    b, e = shard_rows(mo.M, world, rank)
    sh = OracleShard(mo, target, b, e, global_transform=transform, w=0.1)
    mom = torch.from_numpy(sh.mom_local.copy())     # gingr_model_gram_exchange + all-reduce + gingr_model_finalize
    dist.all_reduce(mom)
    sh.finalize(mom.numpy())
    st = go.initial_state(mo, go.cpd_initial_sigma2(mo.ref + mo.mean, target), global_transformation=transform)
    sh.set_state(st)

    def all_reduce_segment(k):
        t = torch.from_numpy(sh.seg(k))        # shares memory with the exchange buffer
        dist.all_reduce(t)

    for _ in range(3):
        drive_update(sh.phase, all_reduce_segment, world)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), alpha=sh.st.alpha, fit=sh.fit, b=b, e=e, sigma2=sh.st.sigma2,
             euler=np.array(sh.st.euler), t=sh.st.translation, scale=sh.st.scale)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,transform", [(2, 1), (2, 2), (3, 1)])
def test_gloo_update_equals_unsharded(tmp_path, world, transform):
    import torch.multiprocessing as mp
    from oracle import gingr_oracle as go
    mp.spawn(_worker, args=(world, _free_port(), transform, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    ref = rng.normal(0, 30, (101, 3))
    mo = go.build_gaussian_gpmm(ref, 45.0, 25.0, rel_tol=1e-9, max_rank=14)
    target = (mo.instance(rng.normal(0, 1, mo.rank)) @ go.euler_to_rot(0.1, -0.05, 0.08).T + 1.5)[:90] + rng.normal(0, 0.2, (90, 3))
    st = go.initial_state(mo, go.cpd_initial_sigma2(mo.ref + mo.mean, target), global_transformation=transform)
    for _ in range(3):
        st = go.cpd_update(mo, target, st, w=0.1)
    assert st.status == 0
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    fit = np.concatenate([p["fit"] for p in parts])
    assert int(parts[0]["b"]) == 0 and int(parts[-1]["e"]) == mo.M
    assert all(int(parts[r]["e"]) == int(parts[r + 1]["b"]) for r in range(world - 1))
    assert np.linalg.norm(fit - st.fit) / np.linalg.norm(st.fit) < 1e-9
    for p in parts:   # replicated state is identical on every rank and equals the unsharded state
        assert np.allclose(p["alpha"], st.alpha, rtol=1e-6, atol=1e-9)
        assert abs(float(p["sigma2"]) - st.sigma2) < 1e-9 * st.sigma2
        assert np.allclose(p["euler"], st.euler, atol=1e-10) and np.allclose(p["t"], st.translation, atol=1e-8)
    assert all(np.array_equal(parts[0]["alpha"], p["alpha"]) for p in parts[1:])


# ---- surface ICP, the sampled proposal and the transition density on row shards (round 4; gingr_fitter_update_sharded_async /
# gingr_fitter_posterior_logpdf_sharded restated per shard in tests/sharded_oracle.py).  Every update starts from the SAME state on
# every rank and in the unsharded oracle: the surface correspondence rejects by exact floating-point comparisons
# (ClosestPointRegistrator.scala:88-96 `f != p`), so trajectories that differ in the last bit may reject different vertices -- in
# the reference as much as here -- and only update-by-update comparisons are meaningful.
ICP_PARAMS = (4.0, 1.0, 10)


def _grid_mesh(n, size, height, seed):
    rng = np.random.default_rng(seed)
    xs = np.linspace(-size, size, n)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = height * np.sin(X / size * 2.0) * np.cos(Y / size * 1.5) + rng.normal(0, 0.05, X.shape)
    v = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    return v, np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.int32)


def _mesh_problem():
    from oracle import gingr_oracle as go
    ref, cells = _grid_mesh(9, 40.0, 6.0, 1)
    tgt, tcells = _grid_mesh(11, 44.0, 7.0, 2)
    tgt = tgt @ go.euler_to_rot(0.03, -0.02, 0.04).T + np.array([0.5, -0.8, 1.0])
    mo = go.build_gaussian_gpmm(ref, 60.0, 20.0, rel_tol=1e-9, max_rank=10)
    return mo, cells, tgt, tcells


def _oracle_step(go, mo, cells, tgt, tcells, st, flavour, z, reversed_direction=False):
    if reversed_direction:
        method = "TriangularClosestPoint" if flavour == 2 else "PointcloudClosestPoint"
        return go.icp_reversed_update(mo, cells, tgt, tcells, st, *ICP_PARAMS, method=method)[0]
    if flavour == 0:
        return go.cpd_update(mo, tgt, st, w=0.1, z=z)
    if flavour == 1:
        return go.icp_update(mo, tgt, st, *ICP_PARAMS, z=z)[0]
    return go.icp_surface_update(mo, cells, tgt, tcells, st, *ICP_PARAMS, z=z)[0]


def _oracle_observations(go, mo, cells, tgt, tcells, st, flavour):
    if flavour == 0:
        return go.cpd_observations(mo, tgt, st, w=0.1)
    if flavour == 1:
        idx, _, _ = go.icp_closest_point(st.fit, tgt)
        return np.arange(mo.M), tgt[idx], np.full(mo.M, st.sigma2)
    cp, w, _ = go.surface_correspondence(st.fit, cells, tgt, tcells)
    pids = np.flatnonzero(w == 1.0)
    return pids, cp[pids], np.full(pids.shape[0], st.sigma2)


def _flavour_worker(rank, world, port, flavour, out_dir):
    import torch
    import torch.distributed as dist
    from gingr_amd.sharded import drive_update, shard_rows
    from oracle import gingr_oracle as go
    from tests.sharded_oracle import OracleShard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mo, cells, tgt, tcells = _mesh_problem()
    b, e = shard_rows(mo.M, world, rank)
    sh = OracleShard(mo, tgt, b, e, global_transform=1, w=0.1, flavour=flavour, icp=ICP_PARAMS, tmpl_tris=cells, tgt_tris=tcells)
    mom = torch.from_numpy(sh.mom_local.copy())
    dist.all_reduce(mom)
    sh.finalize(mom.numpy())

    def all_reduce_segment(k):
        dist.all_reduce(torch.from_numpy(sh.seg(k)))

    st = go.initial_state(mo, go.cpd_initial_sigma2(mo.ref + mo.mean, tgt) if flavour == 0 else ICP_PARAMS[0], global_transformation=1)
    rng = np.random.default_rng(3)
    out = {"b": b, "e": e}
    for it in range(3):
        z = 0.05 * rng.normal(0, 1, mo.rank) if it == 1 else None     # one sampled proposal; the draw is replicated
        sh.set_state(st)
        sh.z = z
        drive_update(sh.phase, all_reduce_segment, world, flavour=flavour)
        out[f"fit{it}"], out[f"alpha{it}"], out[f"sigma2_{it}"] = sh.fit, sh.st.alpha, sh.st.sigma2
        out[f"pose{it}"] = np.concatenate([sh.st.euler, sh.st.translation, [sh.st.scale]])
        st = _oracle_step(go, mo, cells, tgt, tcells, st, flavour, z)
    # the transition density of a mesh under the posterior of the last state (fitter_sharded_logpdf's order of phases and exchanges)
    mesh = st.fit + np.random.default_rng(4).normal(0, 0.1, st.fit.shape)
    sh.set_state(st)
    sh.z = None
    if flavour == 2:
        sh.phase(3)
        all_reduce_segment(2)
    sh.phase(0)
    if flavour == 0:
        all_reduce_segment(0)
    sh.phase(1)
    sh.logpdf_prepare(mesh)
    all_reduce_segment(1)
    out["logpdf"] = sh.logpdf_finish()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,flavour", [(2, 2), (3, 2), (2, 1), (3, 0)])
def test_gloo_surface_icp_sample_and_logpdf_equal_unsharded(tmp_path, world, flavour):
    import torch.multiprocessing as mp
    from oracle import gingr_oracle as go
    mp.spawn(_flavour_worker, args=(world, _free_port(), flavour, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    mo, cells, tgt, tcells = _mesh_problem()
    st = go.initial_state(mo, go.cpd_initial_sigma2(mo.ref + mo.mean, tgt) if flavour == 0 else ICP_PARAMS[0], global_transformation=1)
    rng = np.random.default_rng(3)
    for it in range(3):
        z = 0.05 * rng.normal(0, 1, mo.rank) if it == 1 else None
        st = _oracle_step(go, mo, cells, tgt, tcells, st, flavour, z)
        assert st.status == 0
        fit = np.concatenate([p[f"fit{it}"] for p in parts])
        assert np.linalg.norm(fit - st.fit) / np.linalg.norm(st.fit) < 1e-9, (it, flavour)
        for p in parts:
            assert np.allclose(p[f"alpha{it}"], st.alpha, rtol=1e-6, atol=1e-9)
            assert abs(float(p[f"sigma2_{it}"]) - st.sigma2) <= 1e-9 * st.sigma2
            assert np.allclose(p[f"pose{it}"], np.concatenate([st.euler, st.translation, [st.scale]]), atol=1e-8)
        assert all(np.array_equal(parts[0][f"alpha{it}"], p[f"alpha{it}"]) for p in parts[1:])     # replicated = identical
    if flavour == 2:   # the sampled proposal moved the shape: some correspondences were rejected, some accepted
        _, w, _ = go.surface_correspondence(st.fit, cells, tgt, tcells)
        assert 0 < w.sum() < w.shape[0]
    mesh = st.fit + np.random.default_rng(4).normal(0, 0.1, st.fit.shape)
    want = go.posterior_logpdf_of_mesh(mo, st, *_oracle_observations(go, mo, cells, tgt, tcells, st, flavour), mesh)
    for p in parts:
        assert abs(float(p["logpdf"]) - want) <= 1e-8 * abs(want), (float(p["logpdf"]), want)
    assert all(float(parts[0]["logpdf"]) == float(p["logpdf"]) for p in parts[1:])


def _reversed_worker(rank, world, port, flavour, out_dir):
    import torch
    import torch.distributed as dist
    from gingr_amd.sharded import drive_update, shard_rows
    from oracle import gingr_oracle as go
    from tests.sharded_oracle import OracleShard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mo, cells, tgt, tcells = _mesh_problem()
    b, e = shard_rows(mo.M, world, rank)
    sh = OracleShard(mo, tgt, b, e, global_transform=1, flavour=flavour, icp=ICP_PARAMS, tmpl_tris=cells, tgt_tris=tcells,
                     reversed_direction=True)
    mom = torch.from_numpy(sh.mom_local.copy())
    dist.all_reduce(mom)
    sh.finalize(mom.numpy())
    st = go.initial_state(mo, ICP_PARAMS[0], global_transformation=1)
    out = {"b": b, "e": e}
    for it in range(2):
        sh.set_state(st)
        drive_update(sh.phase, lambda k: dist.all_reduce(torch.from_numpy(sh.seg(k))), world, flavour=flavour, reversed_direction=True)
        out[f"fit{it}"], out[f"alpha{it}"] = sh.fit, sh.st.alpha
        st = _oracle_step(go, mo, cells, tgt, tcells, st, flavour, None, reversed_direction=True)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,flavour", [(2, 2), (3, 1)])
def test_gloo_reversed_direction_equals_unsharded(tmp_path, world, flavour):
    """IcpConfiguration.reverseCorrespondenceDirection on row shards: every rank scans its index range of the target queries against
    the gathered template, the per-template-vertex sums are all-reduced between phases 0 and 1 (GINGR_SEGMENT_REVSUM) and every rank
    keeps the observations of its own rows (gingr_amd/csrc/fitter.hip, run_phase, reversed && sharded)."""
    import torch.multiprocessing as mp
    from oracle import gingr_oracle as go
    mp.spawn(_reversed_worker, args=(world, _free_port(), flavour, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    mo, cells, tgt, tcells = _mesh_problem()
    st = go.initial_state(mo, ICP_PARAMS[0], global_transformation=1)
    for it in range(2):
        st = _oracle_step(go, mo, cells, tgt, tcells, st, flavour, None, reversed_direction=True)
        assert st.status == 0
        fit = np.concatenate([p[f"fit{it}"] for p in parts])
        assert np.linalg.norm(fit - st.fit) / np.linalg.norm(st.fit) < 1e-9, (it, flavour)
        assert all(np.array_equal(parts[0][f"alpha{it}"], p[f"alpha{it}"]) for p in parts[1:])


def test_shard_rows_partition():
    from gingr_amd.sharded import shard_rows
    for M in (1, 7, 8, 50000, 100001):
        for world in (1, 2, 3, 8):
            if world > M:
                continue
            spans = [shard_rows(M, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == M
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1

"""The N > 1 path on CPU: world_size-2 and -3 `gloo` runs of the row-sharded update.

Each rank owns a contiguous share of the reference rows (uneven: 101 rows), runs the three phases (oracle-backed restatement of the native phases, same
exchange-segment layout) and all-reduces every segment with torch.distributed -- through the SAME driver loop
(gingr_amd.sharded.drive_update) and the SAME row partition (shard_rows) that bench.py uses with RCCL on the GPUs.
The result must equal the unsharded oracle update.
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

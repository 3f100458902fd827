"""Device memory comes back: repeated create / use / destroy cycles of the handles added this round (per-coordinate GPMM builds,
model transfer, closest-point queries, rigid ICP, registrators) leave the free device memory where it was."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize(0)
    return torch.cuda.mem_get_info(0)[0]


def test_handles_release_their_device_memory(ctx):
    import gingr_amd as ga
    from gingr_amd import classic
    from gingr_amd.simple import cluster_decimate, new_reference_nearest_neighbor
    rng = np.random.default_rng(0)
    ref = rng.normal(0, 40, (3000, 3))
    tgt = ref[rng.permutation(3000)] + rng.normal(0, 0.5, (3000, 3))

    def cycle():
        g = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.0, maxRank=40)
        m = g.GaussianSymmetry(60.0, 10.0)
        assert m.rank == 40
        dv, _ = cluster_decimate(ref, None, 500)
        small = new_reference_nearest_neighbor(ctx, m, dv)
        assert small.rank == 40
        cpd = ga.CpdRegistration(ctx)
        st = cpd.run(cpd.createInitialState(small, tgt[:800], ga.CpdConfiguration(maxIterations=4, w=0.1)))
        assert st.general.iteration == 3
        cpd.close()
        icp = classic.ICPFactory(ctx, ref).registerRigidly(tgt)
        icp.Iteration()
        icp.close()
        small._src_dev.close()
        small.device().close()
        m.device().close()

    cycle()                                   # warm-up: code objects, allocator pools
    gc.collect()
    before = _free_bytes()
    for _ in range(15):
        cycle()
    gc.collect()
    after = _free_bytes()
    assert before - after < 64 << 20, f"{(before - after) / 2**20:.1f} MiB of device memory did not come back"

#!/usr/bin/env python3
"""Per-rank cost of the REVERSED correspondence direction (ICP.scala:46-48) on a row shard, emulated on one GPU: rank 0 of `world`
ranks of the icosphere workload of tools/bench_icp_surface.py (level 6: 40 962 vertices / 81 920 triangles, template = target
topology), exchange = the library's native RCCL path with a one-rank communicator (the other ranks' contributions are missing: a
timing experiment, not a registration).  Round 4 ran the whole target -> template scan on every rank; round 5 shards it by query
range.     python tools/bench_reversed_shard.py [world] [flavour 1|2] [level]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import gingr_amd as ga  # noqa: E402
from gingr_amd.sharded import ShardedFitter  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
flavour = int(sys.argv[2]) if len(sys.argv) > 2 else 2
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sys.argv = sys.argv[:1]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("ico", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_icp_surface.py"))
src = open(spec.origin).read().split("OPTS = dict")[0]      # the icosphere generator only
ns = {"__file__": spec.origin, "__name__": "ico"}
exec(compile(src, spec.origin, "exec"), ns)
verts, cells = ns["icosphere"](level)
ref = verts * 80.0
bump = 1.0 + 0.08 * np.sin(3 * verts[:, 0]) * np.cos(2 * verts[:, 1]) + 0.05 * np.sin(5 * verts[:, 2])
c, s = np.cos(0.05), np.sin(0.05)
target = (ref * bump[:, None]) @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]).T + np.array([1.5, -1.0, 0.5])
ctx = ga.Context(0)
model = ga.GPMMTriangleMesh3D(ctx, ref, relativeTolerance=0.0, maxRank=100).Gaussian(40.0, 10.0)
ctx.rccl_init(ctx.rccl_unique_id(), 1, 0)
f = ShardedFitter(ctx, model, target, rank=0, world=world, all_reduce=None, rccl=True)
f.set_meshes(cells, cells)
f.set_correspondence_direction(True)
params = (10.0, 1.0, 100)
out = {}
for name, rev in (("reversed", True), ("forward", False)):
    f.set_correspondence_direction(rev)
    f.set_state(np.zeros(100), 10.0)
    f.update(flavour, params, 3)
    ctx.synchronize()
    n = 30
    t0 = time.perf_counter()
    f.update(flavour, params, n)
    ctx.synchronize()
    out[name + "_ms_per_iteration"] = (time.perf_counter() - t0) / n * 1e3
print(json.dumps({"what": "emulated per-rank cost, ICP on a row shard (rank 0 of %d), flavour %d" % (world, flavour), "vertices": int(ref.shape[0]),
                  "triangles": int(cells.shape[0]), "local_rows": f.end - f.begin, **out}))

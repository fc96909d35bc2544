#!/usr/bin/env python3
"""Print the cooperative search's work statistics and the pass timing on the headline scene (tuning aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sc = synthetic.make_pair(n, n, cfg_id=3)
s = sc["spacing"]
nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
dev = torch.device("cuda", 0)
d_src, d_tgt, d_nrm = (torch.from_numpy(a).to(dev) for a in (sc["source"], sc["target"], nrm))
ctx = _lib.Context(0)
for stats in (0, 1):
    ctx.debug_set(True, bool(stats))
    sess = reg.IcpSession(ctx).setup(d_src, d_tgt, d_nrm, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                     reg.ICPConvergenceCriteria(0, 0, 40))
    for _ in range(3):
        sess.enqueue_step()
    ctx.tile_stats(reset=True)
    sess.profile(True)
    for _ in range(20):
        sess.enqueue_step()
    (ms_s, ms_f), k = sess.profile_read2()
    st = ctx.tile_stats()
    print("stats=%d: search %.4f ms  fixup %.4f ms" % (stats, ms_s, ms_f), st if stats else "")
    if stats and st["wave_iters"]:
        w, v = st["wave_iters"], max(st["valid_lanes"], 1)
        print("  per wave trip: %.2f dense item trips; per query: %.2f runs, %.2f items; left to the exact fixup: %.5f" % (
            st["item_trips"] / w, st["runs"] / v, st["items"] / v, st["exact_lanes"] / v))

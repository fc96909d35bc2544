#!/usr/bin/env python3
"""Print wave-tile statistics and pass timing on the headline scene (tuning aid)."""
import sys, os, time
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
for generic in (0,):
    ctx._lib.o3d_debug_set(ctx.handle, generic, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sess = reg.IcpSession(ctx).setup(d_src, d_tgt, d_nrm, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                     reg.ICPConvergenceCriteria(0, 0, 8))
    ctx.tile_stats(reset=True)
    sess.profile(True)
    for _ in range(6):
        sess.enqueue_step()
    ms, k = sess.profile_read()
    st = ctx.tile_stats()
    print("force_fp64=%d pass_ms=%.3f" % (generic, ms), st)
    if st["wave_iters"]:
        print("  runs/lane %.2f  pending frac %.5f" % (st["candidates"] / max(st["valid_lanes"], 1), st["exact_lanes"] / max(st["valid_lanes"], 1)))
    if st["wave_iters"]:
        w = st["wave_iters"]
        print("  per wave-iteration: walk trips %.2f (lane mean %.2f), centre trips %.2f (lane mean %.2f)" % (
            st["walk_trips_wave"] / w, st["walk_trips_lane"] / max(st["valid_lanes"], 1), st["centre_trips_wave"] / w,
            st["centre_trips_lane"] / max(st["valid_lanes"], 1)))
    if False:
        print("  tiled %.3f  cand/tile %.1f  rows/tile %.1f  exact lane frac %.4f" % (
            st["tiled"] / st["wave_iters"], st["candidates"] / max(st["tiled"], 1), st["rows"] / max(st["tiled"], 1),
            st["exact_lanes"] / max(st["valid_lanes"], 1)))

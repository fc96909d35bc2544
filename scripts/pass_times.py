#!/usr/bin/env python3
"""Kernel time of every pass of one fresh 10 M x 10 M point-to-plane registration (pass 0 = the initial search), with the
search counters: how the certified share grows as the loop converges.  usage: pass_times.py [points] [iters]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open3d_amd import _lib, synthetic  # noqa: E402
from open3d_amd.pipelines import registration as reg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
sc = synthetic.make_pair(n, n, cfg_id=3)
s = sc["spacing"]
ctx = _lib.Context(0)
d_src, d_tgt = torch.from_numpy(sc["source"]).to(dev), torch.from_numpy(sc["target"]).to(dev)
d_nrm = torch.empty((n, 3), dtype=torch.float64, device=dev)
ctx.check(ctx._lib.o3d_estimate_normals(ctx.handle, d_tgt.data_ptr(), n, _lib.SEARCH_HYBRID, 30, 4.0 * s, _lib.MEM_DEVICE, d_nrm.data_ptr(), 0))
out = {}
for stats in (False, True):
    ctx.debug_set(collect_tile_stats=stats)
    if len(sys.argv) > 3 and sys.argv[3] == "p2p":
        sess = reg.IcpSession(ctx).setup(d_src, d_tgt, None, 3 * s, np.eye(4), reg.TransformationEstimationPointToPoint(),
                                         reg.ICPConvergenceCriteria(0.0, 0.0, iters))
    else:
        sess = reg.IcpSession(ctx).setup(d_src, d_tgt, d_nrm, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                         reg.ICPConvergenceCriteria(0.0, 0.0, iters))
    sess.profile(True)
    cert = []
    for k in range(iters + 1):
        sess.enqueue_step()
        if stats:
            t = ctx.tile_stats(reset=True)
            cert.append((t["certified"], t["item_trips"]))
    if stats:
        out["certified_per_pass"] = [c for c, _ in cert]
        out["item_trips_per_pass"] = [t for _, t in cert]
    else:
        out["kernel_ms_per_pass"] = [round(float(v), 4) for v in sess.profile_read_all()]
    res = sess.finish(want_corr=False)
    out["rmse"], out["fitness"] = res.inlier_rmse_, res.fitness_
print(json.dumps(out))

#!/usr/bin/env python3
"""Share of the source that k_icp_cert certifies per pass (the rest goes to the search) over the first passes of the headline
scene, for both estimations.  usage: cert_fraction.py [points] [passes]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from open3d_amd import _lib, synthetic  # noqa: E402
from open3d_amd.pipelines import registration as reg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = _lib.Context(0)
sc = synthetic.make_pair(n, n, cfg_id=3)
s = sc["spacing"]
nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
for name, est, r, nr in (("p2plane", reg.TransformationEstimationPointToPlane(), 2 * s, nrm),
                         ("p2p", reg.TransformationEstimationPointToPoint(), 3 * s, None)):
    ctx.debug_set(collect_tile_stats=True)
    sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nr, r, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, 100000))
    ctx.tile_stats()
    row = []
    for k in range(passes):
        sess.enqueue_step()
        torch.cuda.synchronize()
        st = ctx.tile_stats()
        row.append(round(st["certified"] / n, 4))
    sess.finish(want_corr=False)
    print(json.dumps({"estimation": name, "points": n, "certified_share_per_pass": row}), flush=True)
ctx.debug_set(collect_tile_stats=False)

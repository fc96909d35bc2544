#!/usr/bin/env python3
"""Time the search-kernel variants (debug bits 4..5 of the stats flag) on the headline scene (tuning aid)."""
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
for v in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3").split(",")]:
    ctx._lib.o3d_debug_set(ctx.handle, 0, v << 4)
    sess = reg.IcpSession(ctx).setup(d_src, d_tgt, d_nrm, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                     reg.ICPConvergenceCriteria(0, 0, 40))
    for _ in range(4):
        sess.enqueue_step()
    sess.profile(True)
    for _ in range(30):
        sess.enqueue_step()
    (ms_s, ms_a), k = sess.profile_read2()
    print("variant %d: search %.4f ms  accumulate+fixup %.4f ms" % (v, ms_s, ms_a), flush=True)

#!/usr/bin/env python3
"""Kernel time of the first passes of a registration from scratch (10 M x 10 M, device-generated config-3 scene) for the
search radii 2 and 4 spacings (point-to-plane) and 3 spacings (point-to-point), with a checksum of the final correspondences
(library variants must agree).  AB_ONLY=p2l_r2[,...] runs the named scenes once.  usage: search_ab.py [tag]"""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg

tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = int(os.environ.get("AB_POINTS", "10000000"))
ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
sc = synthetic.make_pair_device(ctx, n, n, cfg_id=3)
s = sc["spacing"]
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
out = {"tag": tag}
for name, rs, est, nrm in (("p2l_r2", 2.0, reg.TransformationEstimationPointToPlane(), sc["normals"]),
                           ("p2l_r4", 4.0, reg.TransformationEstimationPointToPlane(), sc["normals"]),
                           ("p2p_r3", 3.0, reg.TransformationEstimationPointToPoint(), None)):
    if os.environ.get("AB_ONLY") and name not in os.environ["AB_ONLY"].split(","):
        continue
    best = None
    for rep in range(1 if os.environ.get("AB_ONLY") else 3):
        sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nrm, rs * s, np.eye(4), est, reg.ICPConvergenceCriteria(0, 0, 6))
        sess.profile(True)
        for _ in range(7):
            sess.enqueue_step()
        ms = sess.profile_read_all()
        sess.profile(False)
        best = ms if best is None else np.minimum(best, ms)
    corr = torch.empty(n, dtype=torch.int32, device=dev)
    try:
        res = sess.finish(corr_out=corr)
        crc, rmse = zlib.crc32(corr.cpu().numpy().tobytes()), res.inlier_rmse_
    except _lib.O3DError as e:  # (timing experiments with deliberately wrong kernels end here)
        crc, rmse = -1, str(e)[:60]
    out[name] = {"ms": [round(float(v), 4) for v in best], "crc": crc, "rmse": rmse,
                 "setup_ms": [round(float(v), 3) for v in sess.setup_ms()]}
print(json.dumps(out), flush=True)

#!/usr/bin/env python3
"""Writes the match positions of a converged headline registration (10 M x 10 M point-to-plane, r = 2 spacings, 10 iterations),
in processing order, to the file named: the addresses the certifying pass gathers its 48-byte records from, for
scripts/micro/cert_pattern.  usage: real_positions.py out.bin"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg

n = int(os.environ.get("AB_POINTS", "10000000"))
ctx = _lib.Context(0)
sc = synthetic.make_pair_device(ctx, n, n, cfg_id=3)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], sc["normals"], 2.0 * sc["spacing"], np.eye(4),
                                 reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0, 0, 10))
for _ in range(11):
    sess.enqueue_step()
res = sess.finish()
pos = ctx.match_positions(n)
pos.tofile(sys.argv[1])
d = np.abs(np.diff(pos.astype(np.int64)))
blk = pos[: n // 64 * 64].reshape(-1, 64).astype(np.int64)
lines = np.array([len(np.unique((b * 48) // 128)) for b in blk[:: max(1, len(blk) // 2000)]])
print("fitness", res.fitness_, "median |step| between neighbours", float(np.median(d)), "mean 128-byte lines holding the first piece of a block's 64 records",
      float(lines.mean()), "span of a block (records, median)", float(np.median(blk.max(1) - blk.min(1))), flush=True)

"""Per pass of a registration from scratch (10 M x 10 M, config-3 scene): the fraction of source points the update leaves
where they are bit for bit, and the fraction of 64-point blocks without a moving point.  Needs a library whose icp.hip was
built with -DO3D_COUNT_STATIC=1 (counter 7 of the search statistics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg
ctx = _lib.Context(0)
n = 10_000_000
sc = synthetic.make_pair_device(ctx, n, n, cfg_id=3)
s = sc["spacing"]
ctx.debug_set(use_tile_search=True, collect_tile_stats=True)
sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], sc["normals"], 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0, 0, 20))
ctx.tile_stats(reset=True)
out = []
for _ in range(21):
    sess.enqueue_step()
    torch.cuda.synchronize()
    v = ctx.tile_stats(reset=True)["_7"]
    out.append((round((v & ((1 << 40) - 1)) / n, 4), round((v >> 40) / (n / 64), 4)))
print("per pass: (fraction of points that did not move bitwise, fraction of wavefront blocks with no moving point)")
print(out)

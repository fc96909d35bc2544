#!/usr/bin/env python3
"""Config 3 pipeline from host buffers with the ICP call split into its phases (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open3d_amd import _lib, geometry, synthetic
from open3d_amd.pipelines import registration as reg
ctx = _lib.Context(0)
sc = synthetic.make_pair(10_000_000, 10_000_000, cfg_id=3)
s = sc["spacing"]
v = 2 * s
src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
est = reg.TransformationEstimationPointToPlane()
for rep in range(5):
    t0 = time.perf_counter()
    sv = src.VoxelDownSample(v, ctx=ctx)
    ta = time.perf_counter()
    tv = tgt.VoxelDownSample(v, ctx=ctx)
    t1 = time.perf_counter()
    tv.EstimateNormals(geometry.KDTreeSearchParamHybrid(2 * v, 30), ctx=ctx)
    t2 = time.perf_counter()
    sess = reg.IcpSession(ctx).setup(sv.points_, tv.points_, tv.normals_, 2 * v, np.eye(4), est, reg.ICPConvergenceCriteria())
    t3 = time.perf_counter()
    sess.run()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    res = sess.finish()
    t5 = time.perf_counter()
    print("rep %d: voxel %.1f + %.1f | normals %.1f | icp setup %.2f (device: grid %.2f, source %.2f) loop %.2f finish %.2f | total %.1f ms"
          % (rep, 1e3 * (ta - t0), 1e3 * (t1 - ta), 1e3 * (t2 - t1), 1e3 * (t3 - t2), *sess.setup_ms(), 1e3 * (t4 - t3),
             1e3 * (t5 - t4), 1e3 * (t5 - t0)), flush=True)

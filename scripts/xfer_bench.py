#!/usr/bin/env python3
"""Host-buffer call timings (PCIe included): VoxelDownSample and EstimateNormals on 10M points, repeated (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_amd import _lib, geometry, synthetic
ctx = _lib.Context(0)
sc = synthetic.make_pair(1000, 10_000_000, cfg_id=3)
pc = geometry.PointCloud(sc["target"])
s = sc["spacing"]
for rep in range(3):
    t0 = time.perf_counter(); v = pc.VoxelDownSample(2 * s, ctx=ctx); t1 = time.perf_counter()
    pc.EstimateNormals(geometry.KDTreeSearchParamHybrid(4 * s, 30), ctx=ctx); t2 = time.perf_counter()
    print("rep %d: voxel %.1f ms (10M -> %d), normals %.1f ms" % (rep, 1e3 * (t1 - t0), len(v.points_), 1e3 * (t2 - t1)), flush=True)
    pc.normals_ = np.zeros((0, 3))

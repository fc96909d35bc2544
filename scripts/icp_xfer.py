import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_amd import _lib, geometry, synthetic
from open3d_amd.pipelines import registration as reg
ctx = _lib.Context(0)
sc = synthetic.make_pair(3_000_000, 3_000_000, cfg_id=3)
s = sc["spacing"]
nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
for rep in range(4):
    t0 = time.perf_counter()
    res = reg.RegistrationICP(src, tgt, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0, 0, 2), ctx=ctx)
    print("rep %d: %.1f ms" % (rep, 1e3 * (time.perf_counter() - t0)), flush=True)

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
# phases of the same call through the session API
est = reg.TransformationEstimationPointToPlane()
for rep in range(3):
    t0 = time.perf_counter()
    sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nrm, 2 * s, np.eye(4), est, reg.ICPConvergenceCriteria(0, 0, 2))
    t1 = time.perf_counter()
    sess.run()
    import torch
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res = sess.finish()
    t3 = time.perf_counter()
    print("session rep %d: setup %.2f ms (device: grid %.2f, source %.2f) | loop %.2f ms | finish %.2f ms"
          % (rep, 1e3 * (t1 - t0), *sess.setup_ms(), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
res = reg.RegistrationICP(src, tgt, 2 * s, np.eye(4), est, reg.ICPConvergenceCriteria(0, 0, 2), ctx=ctx)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

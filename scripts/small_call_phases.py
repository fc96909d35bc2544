#!/usr/bin/env python3
"""Where a small RegistrationICP call (config 1: 100k x 100k, host buffers) spends its time (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg
ctx = _lib.Context(0)
sc = synthetic.make_pair(100_000, 100_000, cfg_id=1)
s = sc["spacing"]
est, crit = reg.TransformationEstimationPointToPoint(), reg.ICPConvergenceCriteria(1e-6, 1e-6, 30)
acc = np.zeros(4)
for rep in range(12):
    t0 = time.perf_counter()
    sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], None, 3 * s, np.eye(4), est, crit)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sess.run()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = sess.finish()
    t3 = time.perf_counter()
    if rep >= 2:
        acc += [t1 - t0, t2 - t1, t3 - t2, t3 - t0]
print("setup %.3f ms | loop (%d iterations) %.3f ms | finish %.3f ms | total %.3f ms; device-side setup: grid %.3f ms, source %.3f ms"
      % (*(1e3 * acc[:1] / 10), res.iterations_, *(1e3 * acc[1:] / 10), *sess.setup_ms()))

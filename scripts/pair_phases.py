#!/usr/bin/env python3
"""Where one pair of a batch (config 5: 200 k x 200 k points, point-to-plane, 30 iterations, device buffers) spends its time
when it has the GPU to itself: whole call, set-up alone (host-synchronised), the loop, the results."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_amd import _lib, synthetic  # noqa: E402
from open3d_amd.pipelines import registration as reg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
sc = synthetic.make_pair(n, n, cfg_id=5, L=synthetic.extent_for(n))
s = sc["spacing"]
d_src, d_tgt = torch.from_numpy(sc["source"]).to(dev), torch.from_numpy(sc["target"]).to(dev)
d_nrm = torch.from_numpy(synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])).to(dev)
T = np.zeros(16)
nc, fit, rmse, it = C.c_int64(), C.c_double(), C.c_double(), C.c_int()
eye = _lib.colmajor(np.eye(4)).ctypes.data_as(_lib.c_dp)


def whole(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.check(ctx._lib.o3d_registration_icp(ctx.handle, _lib.ptr(d_src), n, _lib.ptr(d_tgt), _lib.ptr(d_nrm), n, 2 * s, eye, 2, 0, 0.0, 0.0,
                                            iters, _lib.MEM_DEVICE, T.ctypes.data_as(_lib.c_dp), None, C.byref(nc), C.byref(fit),
                                            C.byref(rmse), C.byref(it)))
    return 1e3 * (time.perf_counter() - t0)


for _ in range(3):
    whole(30)
row = {"points": n, "whole_call_30_iterations_ms": round(min(whole(30) for _ in range(10)), 4),
       "whole_call_1_iteration_ms": round(min(whole(1) for _ in range(10)), 4)}
est, crit = reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0.0, 0.0, 30)
best = {}
for rep in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sess = reg.IcpSession(ctx).setup(d_src, d_tgt, d_nrm, 2 * s, np.eye(4), est, crit)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(31):
        sess.enqueue_step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    sess.finish(want_corr=False)
    t5 = time.perf_counter()
    for k, v in (("setup_host_ms", t1 - t0), ("setup_drain_ms", t2 - t1), ("enqueue_31_passes_host_ms", t3 - t2), ("loop_drain_ms", t4 - t3),
                 ("finish_ms", t5 - t4)):
        best[k] = min(best.get(k, 1e9), 1e3 * v)
row.update({k: round(v, 4) for k, v in best.items()})
print(json.dumps(row))

#!/usr/bin/env python3
"""BASELINE config 5 in full on one GPU: 4096 independent pairs of 200 k x 200 k points, point-to-plane,
ICPConvergenceCriteria(0, 0, 30), clouds and analytic target normals generated on the device (o3d_synthetic_pair: 59 GB that
never cross PCIe), registered by o3d_registration_icp_batch.  Every pair's transformation is checked against the motion its
scene was made with; a sample is registered again one by one and must give the same result to rounding (1e-12).
usage: config5_full.py [pairs] [workers ...]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_amd import _lib, synthetic  # noqa: E402

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
workers = [int(w) for w in sys.argv[2:]] or [8]
n = int(os.environ.get("C5_POINTS", "200000"))   # (other sizes: a tuning experiment, not config 5)
ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
src = torch.empty((npairs, n, 3), dtype=torch.float64, device=dev)
tgt = torch.empty((npairs, n, 3), dtype=torch.float64, device=dev)
nrm = torch.empty((npairs, n, 3), dtype=torch.float64, device=dev)
truth = []
for k in range(npairs):
    seed = (int(0x9E3779B97F4A7C15) ^ (100000 + k)) & 0xFFFFFFFFFFFFFFFF
    L = 2.83
    s = L / np.sqrt(float(n))
    T_true = synthetic.rigid_transform(0.05 + 0.0001 * (k % 50), np.array([1.0, -0.7, 0.5]) * s)
    truth.append(T_true)
    ctx.check(ctx._lib.o3d_synthetic_pair(ctx.handle, n, n, seed, L, 0.1 * s, _lib.colmajor(np.linalg.inv(T_true)).ctypes.data_as(_lib.c_dp),
                                          src[k].data_ptr(), tgt[k].data_ptr(), nrm[k].data_ptr()))
torch.cuda.synchronize()
gen_s = time.perf_counter() - t0
VP, I64 = C.c_void_p * npairs, C.c_int64 * npairs
a_src, a_tgt, a_nrm = VP(*[src[k].data_ptr() for k in range(npairs)]), VP(*[tgt[k].data_ptr() for k in range(npairs)]), VP(*[nrm[k].data_ptr() for k in range(npairs)])
cnt = I64(*[n] * npairs)
T = np.zeros((npairs, 16)); nc = np.zeros(npairs, np.int64); fit = np.zeros(npairs); rmse = np.zeros(npairs)
iters = np.zeros(npairs, np.int32); status = np.zeros(npairs, np.int32)
out = {"config": 5, "pairs": npairs, "points_per_cloud": n, "generate_s": gen_s, "device_bytes": int(3 * src.numel() * 8)}
for w in workers:
    t0 = time.perf_counter()
    ctx.check(ctx._lib.o3d_registration_icp_batch(ctx.handle, npairs, a_src, cnt, a_tgt, a_nrm, cnt, 2 * s, None, _lib.EST_POINT_TO_PLANE, 0, 0.0,
                                                  0.0, 30, _lib.MEM_DEVICE, w, _lib.ptr(T), None, _lib.ptr(nc), _lib.ptr(fit), _lib.ptr(rmse),
                                                  _lib.ptr(iters), _lib.ptr(status)))
    dt = time.perf_counter() - t0
    out["pairs_per_s_workers_%d" % w] = npairs / dt
    out["seconds_workers_%d" % w] = dt
err = max(float(np.abs(_lib.from_colmajor(T[k]) - truth[k]).max()) for k in range(npairs))
out["status_nonzero"] = int(np.count_nonzero(status))
out["iterations_min_max"] = [int(iters.min()), int(iters.max())]
out["fitness_min"] = float(fit.min())
out["max_abs_T_minus_truth"] = err
# a sample again, one by one: same bits
same = 0
sample = list(range(0, npairs, max(1, npairs // 16)))
for k in sample:
    T1 = np.zeros(16)
    nc1, f1, r1, it1 = C.c_int64(), C.c_double(), C.c_double(), C.c_int()
    ctx.check(ctx._lib.o3d_registration_icp(ctx.handle, src[k].data_ptr(), n, tgt[k].data_ptr(), nrm[k].data_ptr(), n, 2 * s,
                                            _lib.colmajor(np.eye(4)).ctypes.data_as(_lib.c_dp), 2, 0, 0.0, 0.0, 30, _lib.MEM_DEVICE,
                                            T1.ctypes.data_as(_lib.c_dp), None, C.byref(nc1), C.byref(f1), C.byref(r1), C.byref(it1)))
    # (a pair of a gang adds its sums over fewer, longer workgroups than its single call: equal to rounding)
    same += int(np.abs(T1 - T[k]).max() < 1e-12 and nc1.value == nc[k] and abs(r1.value - rmse[k]) < 1e-14)
out["sample_single_calls_equal_to_rounding"] = "%d of %d" % (same, len(sample))
print(json.dumps(out))

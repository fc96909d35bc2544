#!/usr/bin/env python3
"""Config 5 slice (64 pairs x 200k points, P2Plane, 30 iterations) with every buffer already in HBM: what the batch costs
once PCIe is out of the way (tuning aid; the host-buffer figure is scripts/bench_configs.py 5)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open3d_amd import _lib, synthetic
ctx = _lib.Context(0)
npairs, n = 64, 200_000
dev = torch.device("cuda", 0)
S, T, N = [], [], []
for k in range(npairs):
    sc = synthetic.make_pair(n, n, cfg_id=1000 + k, L=2.83)
    s = sc["spacing"]
    nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
    S.append(torch.from_numpy(sc["source"]).to(dev)); T.append(torch.from_numpy(sc["target"]).to(dev)); N.append(torch.from_numpy(nrm).to(dev))
torch.cuda.synchronize()
VP, I64 = C.c_void_p * npairs, C.c_int64 * npairs
a_src, a_tgt, a_nrm = VP(*[x.data_ptr() for x in S]), VP(*[x.data_ptr() for x in T]), VP(*[x.data_ptr() for x in N])
n_src, n_tgt = I64(*[n] * npairs), I64(*[n] * npairs)
Tm = np.zeros((npairs, 16)); nc = np.zeros(npairs, np.int64); fit = np.zeros(npairs); rmse = np.zeros(npairs)
iters = np.zeros(npairs, np.int32); status = np.zeros(npairs, np.int32)
out = {}
for w in (1, 2, 4, 8):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.check(ctx._lib.o3d_registration_icp_batch(ctx.handle, npairs, a_src, n_src, a_tgt, a_nrm, n_tgt, 2 * s, None, _lib.EST_POINT_TO_PLANE,
                                                      0, 0.0, 0.0, 30, _lib.MEM_DEVICE, w, _lib.ptr(Tm), None, _lib.ptr(nc), _lib.ptr(fit),
                                                      _lib.ptr(rmse), _lib.ptr(iters), _lib.ptr(status)))
        best = min(best, time.perf_counter() - t0)
    out["pairs_per_s_workers_%d" % w] = npairs / best
print(json.dumps({"config": "5 (device resident)", **out, "iterations": int(iters[0]), "fitness": float(fit[0])}))

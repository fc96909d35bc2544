#!/usr/bin/env python3
"""The sharded RegistrationICP of a multi context (o3d_ctx_create_multi) rehearsed on ONE GPU: device 0 listed G times.
Reports milliseconds per iteration of the whole call's loop for G = 1, 2, 4, 8 at several cloud sizes.  With every rank on
the same GPU the kernels of the ranks share it, so large clouds show no speed-up here; what the numbers show is the fixed
cost per iteration of the exchange path (small clouds) and that nothing is lost at large ones."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open3d_amd import _lib, synthetic  # noqa: E402

dev = torch.device("cuda", 0)
iters = 30
for n in (20_000, 1_250_000, 10_000_000):
    sc = synthetic.make_pair(n, n, cfg_id=3, L=synthetic.extent_for(n))
    s = sc["spacing"]
    d_src, d_tgt = torch.from_numpy(sc["source"]).to(dev), torch.from_numpy(sc["target"]).to(dev)
    d_nrm = torch.from_numpy(synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])).to(dev)
    row = {"points": n}
    for G in (1, 2, 4, 8):
        ctx = _lib.Context([0] * G if G > 1 else 0)
        T = np.zeros(16)
        nc, fit, rmse, it = C.c_int64(), C.c_double(), C.c_double(), C.c_int()
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.check(ctx._lib.o3d_registration_icp(ctx.handle, _lib.ptr(d_src), n, _lib.ptr(d_tgt), _lib.ptr(d_nrm), n, 2 * s,
                                                    _lib.colmajor(np.eye(4)).ctypes.data_as(_lib.c_dp), 2, 0, 0.0, 0.0, iters, _lib.MEM_DEVICE,
                                                    T.ctypes.data_as(_lib.c_dp), None, C.byref(nc), C.byref(fit), C.byref(rmse), C.byref(it)))
            dt = time.perf_counter() - t0
            # the same call with one iteration: set-up + two passes; the difference is 29 iterations of the loop
            t0 = time.perf_counter()
            ctx.check(ctx._lib.o3d_registration_icp(ctx.handle, _lib.ptr(d_src), n, _lib.ptr(d_tgt), _lib.ptr(d_nrm), n, 2 * s,
                                                    _lib.colmajor(np.eye(4)).ctypes.data_as(_lib.c_dp), 2, 0, 0.0, 0.0, 1, _lib.MEM_DEVICE,
                                                    T.ctypes.data_as(_lib.c_dp), None, C.byref(nc), C.byref(fit), C.byref(rmse), C.byref(it)))
            d1 = time.perf_counter() - t0
            per = 1e3 * (dt - d1) / (iters - 1)
            best = per if best is None else min(best, per)
        row["G%d_ms_per_iteration" % G] = round(best, 4)
        ctx.close()
    print(json.dumps(row), flush=True)

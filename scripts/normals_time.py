#!/usr/bin/env python3
"""EstimateNormals on a device-resident cloud of the benchmark surface: milliseconds per call for the reference's default
search (KNN 30), the hybrid search of configs 2 / 3 (radius 4 spacings, 30) and a radius search; O3D_NORMALS_HEAP=1 selects
the per-lane heap kernel for comparison.  One JSON line per size."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_amd import _lib, synthetic  # noqa: E402

ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
only = os.environ.get("NRM_ONLY")  # e.g. hybrid_4s_30: one search only (for counter passes)
for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]:
    sc = synthetic.make_pair(n, n, cfg_id=3, L=synthetic.extent_for(n))
    s = sc["spacing"]
    d_xyz = torch.from_numpy(sc["target"]).to(dev)
    d_nrm = torch.empty_like(d_xyz)
    row = {"points": n, "heap_kernel_forced": bool(os.environ.get("O3D_NORMALS_HEAP"))}
    for name, st, knn, radius in (("knn30", 0, 30, 0.0), ("hybrid_4s_30", 2, 30, 4 * s), ("knn12", 0, 12, 0.0)):
        if only and name != only:
            continue
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.check(ctx._lib.o3d_estimate_normals(ctx.handle, _lib.ptr(d_xyz), n, st, knn, radius, _lib.MEM_DEVICE, _lib.ptr(d_nrm), 0))
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0))
        row[name + "_ms"] = round(best, 3)
        row[name + "_grid"] = {k: v for k, v in ctx.grid_info().items() if k in ("cell_side", "nx", "ny", "nz")}
        row[name + "_mean_abs_nz"] = float(d_nrm[:, 2].abs().mean())
    print(json.dumps(row), flush=True)

#!/usr/bin/env python3
"""The hard scene of bench.py (10 M x 10 M point-to-plane, rotation 1 degree, translation 6.1 spacings, r = 4 spacings), pass by
pass: the search kernel's counters (wave trips, dense item rounds, runs listed, exact fp64 resolutions, items, valid lanes,
certified) next to the kernel time, and the fitness / update size that go with them.  usage: hard_scene_stats.py [passes]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg

n = int(os.environ.get("AB_POINTS", "10000000"))
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
base = synthetic.make_pair_device(ctx, 1000, 1000, cfg_id=3)
hs = synthetic.make_pair_device(ctx, n, n, cfg_id=33, theta_deg=1.0, tau_in_spacings=(4.0, -3.5, 2.9), want_normals=False)
s = hs["spacing"]
nrm = torch.empty((n, 3), dtype=torch.float64, device=dev)
ctx.check(ctx._lib.o3d_estimate_normals(ctx.handle, hs["target"].data_ptr(), n, _lib.SEARCH_HYBRID, 30, 4.0 * s, _lib.MEM_DEVICE, nrm.data_ptr(), 0))
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
est, crit = reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0, 0, passes)
sess = reg.IcpSession(ctx).setup(hs["source"], hs["target"], nrm, 4.0 * s, np.eye(4), est, crit)
sess.profile(True)
for _ in range(passes + 1):
    sess.enqueue_step()
ms = sess.profile_read_all()
sess.profile(False)
sess.finish()
ctx.debug_set(use_tile_search=True, collect_tile_stats=True)
sess = reg.IcpSession(ctx).setup(hs["source"], hs["target"], nrm, 4.0 * s, np.eye(4), est, crit)
ctx.tile_stats(reset=True)
for k in range(passes + 1):
    sess.enqueue_step()
    torch.cuda.synchronize()
    st = ctx.tile_stats(reset=True)
    print(json.dumps({"pass": k, "kernel_ms": round(float(ms[k]), 4), **{a: b for a, b in st.items() if not a.startswith("_")}}), flush=True)
res = sess.finish()
log = sess.log()
print("fitness per pass", [round(float(v), 5) for v in log[:, 0]], "rmse / spacing", [round(float(v) / s, 4) for v in log[:, 1]])
ctx.debug_set(use_tile_search=True, collect_tile_stats=False)

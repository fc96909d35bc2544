#!/usr/bin/env python3
"""A sparse VOLUMETRIC scene against the dense cell table: spheres scattered through a cube, so that cells of side r over
the bounding box exceed the table's caps (2^31 cells, 64 cells per point) and the cell side is coarsened (DESIGN.md,
"Sparse scenes").  Reports the loop's iterations/s and the grid that was built, beside the height field of the same size and
the same radius in spacings; correspondences of the first iterations are checked against the oracle on every point.
usage: sparse_scene.py [points] [cube side in metres]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker only)
from open3d_amd import _lib, synthetic  # noqa: E402
from open3d_amd.pipelines import registration as reg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
side = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rng = np.random.default_rng(7)
nsph, rad = 40, 3.0
centres = rng.uniform(rad, side - rad, (nsph, 3))
v = rng.normal(0, 1, (n, 3))
v /= np.linalg.norm(v, axis=1, keepdims=True)
which = rng.integers(0, nsph, n)
target = centres[which] + rad * v
normals = v.copy()
area = nsph * 4 * np.pi * rad * rad
s = float(np.sqrt(area / n))
T_true = synthetic.rigid_transform(0.02, np.array([1.0, -0.7, 0.5]) * s)
Ti = np.linalg.inv(T_true)
source = (target + rng.normal(0, 0.1 * s, target.shape)) @ Ti[:3, :3].T + Ti[:3, 3]
r = 2.0 * s
ctx = _lib.Context(0)
oracle.build()
est = reg.TransformationEstimationPointToPlane()
out = {"points": n, "cube_side_m": side, "spacing_m": s, "radius_m": r, "cells_of_side_r_over_the_box": float((side / r) ** 3)}


def loop_rate(src, tgt, nrm, tag):
    src, tgt, nrm = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (src, tgt, nrm))   # resident: no PCIe in the set-up times
    torch.cuda.synchronize()
    sess = reg.IcpSession(ctx).setup(src, tgt, nrm, r, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, 100000))
    out[tag + "_first_setup_ms_grid_sort"] = [round(float(x), 3) for x in sess.setup_ms()]   # includes growing the buffers
    sess.finish(want_corr=False)
    sess = reg.IcpSession(ctx).setup(src, tgt, nrm, r, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, 100000))
    ms = sess.setup_ms()
    best = 0.0
    for rep in range(4):
        for _ in range(10):
            sess.enqueue_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            sess.enqueue_step()
        torch.cuda.synchronize()
        best = max(best, 100 / (time.perf_counter() - t0))
    res = sess.finish(want_corr=False)
    out[tag + "_iters_per_s"] = round(best, 1)
    out[tag + "_setup_ms_grid_sort"] = [round(float(x), 3) for x in np.atleast_1d(ms)]
    return res


res = loop_rate(source, target, normals, "spheres")
out["spheres_T_err"] = float(np.abs(res.transformation_ - T_true).max())
from open3d_amd import geometry  # noqa: E402
g = reg.RegistrationICP(geometry.PointCloud(source), geometry.PointCloud(target, normals=normals), r, np.eye(4), est,
                        reg.ICPConvergenceCriteria(0.0, 0.0, 3), ctx=ctx)
ref = oracle.registration_icp(source, target, r, None, 2, normals, rel_fitness=0.0, rel_rmse=0.0, max_iter=3)
out["spheres_corr_differ_from_oracle"] = int(np.count_nonzero(g.corr_ != ref.corr))
out["spheres_T_rel_err_vs_oracle"] = float(np.abs(g.transformation_ - ref.transformation).max() / np.abs(ref.transformation).max())
out["spheres_grid"] = ctx.grid_info()
hf = synthetic.make_pair(n, n, cfg_id=9, L=float(np.sqrt(n)) * s)   # height field of the same spacing
hn = synthetic.surface_normals(hf["target"][:, 0], hf["target"][:, 1])
r = 2.0 * hf["spacing"]
loop_rate(hf["source"], hf["target"], hn, "height_field")
out["height_field_grid"] = ctx.grid_info()
print(json.dumps(out))

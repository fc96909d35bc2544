#!/usr/bin/env python3
"""O3D_TAIL_CHECK=2 python3 scripts/tail_check_demo.py: a few registrations (single calls and a lock-step batch) with the
tail check on; the library prints one line per checked pass on stderr."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_amd import _lib, geometry, synthetic  # noqa: E402
from open3d_amd.pipelines import registration as reg  # noqa: E402

ctx = _lib.Context(0)
n = 30000
sc = synthetic.make_pair(n, n, cfg_id=0, L=6.0)
s = sc["spacing"]
nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
for k in (3, 2):
    got = reg.RegistrationICP(src, tgt, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0.0, 0.0, k), ctx=ctx)
    print("single", k, got.fitness_)
many = reg.RegistrationICPBatch([src] * 4, [tgt] * 4, 2 * s, None, reg.TransformationEstimationPointToPlane(),
                                reg.ICPConvergenceCriteria(0.0, 0.0, 3), n_workers=1, ctx=ctx)
print("batch", [m.fitness_ for m in many])

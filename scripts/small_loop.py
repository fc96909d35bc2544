#!/usr/bin/env python3
"""Iterations/s of the device-resident loop at small and medium sizes (configs 1, 5 and 2: 100 k, 200 k, 1 M points), both
estimations.  Environment switches (O3D_NO_TAIL, O3D_NO_CERT) select the variant; one JSON line per size."""
import json
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from open3d_amd import _lib, synthetic  # noqa: E402
from open3d_amd.pipelines import registration as reg  # noqa: E402

ctx = _lib.Context(0)
tag = ",".join(k for k in ("O3D_NO_TAIL", "O3D_NO_CERT") if os.environ.get(k)) or "default"
for n in [int(a) for a in sys.argv[1:]] or [100_000, 200_000, 1_000_000]:
    sc = synthetic.make_pair(n, n, cfg_id=1, L=synthetic.extent_for(n))
    s = sc["spacing"]
    nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
    row = {"points": n, "variant": tag}
    for name, est, r, nr in (("p2p", reg.TransformationEstimationPointToPoint(), 3 * s, None),
                             ("p2plane", reg.TransformationEstimationPointToPlane(), 2 * s, nrm)):
        sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nr, r, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, 100000))
        best = 0.0
        for rep in range(5):
            for _ in range(20):
                sess.enqueue_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(400):
                sess.enqueue_step()
            torch.cuda.synchronize()
            best = max(best, 400 / (time.perf_counter() - t0))
        row[name + "_iters_per_s"] = round(best)
        row[name + "_us_per_iteration"] = round(1e6 / best, 2)
        res = sess.finish(want_corr=False)
        row[name + "_err_vs_truth"] = float(np.abs(res.transformation_ - sc["T_true"]).max())
    print(json.dumps(row), flush=True)

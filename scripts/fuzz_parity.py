#!/usr/bin/env python3
"""Randomised parity soak (not a pytest: run by hand on a GPU box): random small registrations -- sizes, extents, radii,
initial poses, both estimations, scaling, non-finite points, duplicates -- through the single call and, in groups, through
the batch, against the oracle: correspondences bit for bit, transformation / fitness / rmse within 1e-6.
usage: fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from open3d_amd import _lib, geometry, synthetic  # noqa: E402
from open3d_amd.pipelines import registration as reg  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _lib.Context(0)
t_end = time.time() + budget
cases = batches = 0


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def make_case():
    ns, nt = int(rng.integers(40, 20000)), int(rng.integers(40, 20000))
    L = float(10 ** rng.uniform(-1, 2))
    sc = synthetic.make_pair(ns, nt, cfg_id=int(rng.integers(1 << 30)), L=L, theta_deg=float(rng.uniform(0, 2)),
                             noise_in_spacings=float(rng.uniform(0, 0.3)))
    src, tgt = sc["source"].copy(), sc["target"].copy()
    if rng.random() < 0.3:
        src[rng.integers(ns, size=3)] = np.nan
        tgt[rng.integers(nt, size=2), rng.integers(3)] = np.inf
    if rng.random() < 0.3:
        tgt[rng.integers(nt, size=5)] = tgt[rng.integers(nt, size=5)]  # duplicates: ties to the lowest index
    return src, tgt, sc["spacing"]


while time.time() < t_end:
    plane = rng.random() < 0.5
    scaling = (not plane) and rng.random() < 0.3
    est = reg.TransformationEstimationPointToPlane() if plane else reg.TransformationEstimationPointToPoint(scaling)
    crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, int(rng.integers(1, 12))) if rng.random() < 0.5 else \
        reg.ICPConvergenceCriteria(0.0, 0.0, int(rng.integers(1, 8)))
    group = int(rng.integers(1, 6))
    pairs = [make_case() for _ in range(group)]
    r = float(rng.uniform(1.0, 4.0)) * max(p[2] for p in pairs)
    srcs, tgts, refs = [], [], []
    for src, tgt, s in pairs:
        fin = np.isfinite(tgt).all(axis=1)
        nrm = np.zeros_like(tgt)
        nrm[:, 2] = 1.0
        if plane:
            nrm[fin] = oracle.estimate_normals(tgt[fin], oracle.HYBRID, 30, 4 * s)
        init = np.eye(4)
        if rng.random() < 0.4:
            init[:3, 3] = rng.normal(size=3) * 0.5 * s
        ref = oracle.registration_icp(src, tgt, r, init, 2 if plane else 1, nrm if plane else None, with_scaling=scaling,
                                      rel_fitness=crit.relative_fitness_, rel_rmse=crit.relative_rmse_, max_iter=crit.max_iteration_)
        pc_s, pc_t = geometry.PointCloud(src), geometry.PointCloud(tgt, normals=nrm if plane else None)
        got = reg.RegistrationICP(pc_s, pc_t, r, init, est, crit, ctx=ctx)
        assert got.iterations_ == ref.iterations, (cases, got.iterations_, ref.iterations)
        assert np.array_equal(got.corr_, ref.corr), (cases, "correspondences differ", int((got.corr_ != ref.corr).sum()))
        assert rel(got.transformation_, ref.transformation) < 1e-6 and abs(got.fitness_ - ref.fitness) < 1e-12
        assert abs(got.inlier_rmse_ - ref.inlier_rmse) <= 1e-6 * max(ref.inlier_rmse, 1e-300)
        srcs.append(pc_s); tgts.append(pc_t); refs.append((ref, init))
        cases += 1
    if group > 1:
        res = reg.RegistrationICPBatch(srcs, tgts, r, [i for _, i in refs], est, crit, n_workers=int(rng.integers(1, 4)), want_corr=True,
                                       ctx=ctx)
        for g, (ref, _) in zip(res, refs):
            assert g.iterations_ == ref.iterations and np.array_equal(g.corr_, ref.corr)
            assert rel(g.transformation_, ref.transformation) < 1e-6
        batches += 1
print("fuzz ok: %d registrations and %d batches against the oracle" % (cases, batches))

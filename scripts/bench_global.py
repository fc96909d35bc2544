#!/usr/bin/env python3
"""Timings of the global-registration pieces on one GPU, host buffers: EstimateNormals, ComputeFPFHFeature,
feature matching, RegistrationRANSACBasedOnFeatureMatching.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_amd import _lib, geometry, synthetic
from open3d_amd.pipelines import registration as reg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ctx = _lib.Context(0)
sc = synthetic.make_pair(n, n, cfg_id=77, theta_deg=25.0, tau_in_spacings=(40.0, -25.0, 10.0), noise_in_spacings=0.02, L=6.0 * (n / 6000) ** 0.5)
s = sc["spacing"]
ps, pt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])


def timed(fn):
    fn()
    t0 = time.perf_counter()
    out = fn()
    return 1e3 * (time.perf_counter() - t0), out


t_n, _ = timed(lambda: [pc.EstimateNormals(geometry.KDTreeSearchParamHybrid(4 * s, 30), ctx=ctx) for pc in (ps, pt)])
for pc in (ps, pt):
    pc.normals_[pc.normals_[:, 2] < 0] *= -1.0
t_f, (fs, ft) = timed(lambda: [reg.ComputeFPFHFeature(pc, geometry.KDTreeSearchParamHybrid(10 * s, 100), ctx=ctx) for pc in (ps, pt)])
t_m, (idx, _) = timed(lambda: reg.FeatureNearest(fs, ft, ctx=ctx))
r = 3 * s
chk = [reg.CorrespondenceCheckerBasedOnEdgeLength(0.9), reg.CorrespondenceCheckerBasedOnDistance(r)]
t_r, res = timed(lambda: reg.RegistrationRANSACBasedOnFeatureMatching(ps, pt, fs, ft, r, None, 4, chk,
                                                                      reg.RANSACConvergenceCriteria(100000, 1000), seed=3, ctx=ctx))
fine = reg.RegistrationICP(ps, pt, 2 * s, res.transformation_, reg.TransformationEstimationPointToPlane(), ctx=ctx)
print(json.dumps({"what": "global registration pieces, %d x %d points, host buffers" % (n, n),
                  "estimate_normals_both_ms": t_n, "fpfh_hybrid100_both_ms": t_f, "feature_matching_all_ms": t_m,
                  "feature_pairs_per_s": n * n / (t_m * 1e-3), "ransac_100000x1000_ms": t_r, "ransac_fitness": res.fitness_,
                  "ransac_iterations": res.iterations_, "ransac_validations": res.validations_,
                  "icp_refined_T_error": float(np.abs(fine.transformation_ - sc["T_true"]).max())}))

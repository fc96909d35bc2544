#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small input/output vectors for the ICP path.

The reference ships no fixtures and cannot be built or imported here (C++ needing Eigen3 + FLANN, both absent), so the
expected outputs come from two sources, both recorded in each file:
  * `np_*`  : independent numpy computations (brute-force nearest neighbour with the reference's operation order, numpy
              linear algebra) -- not produced by the oracle;
  * `orc_*` : the CPU oracle (oracle/o3d_oracle.c), i.e. regression vectors for the restatement itself.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from open3d_amd import synthetic  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def np_brute(src, tgt, thr):
    d = src[:, None, :] - tgt[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    j = np.argmin(d2, axis=1)
    m = d2[np.arange(len(src)), j]
    return np.where(m < thr, j, -1).astype(np.int32), np.where(m < thr, m, 0.0)


def main():
    oracle.build()
    sc = synthetic.make_pair(900, 1100, cfg_id=77, theta_deg=0.8, tau_in_spacings=(1.0, -0.7, 0.5), L=6.0)
    s = sc["spacing"]
    src, tgt = sc["source"], sc["target"]
    nrm = oracle.estimate_normals(tgt, oracle.HYBRID, 30, 4 * s)
    out = {"source": src, "target": tgt, "spacing": np.float64(s), "T_true": sc["T_true"], "target_normals_hybrid": nrm}
    # correspondences at the initial pose, numpy brute force (independent of the oracle)
    for name, r in (("r2", 2 * s), ("r3", 3 * s)):
        thr = float(np.float64(np.float32(r * r)))
        corr, d2 = np_brute(src, tgt, thr)
        out["np_corr_" + name] = corr
        out["np_sumd2_" + name] = np.float64(d2.sum())
        out["radius_" + name] = np.float64(r)
    # full ICP runs (oracle): per-pass log {fitness, rmse, n_corr, sum_d2, T colmajor}
    for name, est, r in (("p2p", 1, 3 * s), ("p2plane", 2, 2 * s)):
        res = oracle.registration_icp(src, tgt, r, None, est, nrm, want_log=True, brute=True)
        out["orc_%s_T" % name] = res.transformation
        out["orc_%s_corr" % name] = res.corr
        out["orc_%s_log" % name] = res.log
        out["orc_%s_scalars" % name] = np.array([res.fitness, res.inlier_rmse, res.iterations, r])
    # preprocessing
    v = 3 * s
    vp, _, _, vv = oracle.voxel_down_sample(tgt, v)
    out["voxel_size"] = np.float64(v)
    out["orc_voxel_points"] = vp
    out["orc_voxel_index"] = vv
    out["orc_normals_knn30"] = oracle.estimate_normals(tgt, oracle.KNN, 30, 0.0)
    np.savez_compressed(os.path.join(OUT, "icp_small.npz"), **out)
    print("wrote", os.path.join(OUT, "icp_small.npz"), os.path.getsize(os.path.join(OUT, "icp_small.npz")), "bytes")


if __name__ == "__main__":
    main()

"""Committed golden vectors (tests/golden/icp_small.npz, made by tests/golden/make_golden.py).

CPU part: the oracle reproduces them (the `np_*` entries were computed by numpy alone).  GPU part: the HIP library
reproduces them through the C ABI without the oracle in the loop.
"""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp_small.npz"), allow_pickle=False)


def test_oracle_reproduces_golden(oracle):
    src, tgt, nrm = G["source"], G["target"], G["target_normals_hybrid"]
    for name in ("r2", "r3"):
        corr, nc, sd2, fit, rmse = oracle.correspondences(src, tgt, float(G["radius_" + name]))
        assert np.array_equal(corr, G["np_corr_" + name])
        assert abs(sd2 - float(G["np_sumd2_" + name])) <= 1e-12 * sd2
    for name, est in (("p2p", 1), ("p2plane", 2)):
        fit, rmse, iters, r = G["orc_%s_scalars" % name]
        res = oracle.registration_icp(src, tgt, r, None, est, nrm, want_log=True)  # KD-tree back-end vs brute fixtures
        assert res.iterations == int(iters) and np.array_equal(res.corr, G["orc_%s_corr" % name])
        assert np.allclose(res.transformation, G["orc_%s_T" % name], rtol=0, atol=1e-12)
        assert np.allclose(res.log, G["orc_%s_log" % name], rtol=1e-12, atol=1e-15)
    vp, _, _, vv = oracle.voxel_down_sample(tgt, float(G["voxel_size"]))
    assert np.array_equal(vv, G["orc_voxel_index"]) and np.array_equal(vp, G["orc_voxel_points"])
    assert np.allclose(oracle.estimate_normals(tgt, oracle.KNN, 30, 0.0), G["orc_normals_knn30"], atol=1e-12)


@pytest.mark.gpu
def test_gpu_reproduces_golden(ctx):
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    src, tgt, nrm = G["source"], G["target"], G["target_normals_hybrid"]
    source, target = geometry.PointCloud(src), geometry.PointCloud(tgt, normals=nrm)
    for name in ("r2", "r3"):
        got = reg.EvaluateRegistration(source, target, float(G["radius_" + name]), ctx=ctx)
        assert np.array_equal(got.corr_, G["np_corr_" + name])
        nc = int((G["np_corr_" + name] >= 0).sum())
        assert got.fitness_ == nc / len(src)
        assert abs(got.inlier_rmse_ - np.sqrt(float(G["np_sumd2_" + name]) / nc)) <= 1e-12
    for name, est in (("p2p", reg.TransformationEstimationPointToPoint()), ("p2plane", reg.TransformationEstimationPointToPlane())):
        fit, rmse, iters, r = G["orc_%s_scalars" % name]
        sess = reg.IcpSession(ctx).setup(src, tgt, nrm, r, np.eye(4), est, reg.ICPConvergenceCriteria())
        sess.run()
        got = sess.finish()
        log = sess.log()
        want = G["orc_%s_log" % name]
        assert got.iterations_ == int(iters) and np.array_equal(got.corr_, G["orc_%s_corr" % name])
        assert np.max(np.abs(got.transformation_ - G["orc_%s_T" % name])) < 1e-6 * np.max(np.abs(G["orc_%s_T" % name]))
        assert abs(got.fitness_ - fit) <= 1e-6 * fit and abs(got.inlier_rmse_ - rmse) <= 1e-6 * rmse
        assert len(log) == len(want)
        assert np.array_equal(log[:, 2], want[:, 2])  # n_corr per pass
        assert np.allclose(log[:, :2], want[:, :2], rtol=1e-6) and np.allclose(log[:, 4:20], want[:, 4:20], rtol=1e-6, atol=1e-9)
    ds, vox = geometry.PointCloud(tgt).VoxelDownSample(float(G["voxel_size"]), ctx=ctx, return_voxel_index=True)
    assert np.array_equal(vox, G["orc_voxel_index"]) and np.array_equal(ds.points_, G["orc_voxel_points"])
    pc = geometry.PointCloud(tgt)
    pc.EstimateNormals(ctx=ctx)
    assert np.allclose(pc.normals_, G["orc_normals_knn30"], atol=1e-9)

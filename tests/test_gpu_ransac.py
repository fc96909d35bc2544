"""RegistrationRANSACBasedOnFeatureMatching (Registration.cpp:199-301) on the device, through the C ABI.

The reference samples with std::random_device, so whole-run parity does not exist even for the reference against itself.
Parity is per piece -- feature matching (bit-exact), checkers + ComputeTransformation (1e-6), validation
(tests/test_gpu_companions.py::test_evaluate_hypotheses_*) -- plus the end-to-end property: the known motion is recovered.
"""
import numpy as np
import pytest

from conftest import make_scene, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", [33, 7, 64])
def test_feature_nearest_matches_oracle_bit_exact(ctx, oracle, dim):
    from open3d_amd.pipelines import registration as reg
    rng = np.random.default_rng(dim)
    n_src, n_tgt = 3000, 5000
    # coarse quantisation: many exact ties in distance, decided by the lowest target index
    tf = np.round(rng.random((dim, n_tgt)) * 4.0) * 25.0
    sf = np.round(rng.random((dim, n_src)) * 4.0) * 25.0
    tf[:, 4000:4500] = tf[:, 100:600]      # exact duplicate columns
    sf[:, :200] = tf[:, 4100:4300]         # queries that sit on a duplicated column: distance 0, index 200..399 must win
    idx, d2 = reg.FeatureNearest(reg.Feature(sf), reg.Feature(tf), ctx=ctx)
    ridx, rd2 = oracle.feature_nearest(sf, tf)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)         # same summation order, no fused multiply-add: identical bits
    assert np.array_equal(idx[:200], np.arange(200, 400))
    # a queried subset, with out-of-range ids
    q = np.array([5, 2999, 0, -1, 3000, 17], np.int32)
    idx, d2 = reg.FeatureNearest(reg.Feature(sf), reg.Feature(tf), q, ctx=ctx)
    ridx, rd2 = oracle.feature_nearest(sf, tf, q)
    assert np.array_equal(idx, ridx) and idx[3] == -1 and idx[4] == -1
    assert np.array_equal(d2, rd2)


def test_feature_nearest_continuous_features_and_few_queries(ctx, oracle):
    """FPFH-like continuous values; a handful of queries exercises the many-chunk split of the target range."""
    from open3d_amd.pipelines import registration as reg
    rng = np.random.default_rng(1)
    tf = rng.random((33, 20000)) * 100.0
    sf = rng.random((33, 500)) * 100.0
    q = np.array([3, 499, 250], np.int32)
    for ids in (None, q):
        idx, d2 = reg.FeatureNearest(reg.Feature(sf), reg.Feature(tf), ids, ctx=ctx)
        ridx, rd2 = oracle.feature_nearest(sf, tf, ids)
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    assert len(reg.FeatureNearest(reg.Feature(np.zeros((33, 0))), reg.Feature(tf), ctx=ctx)[0]) == 0
    assert np.array_equal(reg.FeatureNearest(reg.Feature(sf), reg.Feature(np.zeros((33, 0))), ctx=ctx)[0], np.full(500, -1))


@pytest.mark.parametrize("est", ["p2p", "p2p_scale", "p2plane"])
def test_ransac_hypotheses_match_oracle(ctx, oracle, est):
    """Checkers (CorrespondenceChecker.cpp:38-82) and ComputeTransformation over ransac_n pairs, per hypothesis."""
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(4000, 4000, cfg_id=41, theta_deg=5.0, tau=(3.0, -2.0, 1.0), noise=0.0)
    src, tgt, s = sc["source"], sc["target"], sc["spacing"]
    tn = oracle.estimate_normals(tgt, oracle.HYBRID, 30, 4 * s)
    sn = oracle.estimate_normals(src, oracle.HYBRID, 30, 4 * s)
    rng = np.random.default_rng(3)
    n_hyp, ransac_n = 300, (6 if est == "p2plane" else 4)
    c = np.zeros((n_hyp, ransac_n, 2), np.int32)
    c[:, :, 0] = rng.integers(0, len(src), (n_hyp, ransac_n))
    # half of the sets: true neighbours (source i was drawn near target... use the nearest target under the true motion)
    moved = oracle.transform_points(sc["T_true"], src)
    near = oracle.correspondences(moved, tgt, 10 * s)[0]
    c[:, :, 1] = np.where(rng.random((n_hyp, ransac_n)) < 0.7, near[c[:, :, 0]], rng.integers(0, len(tgt), (n_hyp, ransac_n)))
    c[:, :, 1] = np.where(c[:, :, 1] < 0, 0, c[:, :, 1])
    if est == "p2plane":
        estimation = reg.TransformationEstimationPointToPlane()
    else:
        estimation = reg.TransformationEstimationPointToPoint(est == "p2p_scale")
    checkers = [reg.CorrespondenceCheckerBasedOnEdgeLength(0.9), reg.CorrespondenceCheckerBasedOnDistance(3 * s),
                reg.CorrespondenceCheckerBasedOnNormal(0.5)]
    ps, pt = geometry.PointCloud(src, normals=sn), geometry.PointCloud(tgt, normals=tn)
    T, ok = reg.RansacHypotheses(ps, pt, c, estimation, checkers, ctx=ctx)
    n_ok = 0
    for k in range(n_hyp):
        ids, tids = c[k, :, 0], c[k, :, 1]
        edge_ok = oracle.ransac_check(src, tgt, c[k], np.eye(4), edge_sim=0.9)
        if not edge_ok:
            assert not ok[k]
            continue
        if est == "p2plane":
            Tr = oracle.p2plane_compute(src[ids], tgt[tids], tn[tids], np.arange(ransac_n, dtype=np.int32))[0]
        else:
            Tr = oracle.p2p_compute(src[ids], tgt[tids], np.arange(ransac_n, dtype=np.int32), est == "p2p_scale")
        # a handful of pairs: the conditioning of the estimate varies; compare where it is sane
        if np.all(np.isfinite(Tr)) and np.linalg.cond(Tr[:3, :3]) < 1e6:
            assert rel_err(T[k], Tr) < 1e-6, k
        want = oracle.ransac_check(src, tgt, c[k], Tr, 0.9, 3 * s, 0.5, sn, tn)
        assert bool(ok[k]) == want, k
        n_ok += want
    assert 0 < n_ok < n_hyp   # both outcomes occurred
    # no checkers: everything passes; out-of-range indices fail
    T2, ok2 = reg.RansacHypotheses(ps, pt, c, estimation, None, ctx=ctx)
    assert ok2.all()
    bad = c.copy()
    bad[0, 0, 1] = len(tgt)
    assert not reg.RansacHypotheses(ps, pt, bad, estimation, None, ctx=ctx)[1][0]


def test_ransac_feature_matching_recovers_known_motion(ctx, oracle):
    """Global registration: FPFH on both clouds, RANSAC with the usual checkers; no initial guess."""
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(6000, 6000, cfg_id=42, theta_deg=25.0, tau=(40.0, -25.0, 10.0), noise=0.02)
    s = sc["spacing"]
    ps, pt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    for pc in (ps, pt):
        pc.EstimateNormals(geometry.KDTreeSearchParamHybrid(4 * s, 30), ctx=ctx)
        # a consistent orientation (the height field faces +z within 25 degrees)
        pc.normals_[pc.normals_[:, 2] < 0] *= -1.0
    fs = reg.ComputeFPFHFeature(ps, geometry.KDTreeSearchParamHybrid(10 * s, 100), ctx=ctx)
    ft = reg.ComputeFPFHFeature(pt, geometry.KDTreeSearchParamHybrid(10 * s, 100), ctx=ctx)
    r = 3 * s
    checkers = [reg.CorrespondenceCheckerBasedOnEdgeLength(0.9), reg.CorrespondenceCheckerBasedOnDistance(r)]
    crit = reg.RANSACConvergenceCriteria(40000, 1000)
    res = reg.RegistrationRANSACBasedOnFeatureMatching(ps, pt, fs, ft, r, reg.TransformationEstimationPointToPoint(False), 4, checkers,
                                                       crit, seed=7, ctx=ctx)
    assert res.validations_ <= 1000 and res.iterations_ <= 40000 + 256
    # the returned numbers are those of the returned transformation (Registration.cpp:271)
    ecorr, _, efit, ermse = oracle.evaluate_registration(sc["source"], sc["target"], r, res.transformation_)
    assert res.fitness_ == pytest.approx(efit, rel=1e-12) and res.inlier_rmse_ == pytest.approx(ermse, rel=1e-9)
    assert np.array_equal(res.corr_, ecorr)
    assert res.fitness_ > 0.5, "RANSAC did not find the motion: fitness %.3f after %d validations" % (res.fitness_, res.validations_)
    # coarse pose, as global registration gives it: within a few spacings of the truth over the cloud
    moved = oracle.transform_points(res.transformation_, sc["source"])
    truth = oracle.transform_points(sc["T_true"], sc["source"])
    assert np.sqrt(((moved - truth) ** 2).sum(1)).mean() < 3 * s
    # same seed, same run
    res2 = reg.RegistrationRANSACBasedOnFeatureMatching(ps, pt, fs, ft, r, reg.TransformationEstimationPointToPoint(False), 4, checkers,
                                                        crit, seed=7, ctx=ctx)
    assert np.array_equal(res2.transformation_, res.transformation_) and res2.fitness_ == res.fitness_
    # ICP refinement from the RANSAC pose closes the gap (the usual global -> local pipeline)
    fine = reg.RegistrationICP(ps, pt, 2 * s, res.transformation_, reg.TransformationEstimationPointToPlane(), ctx=ctx)
    assert rel_err(fine.transformation_, sc["T_true"]) < 5e-3


def test_ransac_edge_cases(ctx):
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(300, 300, cfg_id=43)
    ps, pt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    f = reg.Feature(np.random.default_rng(0).random((33, 300)))
    for kw in (dict(ransac_n=2), dict(max_correspondence_distance=0.0)):   # Registration.cpp:207-209
        args = dict(max_correspondence_distance=0.1, ransac_n=4)
        args.update(kw)
        res = reg.RegistrationRANSACBasedOnFeatureMatching(ps, pt, f, f, args["max_correspondence_distance"], None, args["ransac_n"],
                                                           None, reg.RANSACConvergenceCriteria(100, 10), seed=1, ctx=ctx)
        assert np.array_equal(res.transformation_, np.eye(4)) and res.fitness_ == 0.0 and (res.corr_ == -1).all()
    # max_validation bounds the validations; the lazy feature matching path (few iterations on a larger cloud)
    res = reg.RegistrationRANSACBasedOnFeatureMatching(ps, pt, f, f, 0.5, None, 3, None, reg.RANSACConvergenceCriteria(20, 7), seed=1,
                                                       ctx=ctx)
    assert res.validations_ == 7
    assert res.iterations_ == 7   # no checkers: every iteration validates, the loop condition (:230) stops it after seven


def test_ransac_based_on_correspondence(ctx, oracle):
    """RegistrationRANSACBasedOnCorrespondence (Registration.cpp:170-197).  The sampling is random (std::mt19937 here,
    random_device in the reference); what must hold for every run: the returned numbers and inlier set are the oracle's
    evaluation (:93-116) of the returned transformation, which is ComputeTransformation of SOME ransac_n listed pairs."""
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(5000, 5000, cfg_id=44, theta_deg=30.0, tau=(30.0, 10.0, -20.0), noise=0.05)
    src, tgt, s = sc["source"], sc["target"], sc["spacing"]
    moved = oracle.transform_points(sc["T_true"], src)
    near = oracle.correspondences(moved, tgt, 5 * s)[0]
    rng = np.random.default_rng(9)
    ids = np.flatnonzero(near >= 0)[:3000]
    c = np.stack([ids, near[ids]], 1).astype(np.int32)
    bad = rng.random(len(c)) < 0.5                          # half of the list is wrong
    c[bad, 1] = rng.integers(0, len(tgt), bad.sum())
    ps, pt = geometry.PointCloud(src), geometry.PointCloud(tgt)
    r = 1.5 * s
    # one round per call: the evaluation of 40 different sampled transformations against the oracle
    hits = 0
    for seed in range(40):
        res = reg.RegistrationRANSACBasedOnCorrespondence(ps, pt, c, r, None, 3, reg.RANSACConvergenceCriteria(1, 1), seed=seed, ctx=ctx)
        fit, rmse, mask = oracle.ransac_corr_evaluate(src, tgt, c, res.transformation_, r)
        if fit == 0.0:   # the reference keeps RegistrationResult(): identity, nothing
            assert res.fitness_ == 0.0 and np.array_equal(res.transformation_, np.eye(4)) and not res.inlier_mask_.any()
            continue
        hits += 1
        assert res.fitness_ == fit and rel_err(res.inlier_rmse_, rmse) < 1e-12
        assert np.array_equal(res.inlier_mask_, mask) and np.array_equal(res.correspondence_set_, c[mask])
    assert hits > 0
    # the default-sized run finds the motion despite 50 % outliers, reproducibly for a given seed
    res = reg.RegistrationRANSACBasedOnCorrespondence(ps, pt, c, r, reg.TransformationEstimationPointToPoint(False), 6,
                                                      reg.RANSACConvergenceCriteria(4000, 4000), seed=3, ctx=ctx)
    fit, rmse, mask = oracle.ransac_corr_evaluate(src, tgt, c, res.transformation_, r)
    assert res.fitness_ == fit and rel_err(res.inlier_rmse_, rmse) < 1e-12 and np.array_equal(res.inlier_mask_, mask)
    assert res.fitness_ > 0.35 and (mask & ~bad).sum() > 0.7 * (~bad).sum()
    again = reg.RegistrationRANSACBasedOnCorrespondence(ps, pt, c, r, reg.TransformationEstimationPointToPoint(False), 6,
                                                        reg.RANSACConvergenceCriteria(4000, 4000), seed=3, ctx=ctx)
    assert np.array_equal(again.transformation_, res.transformation_)
    moved_est = oracle.transform_points(res.transformation_, src)
    assert np.sqrt(((moved_est - moved) ** 2).sum(1)).mean() < 2 * s
    # point-to-plane estimation over the samples (needs target normals; identity updates without them)
    tn = oracle.estimate_normals(tgt, oracle.HYBRID, 30, 4 * s)
    res = reg.RegistrationRANSACBasedOnCorrespondence(ps, geometry.PointCloud(tgt, normals=tn), c, r,
                                                      reg.TransformationEstimationPointToPlane(), 6,
                                                      reg.RANSACConvergenceCriteria(500, 500), seed=3, ctx=ctx)
    fit, rmse, mask = oracle.ransac_corr_evaluate(src, tgt, c, res.transformation_, r)
    assert res.fitness_ == fit and np.array_equal(res.inlier_mask_, mask)
    # :176-178
    for kw in (dict(ransac_n=2), dict(r=0.0), dict(c=c[:2])):
        out = reg.RegistrationRANSACBasedOnCorrespondence(ps, pt, kw.get("c", c), kw.get("r", r), None, kw.get("ransac_n", 3),
                                                          reg.RANSACConvergenceCriteria(10, 10), seed=1, ctx=ctx)
        assert out.fitness_ == 0.0 and np.array_equal(out.transformation_, np.eye(4)) and len(out.correspondence_set_) == 0


def test_ransac_argument_errors(ctx):
    """Bad arguments fail loudly with a status, never silently: dimensions beyond the device path, mismatching counts,
    estimations / checkers without a device path."""
    from open3d_amd import geometry, _lib
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(200, 200, cfg_id=45)
    ps, pt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    f33 = reg.Feature(np.zeros((33, 200)))
    with pytest.raises(_lib.O3DError):   # 129-dimensional features: beyond the LDS-resident generic path
        reg.FeatureNearest(reg.Feature(np.zeros((129, 10))), reg.Feature(np.zeros((129, 10))), ctx=ctx)
    with pytest.raises(_lib.O3DError):   # feature count != point count
        reg.RegistrationRANSACBasedOnFeatureMatching(ps, pt, reg.Feature(np.zeros((33, 199))), f33, 0.1, seed=1, ctx=ctx)
    with pytest.raises(_lib.O3DError):   # dimensions differ
        reg.RegistrationRANSACBasedOnFeatureMatching(ps, pt, reg.Feature(np.zeros((32, 200))), f33, 0.1, seed=1, ctx=ctx)

    class MyChecker:
        require_pointcloud_alignment_ = False

    with pytest.raises(_lib.O3DError):   # a user-defined checker has no device path
        reg.RegistrationRANSACBasedOnFeatureMatching(ps, pt, f33, f33, 0.1, None, 4, [MyChecker()], seed=1, ctx=ctx)
    # several checkers of one kind combine into the strictest
    edge, dist, ang = reg._checker_thresholds([reg.CorrespondenceCheckerBasedOnEdgeLength(0.8), reg.CorrespondenceCheckerBasedOnEdgeLength(0.9),
                                               reg.CorrespondenceCheckerBasedOnDistance(0.3), reg.CorrespondenceCheckerBasedOnDistance(0.2)])
    assert (edge, dist, ang) == (0.9, 0.2, -1.0)
    # the context is still usable
    assert reg.EvaluateRegistration(ps, pt, 3 * sc["spacing"], ctx=ctx).fitness_ > 0

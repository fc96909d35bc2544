"""GPU parity tests of the ICP hot path: HIP library through the C ABI vs the CPU oracle.

Bars (BASELINE.json north_star): correspondence sets bit-exact (ties -> lowest target index); transformation,
fitness and inlier_rmse within 1e-6 relative (the fp64 reduction order differs).
"""
import numpy as np
import pytest

from conftest import make_scene, rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cert_policy")]

TOL = 1e-6


def _reg():
    from open3d_amd import geometry, pipelines
    return geometry, pipelines.registration


@pytest.mark.parametrize("n_src,n_tgt,rs", [(5000, 6000, 3.0), (20000, 20000, 2.0), (1000, 50000, 1.0), (3, 10, 5.0)])
def test_evaluate_registration_matches_oracle(ctx, oracle, n_src, n_tgt, rs):
    geometry, reg = _reg()
    sc = make_scene(n_src, n_tgt, cfg_id=n_src)
    r = rs * sc["spacing"]
    T = sc["T_true"]
    got = reg.EvaluateRegistration(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), r, T, ctx=ctx)
    corr, nc, fit, rmse = oracle.evaluate_registration(sc["source"], sc["target"], r, T)
    assert np.array_equal(got.corr_, corr)
    assert got.fitness_ == fit
    assert rel_err(got.inlier_rmse_, rmse) < 1e-12
    assert len(got.correspondence_set_) == nc


def test_evaluate_identity_skips_transform(ctx, oracle):
    geometry, reg = _reg()
    sc = make_scene(4000, 4000, cfg_id=7)
    r = 3 * sc["spacing"]
    got = reg.EvaluateRegistration(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), r, ctx=ctx)
    corr, nc, fit, rmse = oracle.evaluate_registration(sc["source"], sc["target"], r)
    assert np.array_equal(got.corr_, corr) and got.fitness_ == fit


@pytest.mark.parametrize("est", ["p2p", "p2plane"])
@pytest.mark.parametrize("n", [3000, 30000])
def test_registration_icp_matches_oracle(ctx, oracle, est, n):
    geometry, reg = _reg()
    sc = make_scene(n, n + 500, cfg_id=11 + n)
    s = sc["spacing"]
    tgt_n = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    source = geometry.PointCloud(sc["source"])
    target = geometry.PointCloud(sc["target"], normals=tgt_n)
    if est == "p2p":
        estimation, est_type, r = reg.TransformationEstimationPointToPoint(False), 1, 3 * s
    else:
        estimation, est_type, r = reg.TransformationEstimationPointToPlane(), 2, 2 * s
    got = reg.RegistrationICP(source, target, r, np.eye(4), estimation, reg.ICPConvergenceCriteria(), ctx=ctx)
    ref = oracle.registration_icp(sc["source"], sc["target"], r, None, est_type, tgt_n)
    assert got.iterations_ == ref.iterations
    assert rel_err(got.transformation_, ref.transformation) < TOL
    assert rel_err(got.fitness_, ref.fitness) < TOL
    assert rel_err(got.inlier_rmse_, ref.inlier_rmse) < TOL
    assert np.array_equal(got.corr_, ref.corr)


def test_icp_per_pass_log_matches_oracle(ctx, oracle):
    """Fixed 10 iterations (criteria 0,0,10): every pass's fitness/rmse/T within 1e-6, n_corr identical."""
    geometry, reg = _reg()
    sc = make_scene(20000, 20000, cfg_id=5)
    s = sc["spacing"]
    tgt_n = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], tgt_n, 2 * s, np.eye(4),
                                     reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0, 0, 10))
    sess.run()
    res = sess.finish()
    log = sess.log()
    ref = oracle.registration_icp(sc["source"], sc["target"], 2 * s, None, 2, tgt_n, rel_fitness=0, rel_rmse=0, max_iter=10,
                                  want_log=True)
    assert res.iterations_ == ref.iterations == 10
    assert len(log) == len(ref.log) == 11
    for p in range(11):
        assert log[p, 2] == ref.log[p, 2], "n_corr differs at pass %d" % p
        assert rel_err(log[p, 0], ref.log[p, 0]) < TOL and rel_err(log[p, 1], ref.log[p, 1]) < TOL
        assert rel_err(log[p, 4:20], ref.log[p, 4:20]) < TOL
    assert np.array_equal(res.corr_, ref.corr)


def test_debug_getters_of_the_processing_order(ctx, oracle):
    """o3d_debug_get_source_perm / o3d_debug_get_match_positions (tuning hooks): the processing order is a permutation of the
    source, and the match positions it carries name the cell-sorted target consistently with the correspondences -- none
    where there is no correspondence, one position per target index."""
    geometry, reg = _reg()
    sc = make_scene(30000, 25000, cfg_id=8)
    s = sc["spacing"]
    src = sc["source"].copy()
    src[:500, 2] += 50.0 * s                       # some points without a correspondence
    sess = reg.IcpSession(ctx).setup(src, sc["target"], None, 2 * s, np.eye(4), reg.TransformationEstimationPointToPoint(),
                                     reg.ICPConvergenceCriteria(0, 0, 4))
    sess.run()
    res = sess.finish()
    n = len(src)
    perm, pos = ctx.source_perm(n), ctx.match_positions(n)
    assert np.array_equal(np.sort(perm), np.arange(n))
    corr = res.corr_[perm]                          # correspondences in processing order
    none = pos == 0xFFFFFFFF
    assert np.array_equal(none, corr < 0) and none.sum() >= 500
    assert pos[~none].max() < len(sc["target"])
    by_pos = {}
    for c, q in zip(corr[~none].tolist(), pos[~none].tolist()):
        assert by_pos.setdefault(q, c) == c         # one target index per position ...
    assert len(set(by_pos.values())) == len(by_pos)  # ... and one position per target index


def test_icp_with_init_and_scaling(ctx, oracle):
    geometry, reg = _reg()
    sc = make_scene(8000, 9000, cfg_id=3, theta_deg=2.0, tau=(4, -3, 2))
    s = sc["spacing"]
    init = sc["T_true"].copy()
    init[:3, 3] += 0.5 * s
    got = reg.RegistrationICP(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), 3 * s, init,
                              reg.TransformationEstimationPointToPoint(True), reg.ICPConvergenceCriteria(1e-7, 1e-7, 20), ctx=ctx)
    ref = oracle.registration_icp(sc["source"], sc["target"], 3 * s, init, 1, None, with_scaling=True, rel_fitness=1e-7,
                                  rel_rmse=1e-7, max_iter=20)
    assert got.iterations_ == ref.iterations
    assert rel_err(got.transformation_, ref.transformation) < TOL
    assert np.array_equal(got.corr_, ref.corr)


def test_errors_mirror_reference(ctx):
    geometry, reg = _reg()
    from open3d_amd import _lib
    pc = geometry.PointCloud(np.random.rand(10, 3))
    with pytest.raises(_lib.O3DError, match="Invalid max_correspondence_distance"):
        reg.RegistrationICP(pc, pc, 0.0, ctx=ctx)
    with pytest.raises(_lib.O3DError, match="require pre-computed normal vectors"):
        reg.RegistrationICP(pc, pc, 0.1, np.eye(4), reg.TransformationEstimationPointToPlane(), ctx=ctx)


def test_edge_cases(ctx, oracle):
    geometry, reg = _reg()
    rng = np.random.default_rng(1)
    tgt = rng.random((500, 3))
    # empty source, empty target, no correspondences, NaN / inf points, duplicate target points (tie -> lowest index)
    res = reg.EvaluateRegistration(geometry.PointCloud(np.zeros((0, 3))), geometry.PointCloud(tgt), 0.1, ctx=ctx)
    assert res.fitness_ == 0 and res.inlier_rmse_ == 0 and len(res.correspondence_set_) == 0
    res = reg.EvaluateRegistration(geometry.PointCloud(tgt), geometry.PointCloud(np.zeros((0, 3))), 0.1, ctx=ctx)
    assert res.fitness_ == 0 and np.all(res.corr_ == -1)
    far = tgt + 100.0
    res = reg.RegistrationICP(geometry.PointCloud(far), geometry.PointCloud(tgt), 0.1, ctx=ctx)
    assert res.fitness_ == 0 and res.inlier_rmse_ == 0 and np.allclose(res.transformation_, np.eye(4))
    src = rng.random((300, 3))
    src[5] = np.nan
    src[17, 1] = np.inf
    tgt2 = np.concatenate([tgt, tgt[:50]])  # duplicates of the first 50 points at higher indices
    tgt2[100] = np.nan
    got = reg.EvaluateRegistration(geometry.PointCloud(src), geometry.PointCloud(tgt2), 0.2, ctx=ctx)
    corr, nc, fit, rmse = oracle.evaluate_registration(src, tgt2, 0.2, brute=True)
    assert np.array_equal(got.corr_, corr)
    assert got.corr_[5] == -1 and got.corr_[17] == -1
    assert np.all(got.corr_ < 500) and not np.any(got.corr_ == 100)
    assert got.fitness_ == fit and rel_err(got.inlier_rmse_, rmse) < 1e-12


def test_radius_threshold_is_float_rounded(ctx, oracle):
    """d2 < (double)(float)(r*r): r = 0.05 -> 0.0024999999441206455 (< r^2); r = 0.075 -> 0.005625000223517418 (> r^2)."""
    geometry, reg = _reg()
    for r, inside in [(0.05, False), (0.075, True)]:
        tgt = np.array([[0.0, 0.0, 0.0], [10.0, 10.0, 10.0]])
        src = np.array([[r, 0.0, 0.0]])  # d2 == r*r exactly representable product
        d2 = src[0, 0] * src[0, 0]
        thr = oracle.radius_threshold(r)
        assert (d2 < thr) == inside
        got = reg.EvaluateRegistration(geometry.PointCloud(src), geometry.PointCloud(tgt), r, ctx=ctx)
        assert (got.corr_[0] == 0) == inside


@pytest.mark.parametrize("est", ["p2p", "p2plane"])
@pytest.mark.parametrize("f32", [True, False])
def test_both_search_kernels_match_oracle(oracle, est, f32):
    """fp32-prefilter search (+ fp64 confirmation + exact fixup; the default) and the per-lane fp64 search kernel must
    both give the exact result."""
    geometry, reg = _reg()
    from open3d_amd import _lib
    c = _lib.Context(0)
    c.debug_set(use_tile_search=f32, collect_tile_stats=True)
    c.tile_stats(reset=True)  # the counters are process-wide
    try:
        sc = make_scene(40000, 42000, cfg_id=17)
        s = sc["spacing"]
        tgt_n = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
        if est == "p2p":
            estimation, est_type, r = reg.TransformationEstimationPointToPoint(False), 1, 3 * s
        else:
            estimation, est_type, r = reg.TransformationEstimationPointToPlane(), 2, 2 * s
        got = reg.RegistrationICP(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=tgt_n), r,
                                  np.eye(4), estimation, reg.ICPConvergenceCriteria(), ctx=c)
        st = c.tile_stats()
        assert (st["wave_iters"] > 0) == f32, "wrong search kernel ran"
        ref = oracle.registration_icp(sc["source"], sc["target"], r, None, est_type, tgt_n)
        assert got.iterations_ == ref.iterations
        assert np.array_equal(got.corr_, ref.corr)
        assert rel_err(got.transformation_, ref.transformation) < TOL
        assert rel_err(got.inlier_rmse_, ref.inlier_rmse) < TOL
        # lattice targets with near-tie distances exercise the undecided -> exact fixup route
        rng = np.random.default_rng(3)
        g = np.arange(0, 24) * 0.1
        tgt = np.stack(np.meshgrid(g, g, g[:5], indexing="ij"), -1).reshape(-1, 3)
        src = tgt[rng.integers(0, len(tgt), 20000)] + rng.choice([0.0, 0.05, -0.05, 1e-9, 0.049999], size=(20000, 3))
        gotl = reg.EvaluateRegistration(geometry.PointCloud(src), geometry.PointCloud(tgt), 0.1, ctx=c)
        corr = oracle.evaluate_registration(src, tgt, 0.1)[0]
        assert np.array_equal(gotl.corr_, corr)
    finally:
        c.close()


def test_stepwise_sharded_api_on_one_gpu(ctx, oracle):
    """o3d_icp_pass_to + o3d_icp_solve on torch device tensors (the calls the multi-GPU driver makes), one rank, and a
    two-shard emulation on one GPU: two contexts, sums concatenated by hand, every context solving from both."""
    import torch
    geometry, reg = _reg()
    from open3d_amd import _lib
    from open3d_amd.sharded import shard_bounds
    sc = make_scene(30001, 30000, cfg_id=23)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    ref = oracle.registration_icp(sc["source"], sc["target"], 2 * s, None, 2, nrm, rel_fitness=0, rel_rmse=0, max_iter=6)
    dev = torch.device("cuda", 0)
    d_tgt, d_nrm = torch.from_numpy(sc["target"]).to(dev), torch.from_numpy(nrm).to(dev)
    n = len(sc["source"])
    ctxs, sess, shards = [], [], []
    for rnk in range(2):
        lo, hi = shard_bounds(n, 2, rnk)
        c = _lib.Context(0)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        d_src = torch.from_numpy(sc["source"][lo:hi]).to(dev)
        se = reg.IcpSession(c).setup(d_src, d_tgt, d_nrm, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                     reg.ICPConvergenceCriteria(0, 0, 6), n_src_global=n)
        ctxs.append(c); sess.append(se); shards.append(d_src)
    allsums = torch.zeros(64, dtype=torch.float64, device=dev)
    for _ in range(7):
        for rnk in range(2):
            sess[rnk].enqueue_pass_to(allsums[32 * rnk:].data_ptr())
        for rnk in range(2):
            sess[rnk].enqueue_solve(allsums.data_ptr(), 2)
    res = [se.finish() for se in sess]
    assert np.array_equal(res[0].transformation_, res[1].transformation_)  # bit-identical on both "ranks"
    assert np.array_equal(np.concatenate([r.corr_ for r in res]), ref.corr)
    assert res[0].iterations_ == ref.iterations == 6
    assert rel_err(res[0].transformation_, ref.transformation) < TOL
    assert rel_err(res[0].fitness_, ref.fitness) < TOL and rel_err(res[0].inlier_rmse_, ref.inlier_rmse) < TOL
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_point_to_plane_on_an_exact_plane_zero_pivots(ctx, oracle):
    """A target lying exactly in z = 0 with normals (0, 0, 1): three columns of J are exactly zero, JTJ has exact zero
    rows, and the pivoted LDLT (Eigen semantics, utility/Eigen.cpp:85) meets zero pivots -- those components of the update
    must come out as zero, on the device as in the oracle, every iteration."""
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    rng = np.random.default_rng(5)
    tgt = np.zeros((4000, 3))
    tgt[:, :2] = rng.random((4000, 2)) * 4.0
    nrm = np.tile([0.0, 0.0, 1.0], (len(tgt), 1))
    src = np.zeros((3000, 3))
    src[:, :2] = 0.5 + rng.random((3000, 2)) * 3.0
    src[:, 2] = 0.01 + 0.02 * (src[:, 0] - 2.0)        # tilted and lifted: only z, rot-x, rot-y are observable
    r = 0.2
    crit = reg.ICPConvergenceCriteria(0.0, 0.0, 5)
    got = reg.RegistrationICP(geometry.PointCloud(src), geometry.PointCloud(tgt, normals=nrm), r, np.eye(4),
                              reg.TransformationEstimationPointToPlane(), crit, ctx=ctx)
    ref = oracle.registration_icp(src, tgt, r, None, 2, nrm, rel_fitness=0.0, rel_rmse=0.0, max_iter=5)
    assert np.all(np.isfinite(got.transformation_))
    assert np.array_equal(got.corr_, ref.corr)
    assert rel_err(got.transformation_, ref.transformation) < 1e-6
    assert got.fitness_ == ref.fitness and rel_err(got.inlier_rmse_, ref.inlier_rmse) < 1e-6
    # the unobservable rotation about z stays untouched (zero pivot -> zero component)
    T = got.transformation_
    assert abs(T[1, 0]) < 1e-12 and abs(T[0, 1]) < 1e-12

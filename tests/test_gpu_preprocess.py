"""GPU parity tests of the preprocessing kernels (VoxelDownSample, EstimateNormals) against the CPU oracle."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _geo():
    from open3d_amd import geometry
    return geometry


@pytest.mark.parametrize("n,voxel_s", [(20000, 2.0), (50000, 5.0), (3000, 0.3)])
def test_voxel_down_sample_bit_exact(ctx, oracle, n, voxel_s):
    """Per-voxel sums run in input order on both sides: points / normals / colours must be IDENTICAL doubles."""
    geometry = _geo()
    sc = make_scene(n, n, cfg_id=21)
    rng = np.random.default_rng(3)
    pts = sc["target"]
    nrm = rng.normal(size=pts.shape)
    nrm[::17, 1] = np.nan  # NaN normals are skipped in the sum but counted in the divisor (PointCloud.cpp:193-208)
    col = rng.random(pts.shape)
    v = voxel_s * sc["spacing"]
    got, vox = geometry.PointCloud(pts, nrm, col).VoxelDownSample(v, ctx=ctx, return_voxel_index=True)
    rp, rn, rc, rv = oracle.voxel_down_sample(pts, v, nrm, col)
    assert len(got.points_) == len(rp)
    assert np.array_equal(vox, rv)
    assert np.array_equal(got.points_, rp)
    assert np.array_equal(got.normals_, rn, equal_nan=True)
    assert np.array_equal(got.colors_, rc)


def test_voxel_down_sample_points_only_and_edges(ctx, oracle):
    geometry = _geo()
    rng = np.random.default_rng(0)
    pts = rng.random((5000, 3)) * np.array([3.0, 1.0, 0.2]) - 1.0
    for v in (0.05, 0.5, 10.0):
        got, vox = geometry.PointCloud(pts).VoxelDownSample(v, ctx=ctx, return_voxel_index=True)
        rp, _, _, rv = oracle.voxel_down_sample(pts, v)
        assert np.array_equal(vox, rv) and np.array_equal(got.points_, rp)
        assert not got.HasNormals() and not got.HasColors()
    one = geometry.PointCloud(pts[:1]).VoxelDownSample(0.1, ctx=ctx)
    assert np.array_equal(one.points_, pts[:1])
    empty = geometry.PointCloud(np.zeros((0, 3))).VoxelDownSample(0.1, ctx=ctx)
    assert len(empty.points_) == 0


def test_voxel_down_sample_errors(ctx):
    geometry = _geo()
    from open3d_amd import _lib
    pc = geometry.PointCloud(np.random.default_rng(1).random((100, 3)) * 1e6)
    with pytest.raises(_lib.O3DError, match="voxel_size <= 0"):
        pc.VoxelDownSample(0.0, ctx=ctx)
    with pytest.raises(_lib.O3DError, match="voxel_size is too small"):
        pc.VoxelDownSample(1e-5, ctx=ctx)


def _normals_close(got, ref, tol=1e-9):
    d = np.abs(got - ref).max(axis=1)
    return d < tol


@pytest.mark.parametrize("mode", ["knn", "hybrid", "radius"])
@pytest.mark.parametrize("cloud", ["surface", "volume"])
def test_estimate_normals_matches_oracle(ctx, oracle, mode, cloud):
    """Same neighbour sets (exact k-best by (d2, index)), cumulants summed in the same order for knn / hybrid:
    normals agree to 1e-9 (libm acos / cos differ in the last ulp between host and device)."""
    geometry = _geo()
    if cloud == "surface":
        sc = make_scene(30000, 30000, cfg_id=8)
        pts, s = sc["target"], sc["spacing"]
    else:
        rng = np.random.default_rng(4)
        pts = rng.random((20000, 3))
        s = (1.0 / 20000) ** (1.0 / 3.0)
    if mode == "knn":
        sp, args = geometry.KDTreeSearchParamKNN(30), (oracle.KNN, 30, 0.0)
    elif mode == "hybrid":
        sp, args = geometry.KDTreeSearchParamHybrid(4 * s, 30), (oracle.HYBRID, 30, 4 * s)
    else:
        sp, args = geometry.KDTreeSearchParamRadius(2.5 * s), (oracle.RADIUS, 0, 2.5 * s)
    pc = geometry.PointCloud(pts)
    pc.EstimateNormals(sp, ctx=ctx)
    ref = oracle.estimate_normals(pts, *args)
    ok = _normals_close(pc.normals_, ref, 1e-9 if mode != "radius" else 1e-7)
    assert ok.mean() > 0.9999, "mismatching normals: %d of %d" % ((~ok).sum(), len(ok))
    # the few allowed outliers must be ill-conditioned eigen problems, not wrong neighbourhoods: still unit vectors
    assert np.allclose(np.linalg.norm(pc.normals_, axis=1), 1.0, atol=1e-9)


def test_estimate_normals_knn_small_k_and_sparse(ctx, oracle):
    geometry = _geo()
    rng = np.random.default_rng(6)
    pts = np.concatenate([rng.random((3000, 3)), rng.random((50, 3)) * 50.0 + 10.0])  # dense blob + far sparse points
    for k in (3, 7, 30):
        pc = geometry.PointCloud(pts)
        pc.EstimateNormals(geometry.KDTreeSearchParamKNN(k), ctx=ctx)
        ref = oracle.estimate_normals(pts, oracle.KNN, k, 0.0)
        assert _normals_close(pc.normals_, ref).mean() > 0.999


def test_estimate_normals_fallbacks_and_orientation(ctx, oracle):
    geometry = _geo()
    rng = np.random.default_rng(7)
    pts = rng.random((4000, 3))
    # < 3 neighbours -> (0, 0, 1)
    pc = geometry.PointCloud(pts)
    pc.EstimateNormals(geometry.KDTreeSearchParamKNN(2), ctx=ctx)
    assert np.all(pc.normals_ == np.array([0.0, 0.0, 1.0]))
    pc.EstimateNormals(geometry.KDTreeSearchParamRadius(1e-6), ctx=ctx)
    assert np.all(pc.normals_ == np.array([0.0, 0.0, 1.0]))
    two = geometry.PointCloud(pts[:2])
    two.EstimateNormals(ctx=ctx)
    assert np.all(two.normals_ == np.array([0.0, 0.0, 1.0]))
    # existing normals: result flipped to agree with them (EstimateNormals.cpp:343-345)
    prior = rng.normal(size=pts.shape)
    pc2 = geometry.PointCloud(pts, normals=prior.copy())
    pc2.EstimateNormals(geometry.KDTreeSearchParamKNN(20), ctx=ctx)
    ref = oracle.estimate_normals(pts, oracle.KNN, 20, 0.0, normals=prior)
    assert _normals_close(pc2.normals_, ref).mean() > 0.999
    assert np.all((pc2.normals_ * prior).sum(1) >= -1e-12)


def test_pipeline_downsample_normals_icp(ctx, oracle):
    """Config-3 shaped pipeline at test size: VoxelDownSample both clouds -> EstimateNormals(target) -> P2Plane ICP."""
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(60000, 60000, cfg_id=13)
    v = 2 * sc["spacing"]
    src = geometry.PointCloud(sc["source"]).VoxelDownSample(v, ctx=ctx)
    tgt = geometry.PointCloud(sc["target"]).VoxelDownSample(v, ctx=ctx)
    tgt.EstimateNormals(geometry.KDTreeSearchParamKNN(30), ctx=ctx)
    got = reg.RegistrationICP(src, tgt, 2 * v, np.eye(4), reg.TransformationEstimationPointToPlane(), ctx=ctx)
    # oracle on the same preprocessed clouds (its own preprocessing is covered by the tests above)
    ref = oracle.registration_icp(src.points_, tgt.points_, 2 * v, None, 2, tgt.normals_)
    assert got.iterations_ == ref.iterations
    assert np.array_equal(got.corr_, ref.corr)
    assert np.max(np.abs(got.transformation_ - ref.transformation)) < 1e-6


def test_point_cloud_distances_match_oracle(ctx, oracle):
    """ComputePointCloudDistance (PointCloud.cpp:112-132) and ComputeNearestNeighborDistance (:521-544)."""
    from open3d_amd import geometry
    sc = make_scene(7000, 9000, cfg_id=71, theta_deg=3.0, tau=(4.0, -2.0, 1.0))
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    # include queries far outside the target's bounding box
    far = np.vstack([sc["source"], sc["source"][:50] + np.array([30.0, -12.0, 5.0])])
    got = geometry.PointCloud(far).ComputePointCloudDistance(tgt, ctx=ctx)
    ref = oracle.neighbor_query(0, sc["target"], far, knn=1)
    assert np.array_equal(got, ref)
    got = tgt.ComputeNearestNeighborDistance(ctx=ctx)
    ref = oracle.neighbor_query(0, sc["target"], sc["target"], knn=2)
    assert np.array_equal(got, ref) and (got > 0).all()
    # degenerate sizes
    assert list(geometry.PointCloud(sc["target"][:1]).ComputeNearestNeighborDistance(ctx=ctx)) == [0.0]
    assert not src.ComputePointCloudDistance(geometry.PointCloud(), ctx=ctx).any()     # :122-126: no neighbour -> 0
    dup = geometry.PointCloud(np.repeat(sc["target"][:5], 2, axis=0))
    assert not dup.ComputeNearestNeighborDistance(ctx=ctx).any()                         # duplicates: distance 0


def test_outlier_removal_matches_oracle(ctx, oracle):
    """RemoveRadiusOutliers (PointCloud.cpp:394-419) and RemoveStatisticalOutliers (:421-468)."""
    from open3d_amd import geometry, _lib
    sc = make_scene(100, 12000, cfg_id=72)
    s = sc["spacing"]
    rng = np.random.default_rng(8)
    pts = np.vstack([sc["target"], sc["target"][rng.integers(0, 12000, 300)] + rng.normal(0, 6 * s, (300, 3))])  # + outliers
    nrm = rng.normal(size=pts.shape)
    pc = geometry.PointCloud(pts, normals=nrm)

    radius, nb = 2.5 * s, 10
    cloud, idx = pc.RemoveRadiusOutliers(nb, radius, ctx=ctx)
    cnt = oracle.neighbor_query(2, pts, pts, radius=radius)
    want = np.nonzero(cnt > nb)[0]
    assert np.array_equal(idx, want) and 0 < len(want) < len(pts)
    assert np.array_equal(cloud.points_, pts[want]) and np.array_equal(cloud.normals_, nrm[want])

    k, ratio = 20, 1.5
    cloud, idx = pc.RemoveStatisticalOutliers(k, ratio, ctx=ctx)
    avg = oracle.neighbor_query(1, pts, pts, knn=k)
    valid = np.count_nonzero(avg >= 0)
    mean = 0.0
    for a in avg:  # std::accumulate with the y > 0 filter (:453-455)
        if a > 0:
            mean += a
    mean /= valid
    sq = 0.0
    for a in avg:
        if a > 0:
            sq += (a - mean) * (a - mean)
    thr = mean + ratio * np.sqrt(sq / (valid - 1))
    want = np.nonzero((avg > 0) & (avg < thr))[0]
    assert np.array_equal(idx, want) and 0 < len(want) < len(pts)
    assert np.array_equal(cloud.points_, pts[want])

    for bad in ((0, radius), (5, 0.0)):
        with pytest.raises(_lib.O3DError):
            pc.RemoveRadiusOutliers(*bad, ctx=ctx)
    with pytest.raises(_lib.O3DError):
        pc.RemoveStatisticalOutliers(0, 1.0, ctx=ctx)
    assert len(geometry.PointCloud().RemoveStatisticalOutliers(5, 1.0, ctx=ctx)[1]) == 0


@pytest.mark.parametrize("mode", ["knn", "hybrid", "radius"])
def test_fpfh_matches_oracle(ctx, oracle, mode):
    """ComputeFPFHFeature (Feature.cpp:113-159).  Same neighbour lists and summation order as the oracle; histogram bins
    come from acos / atan2 / sqrt of the device's libm, which may differ from the host's in the last ulp, so a bin can flip
    only when a value sits on a bin edge to within an ulp: the features must agree to 1e-9 on all but a handful of points."""
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(100, 20000, cfg_id=33)
    s = sc["spacing"]
    pts = sc["target"]
    nrm = oracle.estimate_normals(pts, oracle.HYBRID, 30, 4 * s)
    pc = geometry.PointCloud(pts, normals=nrm)
    if mode == "knn":
        sp, args = geometry.KDTreeSearchParamKNN(30), (oracle.KNN, 30, 0.0)
    elif mode == "hybrid":
        sp, args = geometry.KDTreeSearchParamHybrid(5 * s, 40), (oracle.HYBRID, 40, 5 * s)
    else:
        # every neighbour within the radius (about 50 here, up to several hundred at the duplicates below), used in the
        # order of the cell walk: the weighted sums differ from the oracle's ascending-distance order by rounding only
        pts = pts.copy()
        pts[100:140] = pts[99]        # duplicates: zero distances (skipped by the weighting, counted by the histograms)
        pts[7] = np.nan
        nrm = nrm.copy()
        pc = geometry.PointCloud(pts, normals=nrm)
        sp, args = geometry.KDTreeSearchParamRadius(4 * s), (oracle.RADIUS, 0, 4 * s)
    got = reg.ComputeFPFHFeature(pc, sp, ctx=ctx)
    if mode == "radius":
        fin = np.isfinite(pts).all(axis=1)
        ref = np.zeros((len(pts), 33))
        ref[fin] = oracle.fpfh(pts[fin], nrm[fin], *args)
        assert not got.data_.T[7].any()
        g = got.data_.T
        bad = np.abs(g - ref).max(1) > 1e-9 * 100.0
        assert bad.sum() <= max(3, len(pts) // 2000), "too many points differ: %d" % bad.sum()
        return
    ref = oracle.fpfh(pts, nrm, *args)
    assert got.Dimension() == 33 and got.Num() == len(pts)
    g = got.data_.T
    # every histogram of a point with neighbours sums to 100 (three of them) up to rounding... before the SPFH is added: 200
    tot = g.reshape(len(pts), 3, 11).sum(2)
    assert np.allclose(tot[tot[:, 0] > 0], 200.0, rtol=1e-9)
    bad = np.abs(g - ref).max(1) > 1e-9 * 100.0
    assert bad.sum() <= max(3, len(pts) // 2000), "too many points differ: %d" % bad.sum()


def test_fpfh_second_pass_with_and_without_neighbour_table(ctx, monkeypatch):
    """Pass 1 of ComputeFPFHFeature reads the neighbours pass 0 left behind, or (table too large / switched off) searches
    them again: the two routes must give the same bits."""
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(100, 15000, cfg_id=36)
    s, pts = sc["spacing"], sc["target"]
    pc = geometry.PointCloud(pts)
    pc.EstimateNormals(geometry.KDTreeSearchParamHybrid(4 * s, 30), ctx=ctx)
    for sp in (geometry.KDTreeSearchParamHybrid(6 * s, 50), geometry.KDTreeSearchParamKNN(20)):
        monkeypatch.delenv("O3D_FPFH_NO_TABLE", raising=False)
        a = reg.ComputeFPFHFeature(pc, sp, ctx=ctx).data_
        monkeypatch.setenv("O3D_FPFH_NO_TABLE", "1")
        b = reg.ComputeFPFHFeature(pc, sp, ctx=ctx).data_
        assert np.array_equal(a, b) and a.any()


def test_large_neighbourhoods_up_to_256(ctx, oracle):
    """max_nn = 100 (the usual FPFH parameter, Hybrid(5 voxels, 100)) and knn = 128 / 256: lists up to 128 KiB of LDS."""
    from open3d_amd import geometry, _lib
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(100, 6000, cfg_id=35)
    s, pts = sc["spacing"], sc["target"]
    nrm = oracle.estimate_normals(pts, oracle.HYBRID, 30, 4 * s)
    pc = geometry.PointCloud(pts, normals=nrm)
    got = reg.ComputeFPFHFeature(pc, geometry.KDTreeSearchParamHybrid(8 * s, 100), ctx=ctx)
    ref = oracle.fpfh(pts, nrm, oracle.HYBRID, 100, 8 * s)
    bad = np.abs(got.data_.T - ref).max(1) > 1e-9 * 100.0
    assert bad.sum() <= 3, "too many points differ: %d" % bad.sum()
    pn = geometry.PointCloud(pts)
    for k in (128, 256):
        pn = geometry.PointCloud(pts)   # (a cloud that has normals gets the new ones oriented against them)
        pn.EstimateNormals(geometry.KDTreeSearchParamKNN(k), ctx=ctx)
        assert _normals_close(pn.normals_, oracle.estimate_normals(pts, oracle.KNN, k, 0.0)).mean() > 0.999
    with pytest.raises(_lib.O3DError):
        pn.EstimateNormals(geometry.KDTreeSearchParamKNN(257), ctx=ctx)   # beyond the LDS lists: refused, not truncated


def test_fpfh_edge_cases(ctx):
    from open3d_amd import geometry, _lib
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(10, 500, cfg_id=34)
    with pytest.raises(_lib.O3DError):
        reg.ComputeFPFHFeature(geometry.PointCloud(sc["target"]), ctx=ctx)        # no normals: Feature.cpp:117-121
    nrm = np.tile([0.0, 0.0, 1.0], (500, 1))
    pc = geometry.PointCloud(sc["target"], normals=nrm)
    assert not reg.ComputeFPFHFeature(pc, geometry.KDTreeSearchParamHybrid(1e-9, 30), ctx=ctx).data_.any()   # nobody has a neighbour
    assert not reg.ComputeFPFHFeature(pc, geometry.KDTreeSearchParamRadius(1e-9), ctx=ctx).data_.any()
    assert not reg.ComputeFPFHFeature(pc, geometry.KDTreeSearchParamRadius(0.0), ctx=ctx).data_.any()
    one = geometry.PointCloud(sc["target"][:1], normals=nrm[:1])
    assert reg.ComputeFPFHFeature(one, ctx=ctx).data_.shape == (33, 1)


@pytest.mark.parametrize("mode", ["knn", "hybrid", "radius"])
def test_estimate_normals_shards_reassemble_bit_identically(ctx, mode):
    """o3d_estimate_normals_shard x G + o3d_estimate_normals_assemble (the multi-GPU split, rehearsed on one GPU) must
    give the bits of the single call, for any number of shards, with non-finite points and an orientation reference."""
    geometry = _geo()
    from open3d_amd.sharded import HipNormalsEngine
    sc = make_scene(100, 9001, cfg_id=61)
    pts, s = sc["target"].copy(), sc["spacing"]
    pts[17] = np.nan
    pts[4000, 1] = np.inf
    sp = {"knn": geometry.KDTreeSearchParamKNN(20), "hybrid": geometry.KDTreeSearchParamHybrid(4 * s, 30),
          "radius": geometry.KDTreeSearchParamRadius(2.5 * s)}[mode]
    ref_n = np.tile([0.0, 0.0, 1.0], (len(pts), 1))
    for oriented in (None, ref_n):
        whole = geometry.PointCloud(pts, normals=None if oriented is None else oriented.copy())
        whole.EstimateNormals(sp, ctx=ctx)
        eng = HipNormalsEngine(ctx)
        for G in (1, 2, 3, 8):
            parts = []
            for g in range(G):
                part, ng = eng.shard(pts, oriented, sp, g, G)
                assert ng == len(pts) - 2 and part.shape[0] == (ng + G - 1) // G
                parts.append(part)
            # the last shard's call left the grid in the context: assemble follows directly
            got = eng.assemble(pts, np.concatenate(parts), G)
            assert np.array_equal(got, whole.normals_), (mode, G)


@pytest.mark.parametrize("k", [3, 12, 30, 64])
def test_knn_normals_by_counting_equal_the_heap_kernel(ctx, monkeypatch, k):
    """KNN search (the reference's default, KDTreeSearchParamKNN(30)) through the counting kernel -- cells sized so that the
    k nearest usually lie within one cell side, everything else handed to the heap kernel's shell walk -- against the heap
    kernel alone (O3D_NORMALS_HEAP=1): bit-identical, on a surface with a hole and an edge, a volume, a dense blob with far
    outliers (their k-th neighbour is many cells away) and a cloud of fewer than k points."""
    geometry = _geo()
    rng = np.random.default_rng(40 + k)
    sc = make_scene(60_000, 60_000, cfg_id=12)
    surf = sc["target"]
    surf = surf[np.linalg.norm(surf[:, :2] - surf[:, :2].mean(0), axis=1) > 0.7]          # a hole
    clouds = [surf, rng.random((30_000, 3)),
              np.concatenate([rng.random((5000, 3)), rng.random((40, 3)) * 80.0 + 10.0]), rng.random((k - 1, 3))]
    for pts in clouds:
        got = geometry.PointCloud(pts)
        monkeypatch.delenv("O3D_NORMALS_HEAP", raising=False)
        got.EstimateNormals(geometry.KDTreeSearchParamKNN(k), ctx=ctx)
        want = geometry.PointCloud(pts)
        monkeypatch.setenv("O3D_NORMALS_HEAP", "1")
        want.EstimateNormals(geometry.KDTreeSearchParamKNN(k), ctx=ctx)
        monkeypatch.delenv("O3D_NORMALS_HEAP", raising=False)
        assert np.array_equal(got.normals_, want.normals_)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["knn", "hybrid", "radius"])
def test_estimate_normals_with_eigen_solver(ctx, oracle, mode):
    """fast_normal_computation=False (EstimateNormals.cpp:247-251: Eigen's SelfAdjointEigenSolver, column 0) against the
    oracle's restatement of that solver -- the same sequence of IEEE operations, so the normals agree to rounding, sign
    included -- and, independently of both, against numpy's eigh on the covariance of the oracle's neighbour lists: the
    normal is the eigenvector of the smallest eigenvalue up to sign.  With an orientation reference, shards, non-finite
    points and the < 3 neighbours fallback."""
    geometry = _geo()
    from open3d_amd.sharded import HipNormalsEngine
    sc = make_scene(100, 20000, cfg_id=63)
    pts, s = sc["target"].copy(), sc["spacing"]
    pts[11] = np.nan
    sp, args = {"knn": (geometry.KDTreeSearchParamKNN(24), (oracle.KNN, 24, 0.0)),
                "hybrid": (geometry.KDTreeSearchParamHybrid(4 * s, 30), (oracle.HYBRID, 30, 4 * s)),
                "radius": (geometry.KDTreeSearchParamRadius(2.5 * s), (oracle.RADIUS, 0, 2.5 * s))}[mode]
    fin = np.isfinite(pts).all(axis=1)
    pc = geometry.PointCloud(pts)
    pc.EstimateNormals(sp, fast_normal_computation=False, ctx=ctx)
    ref = np.tile([0.0, 0.0, 1.0], (len(pts), 1))
    ref[fin] = oracle.estimate_normals(pts[fin], *args, fast=False)
    ok = _normals_close(pc.normals_, ref, 1e-9 if mode != "radius" else 1e-7)
    if mode == "radius":
        # the radius kernel adds the cumulants in traversal order, the covariance differs from the oracle's in the last
        # digits, and the SIGN the iteration leaves can hinge on such a digit (the sign of a vanishing Givens operand): the
        # direction must agree everywhere, the sign nearly everywhere
        assert ok.mean() > 0.999
        ok |= _normals_close(-pc.normals_, ref, 1e-7)
    assert ok.mean() > 0.9999, "mismatching normals: %d of %d" % ((~ok).sum(), len(ok))
    assert np.allclose(np.linalg.norm(pc.normals_, axis=1), 1.0, atol=1e-12)
    # not the closed form: FastEigen3x3 differs from the iterative solver in the last digits (and often in sign)
    fast = geometry.PointCloud(pts)
    fast.EstimateNormals(sp, ctx=ctx)
    assert not np.array_equal(fast.normals_, pc.normals_)
    par = np.abs((fast.normals_ * pc.normals_).sum(1))
    assert (par > 1 - 1e-6).mean() > 0.995
    # independent: numpy's eigh on sampled points
    fpts = pts[fin]
    fidx = np.flatnonzero(fin)
    rng = np.random.default_rng(1)
    checked = 0
    for i in rng.choice(len(fpts), 300, replace=False):
        idx, _ = oracle.neighbors(fpts, fpts[i], *args)
        if len(idx) < 3:
            continue
        nb = fpts[idx]
        cov = (nb[:, :, None] * nb[:, None, :]).mean(0) - np.outer(nb.mean(0), nb.mean(0))
        w, v = np.linalg.eigh(cov)
        if w[1] - w[0] < 1e-3 * w[2]:
            continue  # (near-degenerate: the eigenvector is ill-conditioned)
        assert abs(abs(v[:, 0] @ pc.normals_[fidx[i]]) - 1.0) < 1e-7
        checked += 1
    assert checked > 200
    # orientation reference, shards
    prior = np.tile([0.0, 0.0, -1.0], (len(pts), 1))
    pc2 = geometry.PointCloud(pts, normals=prior.copy())
    pc2.EstimateNormals(sp, fast_normal_computation=False, ctx=ctx)
    assert np.all((pc2.normals_ * prior).sum(1)[fin] >= 0.0)  # (a non-finite point gets (0, 0, 1) whatever the reference)
    flip = np.where((pc.normals_ * prior).sum(1, keepdims=True) < 0.0, -1.0, 1.0)
    assert np.array_equal(pc2.normals_[fin], (pc.normals_ * flip)[fin])
    eng = HipNormalsEngine(ctx)
    parts = [eng.shard(pts, None, sp, g, 3, fast_normal_computation=False)[0] for g in range(3)]
    assert np.array_equal(eng.assemble(pts, np.concatenate(parts), 3), pc.normals_)
    few = geometry.PointCloud(pts[:2])
    few.EstimateNormals(fast_normal_computation=False, ctx=ctx)
    assert np.all(few.normals_ == np.array([0.0, 0.0, 1.0]))
    # the setting is sticky per context: the default call puts it back
    again = geometry.PointCloud(pts)
    again.EstimateNormals(sp, ctx=ctx)
    assert np.array_equal(again.normals_, fast.normals_)


def test_hybrid_normals_where_the_counting_kernel_hands_over(ctx, oracle):
    """The hand-over paths of k_estimate_normals_hist in hybrid mode, each against the oracle: (a) more candidates inside the
    radius than a lane can note down (255) -- a dense blob with a radius that holds ~600 points: every point goes to the
    heap kernel, and until it gets there the counting kernel must work on a consistent part of its candidates (counting on
    after the notes stopped left list places unwritten in round 3: stale LDS used as positions, a memory fault);
    (b) more than 8 candidates in the slice of the k-th nearest -- a ring of 24 points at nearly the same distance from a
    centre, with max_nn cutting through the ring; (c) both kinds mixed with ordinary points in one cloud, so that wavefronts
    hold lanes of every kind."""
    geometry = _geo()
    rng = np.random.default_rng(23)
    blob = rng.random((6000, 3))                                    # 6000 points per unit cube
    ang_a, ang_b = rng.random(10) * 2 * np.pi, rng.random(24) * 2 * np.pi
    ring = lambda rho, ang: np.stack([rho * np.cos(ang), rho * np.sin(ang), rho * 0.05 * rng.standard_normal(len(ang))], 1)  # noqa: E731
    shell = np.concatenate([ring(0.02, ang_a), ring(0.035, ang_b)]) + 5.0   # 24 points of the outer ring share a slice of d2
    centre = np.array([[5.0, 5.0, 5.0]])
    plain = rng.random((3000, 3)) * np.array([4.0, 4.0, 0.05]) + np.array([10.0, 0.0, 0.0])    # a thin sheet
    for pts, radius, k in ((blob, 0.3, 30), (np.concatenate([centre, shell]), 0.05, 20),
                           (np.concatenate([blob, centre, shell, plain]), 0.3, 30)):
        pc = geometry.PointCloud(pts)
        pc.EstimateNormals(geometry.KDTreeSearchParamHybrid(radius, k), ctx=ctx)
        ref = oracle.estimate_normals(pts, oracle.HYBRID, k, radius)
        ok = _normals_close(pc.normals_, ref, 1e-9)
        assert ok.mean() > 0.999, "mismatching normals: %d of %d" % ((~ok).sum(), len(ok))

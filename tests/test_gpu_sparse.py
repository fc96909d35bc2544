"""Sparse scenes: clusters of surface far apart in a large, mostly empty volume, where a dense table of cells of side r over
the bounding box would exceed its budget.  RegistrationICP then files the target under a BRICKED table (grid.hip: only the
bricks of B^3 cells that a query can see a point from, each with a halo; KDTreeFlann::SetRawData, geometry/KDTreeFlann.cpp:
154-167, has no such size limit either).  Bars as everywhere: correspondences bit-exact against the oracle's exact search,
transformation / fitness / rmse within 1e-6 -- through the cooperative search, through the per-lane fp64 search
(o3d_debug_set), and identical to what the dense table with coarser cells (O3D_NO_BRICKS=1) gives."""
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def cluster_scene(n, side, nclusters, radius, seed, theta_deg=0.02):
    from open3d_amd import synthetic
    rng = np.random.default_rng(seed)
    centres = rng.uniform(radius, side - radius, (nclusters, 3))
    v = rng.normal(0, 1, (n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    which = rng.integers(0, nclusters, n)
    target = centres[which] + radius * v
    s = float(np.sqrt(nclusters * 4 * np.pi * radius * radius / n))
    T_true = synthetic.rigid_transform(theta_deg, np.array([1.0, -0.7, 0.5]) * s)
    Ti = np.linalg.inv(T_true)
    source = (target + rng.normal(0, 0.1 * s, target.shape)) @ Ti[:3, :3].T + Ti[:3, 3]
    return dict(source=np.ascontiguousarray(source), target=np.ascontiguousarray(target), normals=np.ascontiguousarray(v), spacing=s,
                T_true=T_true)


def _reg():
    from open3d_amd import geometry, pipelines
    return geometry, pipelines.registration


@pytest.mark.parametrize("side,expect_brick", [(300.0, 8), (4000.0, 64)])
@pytest.mark.parametrize("est_name", ["p2plane", "p2p"])
def test_bricked_table_icp_matches_oracle(ctx, oracle, side, expect_brick, est_name):
    geometry, reg = _reg()
    sc = cluster_scene(60_000, side, 12, 1.5, seed=int(side))
    s = sc["spacing"]
    src = sc["source"].copy()
    src[:500] += 3.0 * side                        # queries far outside the grid
    src[500:1000, 2] += 4.0 * s                    # ... and just out of reach of their cluster
    tgt = sc["target"]
    nrm = sc["normals"]
    if est_name == "p2p":
        est, et, r = reg.TransformationEstimationPointToPoint(False), 1, 3 * s
    else:
        est, et, r = reg.TransformationEstimationPointToPlane(), 2, 2 * s
    crit = reg.ICPConvergenceCriteria(0.0, 0.0, 6)
    source, target = geometry.PointCloud(src), geometry.PointCloud(tgt, normals=nrm)
    ref = oracle.registration_icp(src, tgt, r, None, et, nrm, rel_fitness=0.0, rel_rmse=0.0, max_iter=6)
    got = reg.RegistrationICP(source, target, r, np.eye(4), est, crit, ctx=ctx)
    info = ctx.grid_info()
    if est_name == "p2p" and expect_brick == 64:
        expect_brick = 32                          # the larger radius makes the grid 2/3 as fine
    assert info["brick_side"] == expect_brick and info["bricks"] > 0 and info["cooperative_search"] == 1.0, info
    assert np.array_equal(got.corr_, ref.corr)
    assert got.iterations_ == ref.iterations and got.fitness_ == ref.fitness
    assert rel_err(got.transformation_, ref.transformation) < 1e-6 and rel_err(got.inlier_rmse_, ref.inlier_rmse) < 1e-6
    assert (got.corr_[:500] == -1).all() and (got.corr_[500:1000] == -1).any()
    # the per-lane fp64 search reads the same table
    ctx.debug_set(use_tile_search=False)
    try:
        gen = reg.RegistrationICP(source, target, r, np.eye(4), est, crit, ctx=ctx)
    finally:
        ctx.debug_set(use_tile_search=True)
    assert np.array_equal(gen.corr_, ref.corr) and rel_err(gen.transformation_, ref.transformation) < 1e-6
    # ... and the dense table with coarser cells gives the same answer (more slowly)
    os.environ["O3D_NO_BRICKS"] = "1"
    try:
        dense = reg.RegistrationICP(source, target, r, np.eye(4), est, crit, ctx=ctx)
        assert ctx.grid_info()["brick_side"] == 0 and ctx.grid_info()["cell_side"] > 1.2 * r
    finally:
        os.environ.pop("O3D_NO_BRICKS", None)
    assert np.array_equal(dense.corr_, got.corr_) and rel_err(dense.transformation_, got.transformation_) < 1e-9
    # EvaluateRegistration on the same scene
    ev = reg.EvaluateRegistration(source, target, r, ref.transformation, ctx=ctx)
    ec, efit, ermse = oracle.evaluate_registration(src, tgt, r, ref.transformation)[:3]
    assert np.array_equal(ev.corr_, ec)


def test_bricked_table_points_on_brick_faces(ctx, oracle):
    """Target points ON the faces, edges and corners of bricks (multiples of 8 cells from the grid origin cannot be forced
    from outside, so a lattice of points one cell apart is laid through several bricks), queries on a shifted lattice:
    every 3x3x3 block that straddles bricks is exercised."""
    geometry, reg = _reg()
    h = 0.01
    rng = np.random.default_rng(3)
    g = np.arange(-14, 15) * h
    blob = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)           # 29^3 lattice points, spacing = r
    blob = blob + rng.uniform(-0.2 * h, 0.2 * h, blob.shape)
    far = np.array([[250.0, 130.0, 90.0], [10.0, 400.0, 33.0]])
    tgt = np.concatenate([blob, blob + far[0], blob[:5000] + far[1]])
    src = tgt + rng.uniform(-0.45 * h, 0.45 * h, tgt.shape)
    r = h
    want = oracle.evaluate_registration(src, tgt, r, np.eye(4))
    got = reg.EvaluateRegistration(geometry.PointCloud(src), geometry.PointCloud(tgt), r, np.eye(4), ctx=ctx)
    assert np.array_equal(got.corr_, want[0])
    # EvaluateRegistration builds its grid like RegistrationICP: the same call through the ICP driver
    res = reg.RegistrationICP(geometry.PointCloud(src), geometry.PointCloud(tgt), r, np.eye(4), reg.TransformationEstimationPointToPoint(),
                              reg.ICPConvergenceCriteria(0.0, 0.0, 2), ctx=ctx)
    assert ctx.grid_info()["brick_side"] >= 8
    ref = oracle.registration_icp(src, tgt, r, None, 1, None, rel_fitness=0.0, rel_rmse=0.0, max_iter=2)
    assert np.array_equal(res.corr_, ref.corr) and rel_err(res.transformation_, ref.transformation) < 1e-6


def test_bricked_table_under_normals_and_neighbour_queries(ctx, oracle):
    """The radius / hybrid searches of the preprocessing entry points read the 3x3x3 block only and use the bricked table of
    a sparse scene too: EstimateNormals (hybrid, by counting; radius, by heaps), ComputePointCloudDistance's companion
    counts (RemoveRadiusOutliers) and FPFH (hybrid) against the oracle, and against the dense table with coarser cells."""
    from open3d_amd import geometry
    sc = cluster_scene(40_000, 2500.0, 10, 1.2, seed=9)
    pts, s = sc["target"], sc["spacing"]
    for sp, args in ((geometry.KDTreeSearchParamHybrid(4 * s, 30), (oracle.HYBRID, 30, 4 * s)),
                     (geometry.KDTreeSearchParamRadius(2.5 * s), (oracle.RADIUS, 0, 2.5 * s))):
        pc = geometry.PointCloud(pts)
        pc.EstimateNormals(sp, ctx=ctx)
        assert ctx.grid_info()["brick_side"] >= 8, ctx.grid_info()
        ref = oracle.estimate_normals(pts, *args)
        # hybrid: the neighbours are summed in (distance, index) order everywhere.  Radius: in the order they are met, which
        # depends on the cell layout, and the covariance comes from cumulants about the origin (EstimateNormals.cpp:222-244)
        # of coordinates up to 2500 m -- sixteen digits minus nine of cancellation
        hybrid = args[0] == oracle.HYBRID
        ok = np.abs(pc.normals_ - ref).max(axis=1) < (1e-9 if hybrid else 1e-4)
        assert ok.mean() > 0.9999
        os.environ["O3D_NO_BRICKS"] = "1"
        try:
            dense = geometry.PointCloud(pts)
            dense.EstimateNormals(sp, ctx=ctx)
            assert ctx.grid_info()["brick_side"] == 0
        finally:
            os.environ.pop("O3D_NO_BRICKS", None)
        if hybrid:
            assert np.array_equal(dense.normals_, pc.normals_)
        else:
            assert (np.abs(dense.normals_ - pc.normals_).max(axis=1) < 1e-4).mean() > 0.9999
    # radius neighbour counts (RemoveRadiusOutliers, PointCloud.cpp:406-411)
    pc = geometry.PointCloud(pts)
    kept, idx = pc.RemoveRadiusOutliers(12, 3 * s, ctx=ctx)
    assert ctx.grid_info()["brick_side"] >= 8
    cnt = oracle.neighbor_query(2, pts, pts, radius=3 * s)
    assert np.array_equal(np.asarray(idx), np.nonzero(np.asarray(cnt) > 12)[0])
    os.environ["O3D_NO_BRICKS"] = "1"
    try:
        kept2, idx2 = geometry.PointCloud(pts).RemoveRadiusOutliers(12, 3 * s, ctx=ctx)
    finally:
        os.environ.pop("O3D_NO_BRICKS", None)
    assert np.array_equal(np.asarray(idx), np.asarray(idx2)) and 0 < len(idx) < len(pts)

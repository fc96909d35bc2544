"""Device-resident clouds (o3d_cloud_*, geometry.DevicePointCloud): uploaded once, down-sampled / given normals / registered in
HBM.  Every result must equal the host-buffer entry point's bit for bit (the kernels are the same), and the oracle's."""
import numpy as np
import pytest
from conftest import make_scene, rel_err

pytestmark = pytest.mark.gpu


def _mods():
    from open3d_amd import _lib, geometry
    from open3d_amd.pipelines import registration as reg
    return _lib, geometry, reg


def test_cloud_roundtrip_voxel_normals(ctx, oracle):
    _lib, geometry, reg = _mods()
    sc = make_scene(40000, 50000, cfg_id=61)
    s = sc["spacing"]
    rng = np.random.default_rng(1)
    nrm = rng.standard_normal(sc["target"].shape)
    col = rng.random(sc["target"].shape)
    host = geometry.PointCloud(sc["target"], nrm, col)
    dev = geometry.DevicePointCloud.upload(host, ctx)
    assert len(dev) == len(host.points_) and dev.HasNormals() and dev.HasColors()
    back = dev.download()
    assert np.array_equal(back.points_, host.points_) and np.array_equal(back.normals_, nrm) and np.array_equal(back.colors_, col)
    v = 2.5 * s
    dv = dev.VoxelDownSample(v).download()
    hv = host.VoxelDownSample(v, ctx=ctx)
    assert np.array_equal(dv.points_, hv.points_) and np.array_equal(dv.normals_, hv.normals_) and np.array_equal(dv.colors_, hv.colors_)
    rp, rn, rc, _ = oracle.voxel_down_sample(sc["target"], v, nrm, col)
    assert np.array_equal(dv.points_, rp) and np.array_equal(dv.normals_, rn) and np.array_equal(dv.colors_, rc)
    # normals: a cloud without normals gets them; one with normals is re-oriented against them (EstimateNormals.cpp:343-346)
    plain = geometry.DevicePointCloud.upload(sc["target"], ctx)
    assert not plain.HasNormals()
    sp = geometry.KDTreeSearchParamHybrid(4 * s, 30)
    got = plain.EstimateNormals(sp).download().normals_
    want = geometry.PointCloud(sc["target"]).EstimateNormals(sp, ctx=ctx).normals_
    assert np.array_equal(got, want)
    assert np.abs(got - oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)).max() < 1e-9
    oriented = dev.EstimateNormals(sp).download().normals_
    want_o = geometry.PointCloud(sc["target"], nrm.copy()).EstimateNormals(sp, ctx=ctx).normals_
    assert np.array_equal(oriented, want_o)
    for c in (dev, plain):
        c.free()
    # empty cloud and error behaviour
    empty = geometry.DevicePointCloud.upload(np.zeros((0, 3)), ctx)
    assert len(empty) == 0 and len(empty.VoxelDownSample(0.1)) == 0
    with pytest.raises(_lib.O3DError, match="voxel_size <= 0"):
        geometry.DevicePointCloud.upload(sc["target"][:100], ctx).VoxelDownSample(0.0)


@pytest.mark.parametrize("est_name", ["p2plane", "p2p"])
def test_cloud_registration_equals_host_call(ctx, oracle, est_name):
    _lib, geometry, reg = _mods()
    sc = make_scene(60000, 50000, cfg_id=62, theta_deg=0.4)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    est = reg.TransformationEstimationPointToPlane() if est_name == "p2plane" else reg.TransformationEstimationPointToPoint()
    r = (2 if est_name == "p2plane" else 3) * s
    src_h, tgt_h = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
    src_d, tgt_d = geometry.DevicePointCloud.upload(src_h, ctx), geometry.DevicePointCloud.upload(tgt_h, ctx)
    init = np.eye(4)
    init[:3, 3] = (0.3 * s, -0.2 * s, 0.1 * s)
    got = reg.RegistrationICP(src_d, tgt_d, r, init, est, ctx=ctx)
    want = reg.RegistrationICP(src_h, tgt_h, r, init, est, ctx=ctx)
    assert np.array_equal(got.corr_, want.corr_) and np.array_equal(got.transformation_, want.transformation_)
    assert got.fitness_ == want.fitness_ and got.inlier_rmse_ == want.inlier_rmse_ and got.iterations_ == want.iterations_
    ref = oracle.registration_icp(sc["source"], sc["target"], r, init, 2 if est_name == "p2plane" else 1, nrm)
    assert np.array_equal(got.corr_, ref.corr) and rel_err(got.transformation_, ref.transformation) < 1e-6
    # the inputs are untouched (Registration.cpp:149 registers a copy)
    assert np.array_equal(src_d.download().points_, sc["source"]) and np.array_equal(tgt_d.download().normals_, nrm)
    if est_name == "p2plane":
        bare = geometry.DevicePointCloud.upload(sc["target"], ctx)
        with pytest.raises(_lib.O3DError, match="require pre-computed normal vectors"):
            reg.RegistrationICP(src_d, bare, r, init, est, ctx=ctx)
        with pytest.raises(_lib.O3DError, match="Invalid max_correspondence_distance"):
            reg.RegistrationICP(src_d, tgt_d, 0.0, init, est, ctx=ctx)


def test_multiscale_device_resident_equals_host_chain(ctx, oracle):
    """MultiScaleICP with one upload per raw cloud (the default) against the chain of host-buffer calls, and on a multi context
    (device 0 listed three times)."""
    _lib, geometry, reg = _mods()
    sc = make_scene(300000, 280000, cfg_id=63, theta_deg=0.2, tau=(3.0, -2.0, 1.0))
    s = sc["spacing"]
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    scales = [6 * s, 3 * s, 1.5 * s]
    fin_d, per_d = reg.MultiScaleICP(src, tgt, scales, ctx=ctx)
    fin_h, per_h = reg.MultiScaleICP(src, tgt, scales, ctx=ctx, device_resident=False)
    for a, b in zip(per_d, per_h):
        assert a.n_source_ == b.n_source_ and a.n_target_ == b.n_target_ and a.iterations_ == b.iterations_
        assert np.array_equal(a.transformation_, b.transformation_) and np.array_equal(a.corr_, b.corr_)
        assert a.fitness_ == b.fitness_ and a.inlier_rmse_ == b.inlier_rmse_
    assert np.abs(fin_d.transformation_ - sc["T_true"]).max() < 1e-4
    multi = _lib.Context([0, 0, 0])
    try:
        fin_m, per_m = reg.MultiScaleICP(src, tgt, scales, ctx=multi)
        for a, b in zip(per_m, per_d):
            assert np.array_equal(a.corr_, b.corr_) and rel_err(a.transformation_, b.transformation_) < 1e-9
    finally:
        multi.close()

"""GPU test of the header-only C++ adapter (open3d_amd/cpp/open3d_amd_registration.h): a g++-built caller, written
against the reference's own names and value types, must get what the Python mirror gets through the same C ABI."""
import os
import subprocess

import numpy as np
import pytest

from conftest import make_scene, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_adapter")


def test_cpp_adapter_matches_python_mirror(ctx, oracle, tmp_path):
    if not os.path.exists(EXE):
        pytest.fail("tests/cpp/test_adapter is missing: run __graft_entry__.build() (make -C tests/cpp)")
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(6000, 7000, cfg_id=55)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    radius, voxel = 2 * s, 3 * s
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([len(sc["source"]), len(sc["target"])], np.int64).tobytes())
        f.write(np.array([radius, voxel], np.float64).tobytes())
        for a in (sc["source"], sc["target"], nrm):
            f.write(np.ascontiguousarray(a, np.float64).tobytes())
    subprocess.run([EXE, fin, fout], check=True, timeout=120)
    out = np.fromfile(fout, np.float64)
    T = out[:16].reshape(4, 4).T
    fit, rmse, npairs = out[16], out[17], int(out[18])
    pairs = out[19:19 + 2 * npairs].reshape(-1, 2).astype(np.int32)
    o = 19 + 2 * npairs
    want = reg.RegistrationICP(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm), radius,
                               np.eye(4), reg.TransformationEstimationPointToPlane(), ctx=ctx)
    assert np.array_equal(T, want.transformation_) and fit == want.fitness_ and rmse == want.inlier_rmse_
    assert np.array_equal(pairs, want.correspondence_set_)
    ev = reg.EvaluateRegistration(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), radius, T, ctx=ctx)
    assert out[o] == ev.fitness_ and out[o + 1] == ev.inlier_rmse_ and int(out[o + 2]) == len(ev.correspondence_set_)
    o += 3
    m = int(out[o]); o += 1
    ds = geometry.PointCloud(sc["target"], normals=nrm).VoxelDownSample(voxel, ctx=ctx)
    assert m == len(ds.points_) and np.array_equal(out[o:o + 3 * m].reshape(-1, 3), ds.points_)
    o += 3 * m
    n = int(out[o]); o += 1
    pc = geometry.PointCloud(sc["target"])
    pc.EstimateNormals(ctx=ctx)
    assert n == len(pc.points_) and np.array_equal(out[o:o + 3 * n].reshape(-1, 3), pc.normals_)
    o += 3 * n
    assert list(out[o:o + 3]) == [1.0, 1.0, 1.0], "reference error messages not surfaced as std::runtime_error"
    o += 3
    src_pc, tgt_pc = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
    G = reg.GetInformationMatrixFromPointClouds(src_pc, tgt_pc, radius, want.transformation_, ctx=ctx)
    assert np.array_equal(out[o:o + 36].reshape(6, 6), G)
    o += 36
    # user subclass through the virtual interface: same correspondences as the device loop, T within the parity bar
    Th = out[o:o + 16].reshape(4, 4).T
    assert rel_err(Th, want.transformation_) < 1e-6 and rel_err(out[o + 16], want.fitness_) < 1e-6
    assert rel_err(out[o + 17], want.inlier_rmse_) < 1e-6 and int(out[o + 18]) == len(want.correspondence_set_)
    assert int(out[o + 19]) == want.iterations_
    o += 20
    assert out[o] == reg.TransformationEstimationPointToPoint().ComputeRMSE(src_pc, tgt_pc, want.correspondence_set_, ctx=ctx)
    assert out[o + 1] == reg.TransformationEstimationPointToPlane().ComputeRMSE(src_pc, tgt_pc, want.correspondence_set_, ctx=ctx)
    o += 2
    shifted = np.eye(4)
    shifted[0, 3] = 0.25 * radius
    for init in (np.eye(4), shifted):
        w = reg.RegistrationICP(src_pc, tgt_pc, radius, init, reg.TransformationEstimationPointToPlane(), ctx=ctx)
        assert np.array_equal(out[o:o + 16].reshape(4, 4).T, w.transformation_)
        assert out[o + 16] == w.fitness_ and out[o + 17] == w.inlier_rmse_
        o += 18
    def col(p):
        return np.stack([0.5 + 0.4 * np.sin(2.0 * p[:, 0]), 0.5 + 0.4 * np.cos(1.5 * p[:, 1]), np.full(len(p), 0.5)], axis=1)
    cw = reg.RegistrationColoredICP(geometry.PointCloud(sc["source"], colors=col(sc["source"])),
                                    geometry.PointCloud(sc["target"], normals=nrm, colors=col(sc["target"])), radius, ctx=ctx)
    # libm sin/cos of the two hosts may differ in the last ulp of a colour: compare at the parity bar
    assert rel_err(out[o:o + 16].reshape(4, 4).T, cw.transformation_) < 1e-6
    assert rel_err(out[o + 16], cw.fitness_) < 1e-6 and rel_err(out[o + 17], cw.inlier_rmse_) < 1e-6
    o += 18
    gs, gt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm.copy())
    gs.EstimateNormals(geometry.KDTreeSearchParamHybrid(2.0 * radius, 30), ctx=ctx)
    gt.EstimateNormals(geometry.KDTreeSearchParamHybrid(2.0 * radius, 30), ctx=ctx)
    for pc in (gs, gt):
        pc.normals_[pc.normals_[:, 2] < 0] *= -1.0
    fs = reg.ComputeFPFHFeature(gs, geometry.KDTreeSearchParamHybrid(5.0 * radius, 100), ctx=ctx)
    ft = reg.ComputeFPFHFeature(gt, geometry.KDTreeSearchParamHybrid(5.0 * radius, 100), ctx=ctx)
    gw = reg.RegistrationRANSACBasedOnFeatureMatching(
        gs, gt, fs, ft, 1.5 * radius, reg.TransformationEstimationPointToPoint(False), 4,
        [reg.CorrespondenceCheckerBasedOnEdgeLength(0.9), reg.CorrespondenceCheckerBasedOnDistance(1.5 * radius)],
        reg.RANSACConvergenceCriteria(20000, 500), seed=5, ctx=ctx)
    # same seed, same device code: the same run
    assert np.array_equal(out[o:o + 16].reshape(4, 4).T, gw.transformation_)
    assert out[o + 16] == gw.fitness_ and out[o + 17] == gw.inlier_rmse_ and int(out[o + 18]) == len(gw.correspondence_set_)
    assert gw.fitness_ > 0.5
    o += 19
    cw2 = reg.RegistrationRANSACBasedOnCorrespondence(src_pc, tgt_pc, want.correspondence_set_, radius,
                                                      reg.TransformationEstimationPointToPoint(False), 6,
                                                      reg.RANSACConvergenceCriteria(200, 200), seed=11, ctx=ctx)
    assert np.array_equal(out[o:o + 16].reshape(4, 4).T, cw2.transformation_)
    assert out[o + 16] == cw2.fitness_ and out[o + 17] == cw2.inlier_rmse_ and int(out[o + 18]) == len(cw2.correspondence_set_)
    assert cw2.fitness_ > 0.9
    o += 19
    # MultiScaleICP over device-resident clouds == the chain of host-buffer calls, bit for bit
    T = np.eye(4)
    bare = geometry.PointCloud(sc["target"])
    for v in (2.0 * voxel, voxel):
        sv, tv = src_pc.VoxelDownSample(v, ctx=ctx), bare.VoxelDownSample(v, ctx=ctx)
        tv.EstimateNormals(geometry.KDTreeSearchParamHybrid(2.0 * v, 30), ctx=ctx)
        w = reg.RegistrationICP(sv, tv, 2.0 * v, T, reg.TransformationEstimationPointToPlane(), ctx=ctx)
        assert np.array_equal(out[o:o + 16].reshape(4, 4).T, w.transformation_)
        assert out[o + 16] == w.fitness_ and out[o + 17] == w.inlier_rmse_ and int(out[o + 18]) == len(w.correspondence_set_)
        T = w.transformation_
        o += 19
    m = int(out[o]); o += 1
    assert m == len(tv.points_)
    assert np.array_equal(out[o:o + 3 * m].reshape(-1, 3), tv.points_) and np.array_equal(out[o + 3 * m:o + 6 * m].reshape(-1, 3), tv.normals_)
    o += 6 * m
    assert o == len(out)


def test_cpp_adapter_over_several_devices(ctx, oracle, tmp_path):
    """The same unchanged C++ caller with O3D_AMD_DEVICES=0,0,0 in its environment: the adapter's default context spans three
    ranks (one GPU listed three times on this box), RegistrationICP shards the source -- same correspondences, same
    iteration count, transformation within 1e-9 of the one-device run."""
    if not os.path.exists(EXE):
        pytest.fail("tests/cpp/test_adapter is missing: run __graft_entry__.build() (make -C tests/cpp)")
    sc = make_scene(9000, 8000, cfg_id=56)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    fin = str(tmp_path / "in.bin")
    with open(fin, "wb") as f:
        f.write(np.array([len(sc["source"]), len(sc["target"])], np.int64).tobytes())
        f.write(np.array([2 * s, 3 * s], np.float64).tobytes())
        for a in (sc["source"], sc["target"], nrm):
            f.write(np.ascontiguousarray(a, np.float64).tobytes())
    outs = []
    for devs in (None, "0,0,0"):
        fout = str(tmp_path / ("out_%s.bin" % (devs or "one")))
        env = dict(os.environ)
        env.pop("O3D_AMD_DEVICES", None)
        if devs:
            env["O3D_AMD_DEVICES"] = devs
        subprocess.run([EXE, fin, fout], check=True, timeout=120, env=env)
        outs.append(np.fromfile(fout, np.float64))
    one, multi = outs
    npairs = int(one[18])
    assert int(multi[18]) == npairs and np.array_equal(one[19:19 + 2 * npairs], multi[19:19 + 2 * npairs])
    assert rel_err(multi[:16], one[:16]) < 1e-9 and multi[16] == one[16] and rel_err(multi[17], one[17]) < 1e-9
    ref = oracle.registration_icp(sc["source"], sc["target"], 2 * s, None, 2, nrm)
    assert npairs == ref.n_corr

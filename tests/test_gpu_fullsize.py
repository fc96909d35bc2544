"""Full-size parity of the BASELINE configs that the small tests only reach in miniature: the config-3 preprocessing at
10 M points (VoxelDownSample with more than 32 key bits, EstimateNormals on the ~3 M down-sampled target) and config 4 --
the three-scale chain over 100 M x 100 M raw points on one GPU.  The oracle orders its voxels with a radix sort at these
sizes (oracle/o3d_oracle.c: orc_voxel_order_radix; the order and the sums are those of its qsort path, which
tests/test_oracle_crosscheck.py compares), so a 100 M-point VoxelDownSample is checked bit for bit in seconds."""
import time

import numpy as np
import pytest
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _mods():
    from open3d_amd import geometry, synthetic
    from open3d_amd.pipelines import registration as reg
    return geometry, synthetic, reg


def _voxel_equal(got, vox, ref):
    rp, rn, rc, rv = ref
    assert len(got.points_) == len(rp), "%d voxels on the GPU, %d in the oracle" % (len(got.points_), len(rp))
    assert np.array_equal(vox, rv)
    assert np.array_equal(got.points_, rp)
    if rn is not None:
        assert np.array_equal(got.normals_, rn, equal_nan=True)
    if rc is not None:
        assert np.array_equal(got.colors_, rc)


def test_config3_preprocessing_full_size(ctx, oracle):
    """The 10 M-point config-3 scan: (a) VoxelDownSample(2 s) with normals and colours, bit for bit (points / normals /
    colours are sums in input order divided by the count, PointCloud.cpp:190-208); (b) the same with a voxel of s / 2,
    whose lattice needs 35 key bits (the six-pass 64-bit sort); (c) EstimateNormals(Hybrid(2 v, 30)) on the 2.9 M voxels of
    (a) against the oracle (1e-9; EstimateNormals.cpp:320-351)."""
    geometry, synthetic, reg = _mods()
    n = 10_000_000
    sc = synthetic.make_pair(n, n, cfg_id=3)
    s = sc["spacing"]
    pts = sc["target"]
    rng = np.random.default_rng(33)
    nrm = rng.standard_normal(pts.shape)
    nrm[::1001, 2] = np.nan
    col = rng.random(pts.shape)
    cloud = geometry.PointCloud(pts, nrm, col)
    v = 2.0 * s
    got, vox = cloud.VoxelDownSample(v, ctx=ctx, return_voxel_index=True)                          # (a)
    _voxel_equal(got, vox, oracle.voxel_down_sample(pts, v, nrm, col))
    assert 2_800_000 < len(got.points_) < 3_000_000
    fine, fvox = geometry.PointCloud(pts).VoxelDownSample(0.5 * s, ctx=ctx, return_voxel_index=True)  # (b)
    bits = sum(int(np.ceil(np.log2(fvox[:, d].max() + 1))) for d in range(3))
    assert bits > 32, bits
    _voxel_equal(fine, fvox, oracle.voxel_down_sample(pts, 0.5 * s))
    down = geometry.PointCloud(got.points_)                                                        # (c)
    down.EstimateNormals(geometry.KDTreeSearchParamHybrid(2.0 * v, 30), ctx=ctx)
    ref_n = oracle.estimate_normals(got.points_, oracle.HYBRID, 30, 2.0 * v)
    assert np.abs(down.normals_ - ref_n).max() < 1e-9


def test_config4_multiscale_100m_on_one_gpu(ctx, oracle):
    """BASELINE config 4 on one GPU: voxel 0.05 / 0.025 / 0.0125 m over 100 M x 100 M raw points (L = 63.2 m).  The loop is
    MultiScaleICP's, written out so that every stage can be checked where it is produced:
      * VoxelDownSample of both 100 M-point clouds at every scale: bit for bit against the oracle;
      * scales 0 and 1 (2.2 M / 8.5 M points): EstimateNormals against the oracle (1e-9), then the oracle's RegistrationICP fed
        with the GPU's normals and the same init: correspondences identical on every point, iteration count identical,
        transformation / fitness / rmse within 1e-6;
      * scale 2 (30 M points): the properties that do not need a second implementation -- the summary follows from the
        correspondence array, a fresh EvaluateRegistration at the result equals the loop's last search -- and the chain ends
        within 2e-6 of the pose the scene was built with."""
    geometry, synthetic, reg = _mods()
    n = 100_000_000
    t0 = time.time()
    dsc = synthetic.make_pair_device(ctx, n, n, cfg_id=4, theta_deg=0.05, tau_in_spacings=(4.0, -3.0, 2.0), want_normals=False)
    src, tgt = dsc["source"].cpu().numpy(), dsc["target"].cpu().numpy()
    T_true = dsc["T_true"]
    del dsc
    import torch
    torch.cuda.empty_cache()
    print("config 4: scene on the host after %.0f s" % (time.time() - t0), flush=True)
    source, target = geometry.PointCloud(src), geometry.PointCloud(tgt)
    crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 30)
    est = reg.TransformationEstimationPointToPlane()
    T = np.eye(4)
    sizes = []
    chain_T = []
    for k, v in enumerate([0.05, 0.025, 0.0125]):
        sv, svox = source.VoxelDownSample(v, ctx=ctx, return_voxel_index=True)
        tv, tvox = target.VoxelDownSample(v, ctx=ctx, return_voxel_index=True)
        _voxel_equal(sv, svox, oracle.voxel_down_sample(src, v))
        _voxel_equal(tv, tvox, oracle.voxel_down_sample(tgt, v))
        del svox, tvox
        tv = geometry.PointCloud(tv.points_)
        tv.EstimateNormals(geometry.KDTreeSearchParamHybrid(2.0 * v, 30), ctx=ctx)
        r = 2.0 * v
        got = reg.RegistrationICP(sv, tv, r, T, est, crit, ctx=ctx)
        sizes.append((len(sv.points_), len(tv.points_), got.iterations_))
        print("config 4: scale %d (v = %g): %d x %d points, %d iterations, fitness %.4f, %.0f s so far"
              % (k, v, len(sv.points_), len(tv.points_), got.iterations_, got.fitness_, time.time() - t0), flush=True)
        if k < 2:
            ref_n = oracle.estimate_normals(tv.points_, oracle.HYBRID, 30, 2.0 * v)
            assert np.abs(tv.normals_ - ref_n).max() < 1e-9
            ref = oracle.registration_icp(sv.points_, tv.points_, r, T, 2, tv.normals_)
            assert got.iterations_ == ref.iterations
            bad = np.nonzero(got.corr_ != ref.corr)[0]
            assert len(bad) == 0, "scale %d: %d correspondences differ, first at %s" % (k, len(bad), bad[:5])
            assert rel_err(got.transformation_, ref.transformation) < 1e-6
            assert rel_err(got.fitness_, ref.fitness) < 1e-6 and rel_err(got.inlier_rmse_, ref.inlier_rmse) < 1e-6
        else:
            m = got.corr_ >= 0
            Tk = got.transformation_
            moved = sv.points_[m] @ Tk[:3, :3].T + Tk[:3, 3]
            d2 = ((moved - tv.points_[got.corr_[m]]) ** 2).sum(1)
            assert d2.max() < oracle.radius_threshold(r) * (1 + 1e-9)
            assert got.fitness_ == m.sum() / len(sv.points_)
            assert abs(got.inlier_rmse_ - np.sqrt(d2.mean())) < 1e-9 * got.inlier_rmse_
            ev = reg.EvaluateRegistration(sv, geometry.PointCloud(tv.points_), r, Tk, ctx=ctx)
            assert np.array_equal(ev.corr_, got.corr_)
            assert ev.fitness_ == got.fitness_
        T = got.transformation_
        chain_T.append((T.copy(), got.fitness_, got.inlier_rmse_, got.iterations_))
    # the same chain as ONE call on device-resident clouds (o3d_cloud_*: one upload per raw cloud, every scale made and
    # consumed in HBM): per scale the same transformation, fitness, rmse and iteration count, bit for bit
    t1 = time.time()
    fin, per = reg.MultiScaleICP(source, target, [0.05, 0.025, 0.0125], criteria=crit, ctx=ctx, want_corr=False)
    print("config 4: MultiScaleICP on device-resident clouds from host buffers: %.2f s" % (time.time() - t1), flush=True)
    for (Tk, fit, rmse, iters), r_ in zip(chain_T, per):
        assert np.array_equal(r_.transformation_, Tk) and r_.fitness_ == fit and r_.inlier_rmse_ == rmse and r_.iterations_ == iters
    assert [(r_.n_source_, r_.n_target_) for r_ in per] == [(a, b) for a, b, _ in sizes]
    assert 2_000_000 < sizes[0][0] < 2_500_000 and 8_000_000 < sizes[1][0] < 9_000_000 and 28_000_000 < sizes[2][0] < 32_000_000, sizes
    assert np.abs(T - T_true).max() < 2e-6, np.abs(T - T_true).max()

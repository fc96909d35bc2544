"""o3d_ctx_create_multi: RegistrationICP sharded over several devices behind the C ABI, rehearsed on the one
GPU of the box by listing device 0 several times -- every rank is a context of its own with its own stream, grid and
source shard, the exchange goes through the pinned host mailbox and cross-stream events exactly as it would between GPUs.
Bars: correspondences equal to the unsharded run and to the oracle bit for bit; transformation / fitness / rmse within
1e-9 of the unsharded run (the fp64 sums are added per shard, then in rank order) and 1e-6 of the oracle."""
import numpy as np
import pytest

from conftest import make_scene, rel_err

pytestmark = pytest.mark.gpu


def _reg():
    from open3d_amd import geometry, pipelines
    return geometry, pipelines.registration


@pytest.mark.parametrize("ranks", [2, 3, 8])
@pytest.mark.parametrize("est", ["p2plane", "p2p"])
def test_multi_context_icp_matches_single_and_oracle(ctx, oracle, ranks, est):
    from open3d_amd import _lib
    geometry, reg = _reg()
    sc = make_scene(60_000, 50_000, cfg_id=100 + ranks)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    source, target = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
    if est == "p2p":
        estimation, est_type, r = reg.TransformationEstimationPointToPoint(False), 1, 3 * s
    else:
        estimation, est_type, r = reg.TransformationEstimationPointToPlane(), 2, 2 * s
    m = _lib.Context([0] * ranks)
    try:
        assert m._lib.o3d_ctx_num_devices(m.handle) == ranks and ctx._lib.o3d_ctx_num_devices(ctx.handle) == 1
        for crit in (reg.ICPConvergenceCriteria(), reg.ICPConvergenceCriteria(0.0, 0.0, 9)):   # with and without the break
            one = reg.RegistrationICP(source, target, r, np.eye(4), estimation, crit, ctx=ctx)
            got = reg.RegistrationICP(source, target, r, np.eye(4), estimation, crit, ctx=m)
            ref = oracle.registration_icp(sc["source"], sc["target"], r, None, est_type, nrm, rel_fitness=crit.relative_fitness_,
                                          rel_rmse=crit.relative_rmse_, max_iter=crit.max_iteration_)
            assert got.iterations_ == one.iterations_ == ref.iterations
            assert np.array_equal(got.corr_, one.corr_) and np.array_equal(got.corr_, ref.corr)
            assert rel_err(got.transformation_, one.transformation_) < 1e-9
            assert got.fitness_ == one.fitness_ and rel_err(got.inlier_rmse_, one.inlier_rmse_) < 1e-9
            assert rel_err(got.transformation_, ref.transformation) < 1e-6 and rel_err(got.inlier_rmse_, ref.inlier_rmse) < 1e-6
    finally:
        m.close()


def test_multi_context_device_buffers_and_tiny_shards(ctx, oracle):
    """Caller buffers resident in HBM (every rank reads the same device pointers); fewer source points than ranks (empty
    shards); argument errors surface as on an ordinary context."""
    import ctypes as C
    import torch
    from open3d_amd import _lib
    geometry, reg = _reg()
    m = _lib.Context([0, 0, 0, 0])
    try:
        sc = make_scene(30_000, 30_000, cfg_id=55)
        s = sc["spacing"]
        nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
        dev = torch.device("cuda", 0)
        d_src, d_tgt, d_nrm = (torch.from_numpy(a).to(dev) for a in (sc["source"], sc["target"], nrm))
        d_corr = torch.full((len(sc["source"]),), -7, dtype=torch.int32, device=dev)
        T = np.zeros(16)
        nc, fit, rmse, iters = C.c_int64(), C.c_double(), C.c_double(), C.c_int()
        torch.cuda.synchronize()
        m.check(m._lib.o3d_registration_icp(m.handle, _lib.ptr(d_src), len(sc["source"]), _lib.ptr(d_tgt), _lib.ptr(d_nrm), len(sc["target"]),
                                            2 * s, _lib.colmajor(np.eye(4)).ctypes.data_as(_lib.c_dp), 2, 0, 1e-6, 1e-6, 30, _lib.MEM_DEVICE,
                                            T.ctypes.data_as(_lib.c_dp), _lib.ptr(d_corr), C.byref(nc), C.byref(fit), C.byref(rmse),
                                            C.byref(iters)))
        ref = oracle.registration_icp(sc["source"], sc["target"], 2 * s, None, 2, nrm)
        assert np.array_equal(d_corr.cpu().numpy(), ref.corr) and iters.value == ref.iterations and nc.value == ref.n_corr
        assert rel_err(_lib.from_colmajor(T), ref.transformation) < 1e-6
        # three source points over four ranks
        few = geometry.PointCloud(sc["source"][:3])
        tgt = geometry.PointCloud(sc["target"], normals=nrm)
        got = reg.RegistrationICP(few, tgt, 2 * s, np.eye(4), reg.TransformationEstimationPointToPoint(), reg.ICPConvergenceCriteria(0, 0, 2), ctx=m)
        one = reg.RegistrationICP(few, tgt, 2 * s, np.eye(4), reg.TransformationEstimationPointToPoint(), reg.ICPConvergenceCriteria(0, 0, 2), ctx=ctx)
        assert np.array_equal(got.corr_, one.corr_) and rel_err(got.transformation_, one.transformation_) < 1e-9
        with pytest.raises(_lib.O3DError) as ei:
            reg.RegistrationICP(few, tgt, 0.0, np.eye(4), reg.TransformationEstimationPointToPoint(), reg.ICPConvergenceCriteria(), ctx=m)
        assert "Invalid max_correspondence_distance" in str(ei.value)
        with pytest.raises(_lib.O3DError):
            reg.RegistrationICP(few, geometry.PointCloud(sc["target"]), s, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                reg.ICPConvergenceCriteria(), ctx=m)
        # the multi context is still an ordinary context for everything else
        ev = reg.EvaluateRegistration(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), 2 * s, ref.transformation, ctx=m)
        assert np.array_equal(ev.corr_, oracle.evaluate_registration(sc["source"], sc["target"], 2 * s, ref.transformation)[0])
    finally:
        m.close()


def test_multiscale_pipeline_on_multi_context(ctx, oracle):
    """BASELINE config 3's pipeline / config 4 in miniature, inside the C ABI: MultiScaleICP (VoxelDownSample ->
    EstimateNormals -> point-to-plane ICP per scale) on a context over three ranks.  EstimateNormals and RegistrationICP
    run sharded, VoxelDownSample on rank 0: normals bit-identical to the one-device pipeline at every scale,
    correspondences identical, transformations within 1e-9; and the oracle chain agrees."""
    from open3d_amd import _lib
    geometry, reg = _reg()
    sc = make_scene(150_000, 140_000, cfg_id=77, theta_deg=1.0)
    s = sc["spacing"]
    voxels, dists = [4 * s, 2 * s, 0.0], [8 * s, 4 * s, 2 * s]
    source, target = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    m = _lib.Context([0, 0, 0])
    try:
        one, one_scales = reg.MultiScaleICP(source, target, voxels, dists, ctx=ctx)
        got, got_scales = reg.MultiScaleICP(source, target, voxels, dists, ctx=m)
        for a, b in zip(one_scales, got_scales):
            assert (a.n_source_, a.n_target_, a.iterations_) == (b.n_source_, b.n_target_, b.iterations_)
            assert np.array_equal(a.corr_, b.corr_)
            assert rel_err(b.transformation_, a.transformation_) < 1e-9
        # sharded EstimateNormals == one-device EstimateNormals, bit for bit (with and without an orientation reference)
        tgt = target.VoxelDownSample(2 * s, ctx=ctx)
        pa, pb = geometry.PointCloud(tgt.points_), geometry.PointCloud(tgt.points_)
        sp = geometry.KDTreeSearchParamHybrid(4 * s, 30)
        pa.EstimateNormals(sp, ctx=ctx)
        pb.EstimateNormals(sp, ctx=m)
        assert np.array_equal(pa.normals_, pb.normals_)
        flip = -pa.normals_
        pa2, pb2 = geometry.PointCloud(tgt.points_, normals=flip.copy()), geometry.PointCloud(tgt.points_, normals=flip.copy())
        pa2.EstimateNormals(geometry.KDTreeSearchParamKNN(12), ctx=ctx)
        pb2.EstimateNormals(geometry.KDTreeSearchParamKNN(12), ctx=m)
        assert np.array_equal(pa2.normals_, pb2.normals_)
        # the oracle chain of the same calls
        T = np.eye(4)
        for k, (v, r) in enumerate(zip(voxels, dists)):
            src = oracle.voxel_down_sample(sc["source"], v)[0] if v > 0 else sc["source"]
            tg = oracle.voxel_down_sample(sc["target"], v)[0] if v > 0 else sc["target"]
            nrm = oracle.estimate_normals(tg, oracle.HYBRID, 30, 2.0 * v)
            ref = oracle.registration_icp(src, tg, r, T, 2, nrm)
            assert got_scales[k].iterations_ == ref.iterations and rel_err(got_scales[k].transformation_, ref.transformation) < 1e-6
            assert np.count_nonzero(got_scales[k].corr_ != ref.corr) <= 2   # normals differ in the last digits (device libm)
            T = ref.transformation
    finally:
        m.close()


def test_rccl_exchange_path_with_one_rank(oracle, tmp_path):
    """With one GPU per rank the multi context exchanges its 32 sums per iteration through an RCCL all-gather on each rank's
    stream (capi.hip, multi_registration_icp).  RCCL refuses a device listed twice, so on this one-GPU box the path is
    rehearsed with ONE rank (O3D_MULTI_ONE_RANK=1 makes o3d_ctx_create_multi build a real multi context of one rank):
    communicator from ncclCommInitAll, pass -> ncclAllGather -> solve over the gathered table, the chunked look at the loop's
    flag -- against the ordinary context and the oracle.  In a process of its own: RCCL is loaded there and nowhere else.
    N > 1 ranks over RCCL has not run anywhere."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        sys.path.insert(0, %r)
        import oracle
        from conftest import make_scene, rel_err
        from open3d_amd import _lib, geometry
        from open3d_amd.pipelines import registration as reg
        sc = make_scene(200_000, 150_000, cfg_id=77, theta_deg=0.3)
        s = sc["spacing"]
        nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
        src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
        m = _lib.Context([0])
        assert m._lib.o3d_ctx_num_devices(m.handle) == 1
        one = _lib.Context(0)
        for est, r, et in ((reg.TransformationEstimationPointToPlane(), 2 * s, 2), (reg.TransformationEstimationPointToPoint(), 3 * s, 1)):
            for crit in (reg.ICPConvergenceCriteria(), reg.ICPConvergenceCriteria(0.0, 0.0, 14)):
                a = reg.RegistrationICP(src, tgt, r, np.eye(4), est, crit, ctx=m)
                b = reg.RegistrationICP(src, tgt, r, np.eye(4), est, crit, ctx=one)
                ref = oracle.registration_icp(sc["source"], sc["target"], r, None, et, nrm, rel_fitness=crit.relative_fitness_,
                                              rel_rmse=crit.relative_rmse_, max_iter=crit.max_iteration_)
                assert a.iterations_ == b.iterations_ == ref.iterations, (a.iterations_, b.iterations_, ref.iterations)
                assert np.array_equal(a.corr_, b.corr_) and np.array_equal(a.corr_, ref.corr)
                assert rel_err(a.transformation_, b.transformation_) < 1e-12 and a.fitness_ == b.fitness_
                assert rel_err(a.transformation_, ref.transformation) < 1e-6
        d = geometry.DevicePointCloud.upload(src, m), geometry.DevicePointCloud.upload(tgt, m)
        c = reg.RegistrationICP(d[0], d[1], 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(), ctx=m)
        assert c.fitness_ > 0.99 and len(c.corr_) == len(sc["source"])
        with open("/proc/self/maps") as f:
            assert "librccl" in f.read(), "the exchange did not go through RCCL"
        m.close(); one.close()
        print("RCCL_ONE_RANK_OK")
    """) % (root, os.path.join(root, "tests"))
    env = dict(os.environ, O3D_MULTI_ONE_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("O3D_NO_RCCL", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])

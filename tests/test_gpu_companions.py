"""GPU parity tests of the loop's companions: information matrix (Registration.cpp:303-355), the
TransformationEstimation members over a given CorrespondenceSet (TransformationEstimation.cpp:38-98), the
host-estimation path (virtual ComputeTransformation) and the batch entry point -- all through the C ABI, checked
against the CPU oracle.  Bars as in test_gpu_icp.py: indices bit-exact, fp64 results within 1e-6 relative."""
import numpy as np
import pytest

from conftest import make_scene, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-6


def _reg():
    from open3d_amd import geometry, pipelines
    return geometry, pipelines.registration


def _pairs(corr):
    idx = np.nonzero(corr >= 0)[0].astype(np.int32)
    return np.stack([idx, corr[idx]], axis=1)


@pytest.mark.parametrize("n_src,n_tgt,rs,use_T", [(5000, 6000, 3.0, True), (20000, 15000, 1.5, True), (4000, 4000, 2.0, False)])
def test_information_matrix_matches_oracle(ctx, oracle, n_src, n_tgt, rs, use_T):
    geometry, reg = _reg()
    sc = make_scene(n_src, n_tgt, cfg_id=31 + n_src)
    r = rs * sc["spacing"]
    T = sc["T_true"] if use_T else np.eye(4)
    got = reg.GetInformationMatrixFromPointClouds(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), r, T,
                                                  ctx=ctx)
    ref = oracle.information_matrix(sc["source"], sc["target"], r, T)
    assert ref[3, 3] > 0
    assert got[3, 3] == ref[3, 3] and got[4, 4] == ref[3, 3] and got[5, 5] == ref[3, 3]  # number of correspondences
    assert rel_err(got, ref) < 1e-9
    assert np.array_equal(got, got.T)


def test_information_matrix_empty_cases(ctx):
    geometry, reg = _reg()
    sc = make_scene(1000, 1000, cfg_id=3)
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    assert not reg.GetInformationMatrixFromPointClouds(src, tgt, 0.0, np.eye(4), ctx=ctx).any()   # Registration.cpp:45-47
    far = np.eye(4)
    far[:3, 3] = 1e3
    assert not reg.GetInformationMatrixFromPointClouds(src, tgt, sc["spacing"], far, ctx=ctx).any()  # no correspondences


@pytest.mark.parametrize("n", [2000, 40000])
def test_compute_transformation_and_rmse_match_oracle(ctx, oracle, n):
    geometry, reg = _reg()
    sc = make_scene(n, n + 300, cfg_id=41 + n)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    corr = oracle.correspondences(sc["source"], sc["target"], 3 * s)[0]
    pairs = _pairs(corr)
    source, target = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)

    p2l = reg.TransformationEstimationPointToPlane()
    T_ref = oracle.p2plane_compute(sc["source"], sc["target"], nrm, corr)[0]
    assert rel_err(p2l.ComputeTransformation(source, target, pairs, ctx=ctx), T_ref) < TOL
    assert rel_err(p2l.ComputeRMSE(source, target, pairs, ctx=ctx), oracle.compute_rmse(sc["source"], sc["target"], nrm, corr, 2)) < 1e-12

    for scaling in (False, True):
        p2p = reg.TransformationEstimationPointToPoint(scaling)
        T_ref = oracle.p2p_compute(sc["source"], sc["target"], corr, scaling)
        assert rel_err(p2p.ComputeTransformation(source, target, pairs, ctx=ctx), T_ref) < TOL
    assert rel_err(reg.TransformationEstimationPointToPoint().ComputeRMSE(source, target, pairs, ctx=ctx),
                   oracle.compute_rmse(sc["source"], sc["target"], None, corr, 1)) < 1e-12

    # order of the set does not matter beyond rounding; repeats are allowed (the reference sums over the set as given)
    perm = np.random.default_rng(1).permutation(len(pairs))
    T_ref = oracle.p2plane_compute(sc["source"], sc["target"], nrm, corr)[0]
    assert rel_err(p2l.ComputeTransformation(source, target, pairs[perm], ctx=ctx), T_ref) < TOL


def test_compute_transformation_edge_cases(ctx):
    geometry, reg = _reg()
    from open3d_amd import _lib
    sc = make_scene(500, 500, cfg_id=2)
    source, target = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    empty = np.zeros((0, 2), np.int32)
    for est in (reg.TransformationEstimationPointToPoint(), reg.TransformationEstimationPointToPlane()):
        assert np.array_equal(est.ComputeTransformation(source, target, empty, ctx=ctx), np.eye(4))   # :51-52, :76-77
        assert est.ComputeRMSE(source, target, empty, ctx=ctx) == 0.0                                  # :40-41, :64-65
    # point-to-plane without target normals: identity / 0 (:64-65, :76-77)
    some = np.array([[0, 1], [2, 3], [4, 4]], np.int32)
    p2l = reg.TransformationEstimationPointToPlane()
    assert np.array_equal(p2l.ComputeTransformation(source, target, some, ctx=ctx), np.eye(4))
    assert p2l.ComputeRMSE(source, target, some, ctx=ctx) == 0.0
    with pytest.raises(_lib.O3DError):
        reg.TransformationEstimationPointToPoint().ComputeTransformation(source, target, np.array([[0, 500]], np.int32), ctx=ctx)


@pytest.mark.parametrize("inner", ["oracle", "device"])
def test_host_estimation_path_matches_device_loop(ctx, oracle, inner):
    """A user subclass of TransformationEstimation (type Unspecified) goes through o3d_icp_pass / fetch / set_update with the
    virtual ComputeTransformation on the host; with the point-to-plane formula inside, it must reproduce the oracle's
    RegistrationICP (same correspondences, same iteration count)."""
    geometry, reg = _reg()
    sc = make_scene(6000, 6500, cfg_id=77)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    calls = []

    class HostPointToPlane(reg.TransformationEstimation):
        def ComputeTransformation(self, source, target, corres, ctx_=None):
            corr = np.full(len(source.points_), -1, np.int32)
            corr[corres[:, 0]] = corres[:, 1]
            calls.append(len(corres))
            if inner == "device":  # o3d_compute_transformation on the same context, between two passes of the loop
                return reg.TransformationEstimationPointToPlane().ComputeTransformation(source, target, corres, ctx=ctx)
            return oracle.p2plane_compute(source.points_, target.points_, target.normals_, corr)[0]

    source, target = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
    init = np.eye(4)
    init[:3, 3] = 0.3 * s
    got = reg.RegistrationICP(source, target, 2 * s, init, HostPointToPlane(), reg.ICPConvergenceCriteria(1e-6, 1e-6, 12), ctx=ctx)
    ref = oracle.registration_icp(sc["source"], sc["target"], 2 * s, init, 2, nrm, max_iter=12)
    assert got.iterations_ == ref.iterations == len(calls)
    assert np.array_equal(got.corr_, ref.corr)
    assert rel_err(got.transformation_, ref.transformation) < TOL
    assert rel_err(got.fitness_, ref.fitness) < TOL and rel_err(got.inlier_rmse_, ref.inlier_rmse) < TOL


@pytest.mark.parametrize("workers", [1, 4])
def test_batch_matches_single_calls(ctx, oracle, workers):
    """BASELINE config 5 in miniature: independent pairs of different sizes; every pair must equal its own single call."""
    geometry, reg = _reg()
    sources, targets, inits, refs = [], [], [], []
    sizes = [(3000, 3200), (500, 1500), (8000, 6000), (1200, 1200), (200, 1500), (5000, 5000), (2500, 2400)]
    for k, (ns, nt) in enumerate(sizes):
        sc = make_scene(ns, nt, cfg_id=100 + k, L=3.0)
        s = sc["spacing"]
        nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
        sources.append(geometry.PointCloud(sc["source"]))
        targets.append(geometry.PointCloud(sc["target"], normals=nrm))
        T0 = np.eye(4)
        T0[:3, 3] = 0.1 * k * s
        inits.append(T0)
    # one radius for the whole batch, as in config 5 (pairs there share N and L)
    r = 0.12
    est, crit = reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(1e-6, 1e-6, 15)
    got = reg.RegistrationICPBatch(sources, targets, r, inits, est, crit, n_workers=workers, want_corr=True, ctx=ctx)
    assert len(got) == len(sizes)
    for k in range(len(sizes)):
        ref = oracle.registration_icp(sources[k].points_, targets[k].points_, r, inits[k], 2, targets[k].normals_, max_iter=15)
        assert got[k].iterations_ == ref.iterations
        assert np.array_equal(got[k].corr_, ref.corr)
        assert rel_err(got[k].transformation_, ref.transformation) < TOL
        assert rel_err(got[k].fitness_, ref.fitness) < TOL and rel_err(got[k].inlier_rmse_, ref.inlier_rmse) < TOL


def test_batch_reports_per_pair_errors(ctx):
    geometry, reg = _reg()
    from open3d_amd import _lib
    sc = make_scene(300, 300, cfg_id=9)
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    with pytest.raises(_lib.O3DError) as ei:
        reg.RegistrationICPBatch([src, src], [tgt, tgt], -1.0, ctx=ctx)
    assert ei.value.status == 1  # Registration.cpp:134-136
    assert reg.RegistrationICPBatch([], [], 1.0, ctx=ctx) == []


def test_multi_scale_pipeline_matches_oracle_chain(ctx, oracle):
    """BASELINE configs 3 / 4 in miniature: VoxelDownSample -> EstimateNormals(hybrid) -> point-to-plane ICP per scale,
    coarse to fine, each scale starting from the previous transformation; the oracle runs the same chain."""
    geometry, reg = _reg()
    sc = make_scene(60000, 60000, cfg_id=23, theta_deg=1.0, tau=(2.0, -1.5, 1.0))
    s = sc["spacing"]
    scales = [4 * s, 2 * s]
    crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 20)
    final, per_scale = reg.MultiScaleICP(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), scales,
                                         criteria=crit, ctx=ctx)
    T = np.eye(4)
    for k, v in enumerate(scales):
        src_v = oracle.voxel_down_sample(sc["source"], v)[0]
        tgt_v = oracle.voxel_down_sample(sc["target"], v)[0]
        nrm_v = oracle.estimate_normals(tgt_v, oracle.HYBRID, 30, 2 * v)
        ref = oracle.registration_icp(src_v, tgt_v, 2 * v, T, 2, nrm_v, max_iter=20)
        got = per_scale[k]
        assert got.n_source_ == len(src_v) and got.n_target_ == len(tgt_v)
        assert got.iterations_ == ref.iterations
        assert np.array_equal(got.corr_, ref.corr)
        assert rel_err(got.transformation_, ref.transformation) < TOL
        assert rel_err(got.fitness_, ref.fitness) < TOL and rel_err(got.inlier_rmse_, ref.inlier_rmse) < TOL
        T = ref.transformation
    assert final is per_scale[-1]
    # the chain recovers the known motion to well below a spacing
    err = np.abs(final.transformation_ - sc["T_true"]).max()
    assert err < 0.2 * s


def _colored_scene(n_src, n_tgt, cfg_id):
    """Height-field pair with a smooth colour texture sampled at the (untransformed) surface position."""
    sc = make_scene(n_src, n_tgt, cfg_id=cfg_id, theta_deg=0.3, tau=(0.6, -0.4, 0.2))

    def tex(p):
        x, y = p[:, 0], p[:, 1]
        r = 0.5 + 0.4 * np.sin(2.1 * x) * np.cos(1.7 * y)
        g = 0.5 + 0.4 * np.cos(1.3 * x + 0.5)
        b = 0.5 + 0.4 * np.sin(0.9 * y - 0.3) * np.sin(1.1 * x)
        return np.stack([r, g, b], axis=1)

    Tt = sc["T_true"]
    src_world = sc["source"] @ Tt[:3, :3].T + Tt[:3, 3]   # where the source points really sit on the surface
    sc["source_colors"], sc["target_colors"] = tex(src_world), tex(sc["target"])
    return sc


@pytest.mark.parametrize("n", [3000, 30000])
def test_registration_colored_icp_matches_oracle(ctx, oracle, n):
    geometry, reg = _reg()
    sc = _colored_scene(n, n + 400, cfg_id=61 + n)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    source = geometry.PointCloud(sc["source"], colors=sc["source_colors"])
    target = geometry.PointCloud(sc["target"], normals=nrm, colors=sc["target_colors"])
    r = 2.5 * s
    got = reg.RegistrationColoredICP(source, target, r, np.eye(4), reg.ICPConvergenceCriteria(1e-6, 1e-6, 20), ctx=ctx,
                                     return_gradients=True)
    ref = oracle.registration_colored_icp(sc["source"], sc["source_colors"], sc["target"], nrm, sc["target_colors"], r, max_iter=20)
    # gradients: same neighbour sets, 3x3 normal equations summed in the same order -> tight
    scale = np.abs(ref.gradients).max()
    assert scale > 0
    assert np.abs(got.color_gradient_ - ref.gradients).max() < 1e-9 * scale
    assert got.iterations_ == ref.iterations
    assert np.array_equal(got.corr_, ref.corr)
    assert rel_err(got.transformation_, ref.transformation) < TOL
    assert rel_err(got.fitness_, ref.fitness) < TOL and rel_err(got.inlier_rmse_, ref.inlier_rmse) < TOL
    # and it differs from plain point-to-plane (the photometric rows matter)
    p2l = oracle.registration_icp(sc["source"], sc["target"], r, None, 2, nrm, max_iter=20)
    assert rel_err(ref.transformation, p2l.transformation) > 1e-7


def test_colored_icp_edge_cases(ctx, oracle):
    geometry, reg = _reg()
    from open3d_amd import _lib
    sc = _colored_scene(2000, 2200, cfg_id=5)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    src_c = geometry.PointCloud(sc["source"], colors=sc["source_colors"])
    tgt_c = geometry.PointCloud(sc["target"], normals=nrm, colors=sc["target_colors"])
    with pytest.raises(_lib.O3DError) as ei:
        reg.RegistrationColoredICP(src_c, tgt_c, 0.0, ctx=ctx)
    assert ei.value.status == 1
    with pytest.raises(_lib.O3DError) as ei:
        reg.RegistrationColoredICP(src_c, geometry.PointCloud(sc["target"], colors=sc["target_colors"]), 2 * s, ctx=ctx)
    assert ei.value.status == 2
    # without colours every update is the identity (ColoredICP.cpp:127-128): the result of the initial search, one iteration
    got = reg.RegistrationColoredICP(geometry.PointCloud(sc["source"]), tgt_c, 2 * s, ctx=ctx)
    corr, nc, fit, rmse = oracle.evaluate_registration(sc["source"], sc["target"], 2 * s)
    assert np.array_equal(got.transformation_, np.eye(4)) and got.iterations_ == 1
    assert np.array_equal(got.corr_, corr) and got.fitness_ == fit
    # lambda outside [0, 1] falls back to 0.968 (:50-53)
    a = reg.RegistrationColoredICP(src_c, tgt_c, 2 * s, lambda_geometric=7.0, ctx=ctx)
    b = reg.RegistrationColoredICP(src_c, tgt_c, 2 * s, lambda_geometric=0.968, ctx=ctx)
    assert np.array_equal(a.transformation_, b.transformation_)


def test_evaluate_hypotheses_matches_oracle(ctx, oracle):
    """RANSAC's validation step (Registration.cpp:269-275) over a list of candidate transformations."""
    geometry, reg = _reg()
    sc = make_scene(9000, 10000, cfg_id=88)
    s = sc["spacing"]
    r = 2 * s
    rng = np.random.default_rng(4)
    Ts = [np.eye(4), sc["T_true"]]
    for k in range(10):
        T = sc["T_true"].copy()
        T[:3, 3] += rng.normal(0, (k + 1) * 0.5 * s, 3)
        Ts.append(T)
    far = np.eye(4)
    far[:3, 3] = 100.0
    Ts.append(far)
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    fit, rmse, best = reg.EvaluateHypotheses(src, tgt, r, Ts, ctx=ctx)
    want_best, bf, br = -1, 0.0, 0.0
    for k, T in enumerate(Ts):
        corr, nc, f, e = oracle.evaluate_registration(sc["source"], sc["target"], r, T)
        assert fit[k] == f and rel_err(rmse[k], e) < 1e-12 if nc else (fit[k] == 0 and rmse[k] == 0)
        if f > bf or (f == bf and e < br):
            want_best, bf, br = k, f, e
    assert fit[-1] == 0.0 and best.index_ == want_best == 1
    assert np.array_equal(best.corr_, oracle.evaluate_registration(sc["source"], sc["target"], r, Ts[want_best])[0])
    # the context still serves ordinary calls afterwards
    ev = reg.EvaluateRegistration(src, tgt, r, sc["T_true"], ctx=ctx)
    assert ev.fitness_ == fit[1]
    # degenerate: no hypothesis, invalid distance, nothing matches
    assert reg.EvaluateHypotheses(src, tgt, r, [], ctx=ctx)[2] is None
    f0, r0, b0 = reg.EvaluateHypotheses(src, tgt, 0.0, Ts[:3], ctx=ctx)
    assert not f0.any() and b0 is None
    assert reg.EvaluateHypotheses(src, tgt, r, [far], ctx=ctx)[2] is None


def test_sharded_batch_runner_on_one_gpu(ctx):
    """ShardedBatch + hip_batch_runner with a single rank: the list RegistrationICPBatch gives, in pair order."""
    geometry, reg = _reg()
    from open3d_amd.sharded import ShardedBatch, hip_batch_runner
    S, T_ = [], []
    for k in range(5):
        sc = make_scene(1500 + 100 * k, 1800, cfg_id=400 + k)
        S.append(geometry.PointCloud(sc["source"]))
        T_.append(geometry.PointCloud(sc["target"]))
    r = 3 * sc["spacing"]
    est, crit = reg.TransformationEstimationPointToPoint(), reg.ICPConvergenceCriteria()
    got = ShardedBatch(hip_batch_runner(S, T_, r, None, est, crit, n_workers=2, ctx=ctx)).run(len(S))
    want = reg.RegistrationICPBatch(S, T_, r, None, est, crit, n_workers=2, ctx=ctx)
    for a, b in zip(got, want):
        assert np.array_equal(a.transformation_, b.transformation_)
        assert a.fitness_ == b.fitness_ and a.inlier_rmse_ == b.inlier_rmse_ and a.iterations_ == b.iterations_


def test_device_scene_generator_matches_numpy(ctx):
    """o3d_synthetic_pair (benchmark scaffolding for config 5): the same clouds as synthetic.make_pair up to the device's
    libm -- same counter stream, same Box-Muller pairing, same inverse motion."""
    import torch
    from open3d_amd import synthetic
    for n_src, n_tgt, noise in ((5001, 7000, 0.1), (300, 200, 0.0)):
        want = synthetic.make_pair(n_src, n_tgt, cfg_id=1234, theta_deg=0.7, L=6.0, noise_in_spacings=noise)
        got = synthetic.make_pair_device(ctx, n_src, n_tgt, cfg_id=1234, theta_deg=0.7, L=6.0, noise_in_spacings=noise)
        torch.cuda.synchronize()
        assert np.abs(got["target"].cpu().numpy() - want["target"]).max() < 1e-13
        assert np.abs(got["source"].cpu().numpy() - want["source"]).max() < 1e-12
        nrm = synthetic.surface_normals(want["target"][:, 0], want["target"][:, 1])
        assert np.abs(got["normals"].cpu().numpy() - nrm).max() < 1e-13
        assert np.array_equal(got["T_true"], want["T_true"]) and got["spacing"] == want["spacing"]


def test_batch_of_200k_pairs_matches_single_calls(ctx, oracle):
    """BASELINE config 5 shape (200 k x 200 k, point-to-plane, 30 fixed iterations), clouds generated on the device: a
    handful of pairs through o3d_registration_icp_batch with 4 workers must equal their single calls (correspondences bit
    for bit, the transformation to 1e-12), and one of them the oracle."""
    import ctypes as C
    import torch
    from open3d_amd import _lib, synthetic
    n, npairs = 200_000, 6
    pairs = [synthetic.make_pair_device(ctx, n, n, cfg_id=5000 + k, L=2.83) for k in range(npairs)]
    torch.cuda.synchronize()
    s = pairs[0]["spacing"]
    VP, I64 = C.c_void_p * npairs, C.c_int64 * npairs
    a_src = VP(*[p["source"].data_ptr() for p in pairs])
    a_tgt = VP(*[p["target"].data_ptr() for p in pairs])
    a_nrm = VP(*[p["normals"].data_ptr() for p in pairs])
    cnt = I64(*[n] * npairs)
    corr = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(npairs)]
    a_corr = VP(*[c.data_ptr() for c in corr])
    T = np.zeros((npairs, 16)); nc = np.zeros(npairs, np.int64); fit = np.zeros(npairs); rmse = np.zeros(npairs)
    iters = np.zeros(npairs, np.int32); status = np.zeros(npairs, np.int32)
    ctx.check(ctx._lib.o3d_registration_icp_batch(ctx.handle, npairs, a_src, cnt, a_tgt, a_nrm, cnt, 2 * s, None, _lib.EST_POINT_TO_PLANE, 0,
                                                  0.0, 0.0, 30, _lib.MEM_DEVICE, 4, _lib.ptr(T), a_corr, _lib.ptr(nc), _lib.ptr(fit),
                                                  _lib.ptr(rmse), _lib.ptr(iters), _lib.ptr(status)))
    assert not status.any() and (iters == 30).all()
    for k in range(npairs):
        p = pairs[k]
        T1 = np.zeros(16); c1 = torch.empty(n, dtype=torch.int32, device="cuda")
        nc1, f1, r1, it1 = C.c_int64(), C.c_double(), C.c_double(), C.c_int()
        ctx.check(ctx._lib.o3d_registration_icp(ctx.handle, _lib.ptr(p["source"]), n, _lib.ptr(p["target"]), _lib.ptr(p["normals"]), n, 2 * s,
                                                _lib.colmajor(np.eye(4)).ctypes.data_as(_lib.c_dp), 2, 0, 0.0, 0.0, 30, _lib.MEM_DEVICE,
                                                T1.ctypes.data_as(_lib.c_dp), _lib.ptr(c1), C.byref(nc1), C.byref(f1), C.byref(r1), C.byref(it1)))
        # the pairs of a batch advance in lock step on a shared grid (o3d_launch_pass_gang): fewer, longer workgroups per pair
        # than the single call uses, so the fp64 sums are added in another order -- same correspondences, same count, the
        # transformation equal to rounding
        assert np.abs(T[k] - T1).max() < 1e-12 and nc[k] == nc1.value and fit[k] == f1.value and abs(rmse[k] - r1.value) < 1e-14
        assert torch.equal(corr[k], c1)
        assert np.abs(_lib.from_colmajor(T[k]) - p["T_true"]).max() < 1e-3
    p = pairs[2]
    ref = oracle.registration_icp(p["source"].cpu().numpy(), p["target"].cpu().numpy(), 2 * s, None, 2, p["normals"].cpu().numpy(),
                                  rel_fitness=0.0, rel_rmse=0.0, max_iter=30)
    assert np.array_equal(corr[2].cpu().numpy(), ref.corr)
    assert np.abs(_lib.from_colmajor(T[2]) - ref.transformation).max() < 1e-6


@pytest.mark.parametrize("est_name", ["p2plane", "p2p"])
def test_batch_gangs_of_unequal_pairs_equal_one_by_one(ctx, oracle, est_name):
    """The lock-step form of the batch (o3d_launch_pass_gang) with pairs of very different sizes in one gang -- every slot
    brings its own number of workgroups to a launch sized for the largest, the others' surplus workgroups leave at once --
    an empty source among them, pairs that converge at different iterations (default criteria) and more pairs than one gang
    holds: against the same batch one registration at a time (O3D_BATCH_GANG=1), correspondences bit for bit, the
    transformation to rounding; and two of them against the oracle."""
    import os
    geometry, reg = _reg()
    sizes = [(60_000, 50_000), (150, 900), (9_000, 9_000), (0, 2_000), (33_000, 20_000), (700, 700), (20_000, 45_000), (2_500, 2_000),
             (12_000, 12_500), (64, 4_000), (5_000, 5_100)]
    sources, targets = [], []
    for k, (ns, nt) in enumerate(sizes):
        sc = make_scene(max(ns, 1), nt, cfg_id=700 + k, L=3.0, theta_deg=0.2 + 0.1 * (k % 4))
        nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 0.2)
        sources.append(geometry.PointCloud(sc["source"][:ns]))
        targets.append(geometry.PointCloud(sc["target"], normals=nrm))
    r = 0.1
    est = reg.TransformationEstimationPointToPlane() if est_name == "p2plane" else reg.TransformationEstimationPointToPoint()
    for crit in (reg.ICPConvergenceCriteria(), reg.ICPConvergenceCriteria(0.0, 0.0, 9)):
        os.environ["O3D_BATCH_GANG"] = "4"        # 11 pairs over 2 threads: gangs of 4, 4 and 3
        try:
            got = reg.RegistrationICPBatch(sources, targets, r, None, est, crit, n_workers=2, want_corr=True, ctx=ctx)
            os.environ["O3D_BATCH_GANG"] = "1"
            want = reg.RegistrationICPBatch(sources, targets, r, None, est, crit, n_workers=2, want_corr=True, ctx=ctx)
        finally:
            os.environ.pop("O3D_BATCH_GANG", None)
        for k in range(len(sizes)):
            assert got[k].iterations_ == want[k].iterations_
            assert np.array_equal(got[k].corr_, want[k].corr_)
            assert np.abs(got[k].transformation_ - want[k].transformation_).max() < 1e-11
            assert got[k].fitness_ == want[k].fitness_ and abs(got[k].inlier_rmse_ - want[k].inlier_rmse_) < 1e-13
        for k in (0, 6):
            ref = oracle.registration_icp(sources[k].points_, targets[k].points_, r, None, 2 if est_name == "p2plane" else 1,
                                          targets[k].normals_, rel_fitness=crit.relative_fitness_, rel_rmse=crit.relative_rmse_,
                                          max_iter=crit.max_iteration_)
            assert got[k].iterations_ == ref.iterations and np.array_equal(got[k].corr_, ref.corr)
            assert rel_err(got[k].transformation_, ref.transformation) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("est_name", ["p2p", "p2plane"])
def test_batch_common_sort_equals_sorts_per_pair(ctx, oracle, est_name):
    """A gang's set-ups leave their two radix sorts to ONE sort over the whole gang's keys (o3d_gang_sort: job number in the
    key's top bits; the sort is stable, so every pair's segment comes out as its own sort would have left it).  The same
    batch with O3D_NO_GANG_SORT=1 must agree in every bit -- with non-finite points in a pair, an empty source, and a pair
    whose cell keys do not fit the common key (it sorts for itself inside the gang)."""
    import os
    geometry, reg = _reg()
    sizes = [(30_000, 28_000), (400, 2_000), (0, 1_500), (18_000, 18_000), (7_000, 9_000), (2_000, 2_000), (25_000, 11_000)]
    sources, targets = [], []
    for k, (ns, nt) in enumerate(sizes):
        sc = make_scene(max(ns, 1), nt, cfg_id=810 + k, L=60.0 if k == 4 else 3.0, theta_deg=0.3)
        src, tgt = sc["source"][:ns].copy(), sc["target"].copy()
        nrm = oracle.estimate_normals(tgt, oracle.HYBRID, 30, 0.2 if k != 4 else 4.0)
        if k == 3:  # non-finite points: skipped by the search, last in their segment of the sort
            src[5] = np.nan
            src[1000, 1] = np.inf
            tgt[7] = np.inf
        sources.append(geometry.PointCloud(src))
        targets.append(geometry.PointCloud(tgt, normals=nrm))
    est = reg.TransformationEstimationPointToPlane() if est_name == "p2plane" else reg.TransformationEstimationPointToPoint()
    crit = reg.ICPConvergenceCriteria(0.0, 0.0, 8)
    reg.RegistrationICP(sources[4], targets[4], 0.1, np.eye(4), est, crit, ctx=ctx)
    info = ctx.grid_info()
    assert max(info["nx"], info["ny"], info["nz"]) > 256, info  # 27 and more key bits for the source: over the common key
    os.environ["O3D_BATCH_GANG"] = "4"
    try:
        got = reg.RegistrationICPBatch(sources, targets, 0.1, None, est, crit, n_workers=2, want_corr=True, ctx=ctx)
        os.environ["O3D_NO_GANG_SORT"] = "1"
        want = reg.RegistrationICPBatch(sources, targets, 0.1, None, est, crit, n_workers=2, want_corr=True, ctx=ctx)
    finally:
        os.environ.pop("O3D_BATCH_GANG", None)
        os.environ.pop("O3D_NO_GANG_SORT", None)
    for k in range(len(sizes)):
        assert got[k].iterations_ == want[k].iterations_
        assert np.array_equal(got[k].corr_, want[k].corr_)
        assert np.array_equal(got[k].transformation_, want[k].transformation_)
        assert got[k].fitness_ == want[k].fitness_ and got[k].inlier_rmse_ == want[k].inlier_rmse_
    ref = oracle.registration_icp(sources[0].points_, targets[0].points_, 0.1, None, 2 if est_name == "p2plane" else 1,
                                  targets[0].normals_, rel_fitness=0.0, rel_rmse=0.0, max_iter=8)
    assert np.array_equal(got[0].corr_, ref.corr) and rel_err(got[0].transformation_, ref.transformation) < TOL

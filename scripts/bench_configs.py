#!/usr/bin/env python3
"""Timings of the BASELINE configs other than the headline, for DESIGN.md -- not the bench.py line.
Prints one JSON object per config.  Host buffers (PCIe included) unless stated."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open3d_amd import _lib, geometry, synthetic
from open3d_amd.pipelines import registration as reg

ctx = _lib.Context(0)
dev = torch.device("cuda", 0)


def sync():
    torch.cuda.synchronize()


def timed(fn, reps=1):
    fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    sync()
    return (time.perf_counter() - t0) / reps, out


def device_loop(src, tgt, nrm, r, est, iters):
    """iterations/s of the device-resident loop with inputs resident in HBM."""
    d = [torch.from_numpy(a).to(dev) if a is not None else None for a in (src, tgt, nrm)]
    sess = reg.IcpSession(ctx).setup(d[0], d[1], d[2], r, np.eye(4), est, reg.ICPConvergenceCriteria(0, 0, iters + 4))
    for _ in range(4):
        sess.enqueue_step()
    sess.poll()
    t0 = time.perf_counter()
    for _ in range(iters):
        sess.enqueue_step()
    sess.poll()
    dt = time.perf_counter() - t0
    return iters / dt, sess.setup_ms()


which = sys.argv[1:] or ["1", "2", "3", "5"]

if "1" in which:  # config 1: point-to-point, 100k x 100k
    sc = synthetic.make_pair(100_000, 100_000, cfg_id=1, theta_deg=0.05, tau_in_spacings=(1.0, -0.7, 0.5))
    s = sc["spacing"]
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    dt, res = timed(lambda: reg.RegistrationICP(src, tgt, 3 * s, np.eye(4), reg.TransformationEstimationPointToPoint(), ctx=ctx), 5)
    ips, _ = device_loop(sc["source"], sc["target"], None, 3 * s, reg.TransformationEstimationPointToPoint(), 200)
    print(json.dumps({"config": 1, "what": "P2P 100k x 100k, r = 3 s, default criteria, whole call from host buffers",
                      "call_ms": 1e3 * dt, "iterations": res.iterations_, "fitness": res.fitness_,
                      "device_loop_iters_per_s": ips}), flush=True)

if "2" in which:  # config 2: point-to-plane 1M, normals from EstimateNormals(Hybrid(4 s, 30))
    sc = synthetic.make_pair(1_000_000, 1_000_000, cfg_id=2)
    s = sc["spacing"]
    tgt = geometry.PointCloud(sc["target"])
    dtn, _ = timed(lambda: tgt.EstimateNormals(geometry.KDTreeSearchParamHybrid(4 * s, 30), ctx=ctx))
    src = geometry.PointCloud(sc["source"])
    dt, res = timed(lambda: reg.RegistrationICP(src, tgt, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(), ctx=ctx), 3)
    ips, setup = device_loop(sc["source"], sc["target"], tgt.normals_, 2 * s, reg.TransformationEstimationPointToPlane(), 200)
    print(json.dumps({"config": 2, "what": "P2Plane 1M x 1M, r = 2 s", "estimate_normals_ms": 1e3 * dtn, "call_ms_default_criteria": 1e3 * dt,
                      "iterations": res.iterations_, "device_loop_iters_per_s": ips, "grid_build_ms": float(setup[0]),
                      "source_sort_ms": float(setup[1])}), flush=True)

if "3" in which:  # config 3 pipeline: 10M raw -> VoxelDownSample(2 s) both -> EstimateNormals -> P2Plane ICP r = 2 v
    sc = synthetic.make_pair(10_000_000, 10_000_000, cfg_id=3)
    s = sc["spacing"]
    v = 2 * s
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    for rep in range(2):  # the second run is reported: the first one creates the context's staging buffers and streams
        t0 = time.perf_counter()
        sv = src.VoxelDownSample(v, ctx=ctx)
        tv = tgt.VoxelDownSample(v, ctx=ctx)
        t1 = time.perf_counter()
        tv.EstimateNormals(geometry.KDTreeSearchParamHybrid(2 * v, 30), ctx=ctx)
        t2 = time.perf_counter()
        res = reg.RegistrationICP(sv, tv, 2 * v, np.eye(4), reg.TransformationEstimationPointToPlane(), ctx=ctx)
        t3 = time.perf_counter()
        if rep == 0:
            first_ms = 1e3 * (t3 - t0)
    print(json.dumps({"config": 3, "what": "10M raw -> VoxelDownSample(2 s) x2 -> EstimateNormals(Hybrid(2 v, 30)) -> P2Plane ICP r = 2 v; "
                                           "host buffers, PCIe included",
                      "voxel_ms_both": 1e3 * (t1 - t0), "n_after": [len(sv.points_), len(tv.points_)], "normals_ms": 1e3 * (t2 - t1),
                      "icp_ms": 1e3 * (t3 - t2), "iterations": res.iterations_, "fitness": res.fitness_, "total_ms": 1e3 * (t3 - t0),
                      "first_run_total_ms": first_ms}),
          flush=True)

if "3h" in which:  # config 3 pipeline from host buffers through device-resident clouds: one upload per raw cloud
    sc = synthetic.make_pair(10_000_000, 10_000_000, cfg_id=3)
    s = sc["spacing"]
    v = 2 * s
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        fin, per = reg.MultiScaleICP(src, tgt, [v], ctx=ctx)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(json.dumps({"config": "3 (host buffers, device-resident clouds)", "what": "10M raw x2 uploaded once -> VoxelDownSample(2 s) x2 -> "
                      "EstimateNormals(Hybrid(2 v, 30)) -> P2Plane ICP r = 2 v, default criteria; results (T, correspondences) to the host",
                      "total_ms_per_repetition": ts, "n_after": [per[0].n_source_, per[0].n_target_], "iterations": fin.iterations_,
                      "fitness": fin.fitness_}), flush=True)

if "5" in which:  # config 5 in part: 64 pairs x 200k, P2Plane, criteria (0, 0, 30)
    npairs, n = 64, 200_000
    S, T_ = [], []
    for k in range(npairs):
        sc = synthetic.make_pair(n, n, cfg_id=1000 + k, L=2.83)
        s = sc["spacing"]
        nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
        S.append(geometry.PointCloud(sc["source"]))
        T_.append(geometry.PointCloud(sc["target"], normals=nrm))
    crit = reg.ICPConvergenceCriteria(0.0, 0.0, 30)
    est = reg.TransformationEstimationPointToPlane()
    out = {}
    for w in (1, 4, 8):
        dt, res = timed(lambda: reg.RegistrationICPBatch(S, T_, 2 * s, None, est, crit, n_workers=w, ctx=ctx))
        out["pairs_per_s_workers_%d" % w] = npairs / dt
    print(json.dumps({"config": 5, "what": "64 of the 4096 pairs x 200k points, P2Plane, r = 2 s, criteria (0,0,30); host buffers "
                                           "(2 x 4.8 MB + normals per pair over PCIe), analytic normals", **out,
                      "iterations": res[0].iterations_}), flush=True)

if "4" in which:  # config 4 on ONE GPU: multi-scale ICP (voxel 0.05 / 0.025 / 0.0125 m) on a 100M-point cloud (L = 63 m)
    n = 100_000_000
    t0 = time.perf_counter()
    sc = synthetic.make_pair(n, n, cfg_id=4, theta_deg=0.05, tau_in_spacings=(4.0, -3.0, 2.0))
    print("# config 4: scene generated in %.0f s" % (time.perf_counter() - t0), flush=True)
    src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"])
    crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 30)
    t0 = time.perf_counter()
    final, per = reg.MultiScaleICP(src, tgt, [0.05, 0.025, 0.0125], criteria=crit, ctx=ctx)
    dt = time.perf_counter() - t0
    err = np.abs(final.transformation_ - sc["T_true"]).max()
    print(json.dumps({"config": 4, "what": "multi-scale P2Plane ICP, voxel 0.05/0.025/0.0125 m, 100M x 100M raw points (L = 63 m, spacing "
                                           "6.3 mm), one GPU, host buffers (PCIe included), criteria (1e-6, 1e-6, 30) per scale",
                      "total_s": dt, "per_scale": [{"n_source": r.n_source_, "n_target": r.n_target_, "iterations": r.iterations_,
                                                    "fitness": r.fitness_, "rmse": r.inlier_rmse_} for r in per],
                      "max_abs_T_error_vs_truth": float(err)}), flush=True)

if "3d" in which:  # config 3 pipeline with every buffer resident in HBM (torch tensors through the C ABI, mem = device)
    import ctypes as C
    sc = synthetic.make_pair(10_000_000, 10_000_000, cfg_id=3)
    s = sc["spacing"]
    v = 2 * s
    d_src, d_tgt = (torch.from_numpy(a).to(dev) for a in (sc["source"], sc["target"]))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def voxel(d_xyz):
        n = d_xyz.shape[0]
        out = torch.empty((n, 3), dtype=torch.float64, device=dev)
        vox = torch.empty((n, 3), dtype=torch.int32, device=dev)
        m = C.c_int64()
        ctx.check(ctx._lib.o3d_voxel_down_sample(ctx.handle, _lib.ptr(d_xyz), None, None, n, float(v), _lib.MEM_DEVICE, _lib.ptr(out),
                                                 None, None, _lib.ptr(vox), C.byref(m)))
        return out[: m.value].contiguous()

    def pipeline():
        sv, tv = voxel(d_src), voxel(d_tgt)
        nrm = torch.zeros_like(tv)
        sp = geometry.KDTreeSearchParamHybrid(2 * v, 30)
        ctx.check(ctx._lib.o3d_estimate_normals(ctx.handle, _lib.ptr(tv), tv.shape[0], sp.search_type, sp.knn_, sp.radius_,
                                                _lib.MEM_DEVICE, _lib.ptr(nrm), 0))
        sess = reg.IcpSession(ctx).setup(sv, tv, nrm, 2 * v, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                         reg.ICPConvergenceCriteria())
        sess.run()
        corr = torch.empty(sv.shape[0], dtype=torch.int32, device=dev)
        return sess.finish(corr_out=corr), sv.shape[0], tv.shape[0]

    dt, (res, ns, nt) = timed(pipeline, 3)
    print(json.dumps({"config": "3 (device resident)", "what": "10M raw -> VoxelDownSample(2 s) x2 -> EstimateNormals(Hybrid(2 v, 30)) -> "
                      "P2Plane ICP r = 2 v, default criteria; all buffers in HBM (mem = device through the C ABI)",
                      "total_ms": 1e3 * dt, "n_after": [ns, nt], "iterations": res.iterations_, "fitness": res.fitness_}), flush=True)

#!/usr/bin/env python3
"""Headline benchmark: point-to-plane ICP on a 10M-point synthetic cloud (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one ICP iteration of the hot path over the whole source cloud: pending update applied in place, exact
radius-bounded nearest neighbour for every source point against the target grid, the 27+2 fp64 sums of the next
point-to-plane estimate, and the 6x6 solve -- Registration.cpp:155-166 of the reference.  Inputs are resident in HBM and
the target grid is built before the timed region (build time is reported in config).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def cpu_baseline(scene, normals, radius, n_sample, iters, est_type=2, serial=False):
    """Oracle (FLANN-shaped KD-tree restatement, OpenMP) on a bounded sample of the same workload.  Returns the baseline
    record and the oracle's result (per-pass log, final correspondences) for the parity check."""
    import oracle
    oracle.build()
    threads = 1 if serial else oracle.max_threads()
    src = scene["source"][:n_sample]
    t0 = time.time()
    res = oracle.registration_icp(src, scene["target"], radius, None, est_type, normals, rel_fitness=0.0, rel_rmse=0.0, max_iter=iters,
                                  want_log=True, serial=serial)
    wall = time.time() - t0
    n_full = len(scene["source"])
    loop_s = res.loop_seconds  # first LOOP-A pass .. last, KD-tree build excluded (as for the GPU figure)
    per_point_pass = loop_s / ((iters + 1) * n_sample)
    iters_per_s_full = 1.0 / (per_point_pass * n_full * (iters + 1) / iters)
    return {
        "value": iters_per_s_full,
        "unit": "iter/s",
        "cores": threads,
        "kind": "port",
        "sample": "KD-tree (leaf 15) over the full %d-point target; %d ICP iterations (%d correspondence passes) over the "
                  "first %d source points, %d OpenMP threads; per-point time scaled to %d source points; "
                  "kd build + loop wall %.1f s" % (len(scene["target"]), iters, iters + 1, n_sample, threads, n_full, wall),
        "corr_per_s": 1.0 / per_point_pass,
    }, res


def parity_check(ctx, reg, d_src, d_tgt, d_nrm, radius, est, iters, n_sample, ref):
    """The first `iters` iterations of the same registration on the GPU against the oracle run the CPU baseline has just
    timed: the correspondence count of every pass identical, fitness / rmse / transformation of every pass within 1e-6,
    the final correspondence array bit-exact on every point.  Returns a one-line summary; raises SystemExit on a mismatch."""
    sess = reg.IcpSession(ctx).setup(d_src[:n_sample], d_tgt, d_nrm, radius, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, iters))
    sess.run()
    got = sess.finish()
    log = sess.log()

    def rel(a, b):
        a, b = np.asarray(a, float), np.asarray(b, float)
        return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
    if got.iterations_ != ref.iterations or len(log) != len(ref.log):
        raise SystemExit("parity: %d iterations / %d passes on the GPU, %d / %d in the oracle" % (got.iterations_, len(log), ref.iterations, len(ref.log)))
    worst = 0.0
    for p in range(len(log)):
        if log[p, 2] != ref.log[p, 2]:
            raise SystemExit("parity: pass %d has %d correspondences on the GPU, %d in the oracle" % (p, log[p, 2], ref.log[p, 2]))
        worst = max(worst, rel(log[p, 0], ref.log[p, 0]), rel(log[p, 1], ref.log[p, 1]), rel(log[p, 4:20], ref.log[p, 4:20]))
    worst = max(worst, rel(got.transformation_, ref.transformation), rel(got.inlier_rmse_, ref.inlier_rmse))
    bad = int(np.count_nonzero(got.corr_ != ref.corr))
    if bad or not worst < 1e-6:
        raise SystemExit("parity: %d of %d final correspondences differ from the oracle, worst relative difference of "
                         "fitness / rmse / transformation %.3g" % (bad, n_sample, worst))
    return ("ok: %d iterations (%d passes) over %d source points against the oracle: correspondence count identical in every pass, "
            "final correspondences identical on all %d points, fitness / rmse / transformation within %.1e (bar 1e-6)"
            % (iters, iters + 1, n_sample, n_sample, worst))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--estimation", default="p2plane", choices=["p2plane", "p2p"])
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--cpu-sample-1t", type=int, default=2_000_000, help="source points of the one-thread CPU baseline leg")
    ap.add_argument("--no-hard-scene", dest="hard_scene", action="store_false", help="skip the hard-scene leg (config.hard_scene)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks", type=int, default=5,
                    help="the timed region is repeated this many times back to back (value = median block), steps iterations each")
    ap.add_argument("--search", default="f32", choices=["f32", "f64"],
                    help="f32: fp32 prefilter search + fp64 confirmation + exact fixup (default); f64: per-lane fp64 search")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearse N ranks on fewer GPUs (ranks share devices, sums exchanged through the host): correctness "
                         "of the N > 1 path only, its timings mean nothing")
    ap.add_argument("--timed-only", action="store_true",
                    help="only the rehearsal and the timed registrations (no steady-state, default-criteria, fresh-30 or hard-scene "
                         "legs): under rocprofv3 --kernel-trace --stats the average k_icp_pass duration is then roofline.kernel_ms")
    ap.add_argument("--force-exchange", action="store_true",
                    help="with one rank under torch.distributed.run: still run pass -> RCCL all-gather -> solve (rehearsal of N > 1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dist_backend == "gloo":  # rehearsal of several ranks on fewer GPUs: ranks share devices, the exchange goes through the host
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1 or (args.force_exchange and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: open3d_amd has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from open3d_amd import _lib, synthetic
    from open3d_amd.pipelines import registration as reg
    from open3d_amd.sharded import HipEngine, ShardedICP, shard_bounds

    n = args.points
    scene = synthetic.make_pair(n, n, cfg_id=3)  # config 3 headline: 10 M / 10 M, L = 20 m, spacing ~6.3 mm
    s = scene["spacing"]
    radius = 2.0 * s if args.estimation == "p2plane" else 3.0 * s
    ctx = _lib.Context(local_rank)
    # target normals from the library's own EstimateNormals (hybrid search, radius 4 spacings, max_nn 30: config 2/3 of
    # BASELINE.json); preprocessing, outside the timed ICP loop, its time is reported in config
    lo, hi = shard_bounds(n, world, rank)
    d_src = torch.from_numpy(scene["source"][lo:hi]).to(dev)
    d_tgt = torch.from_numpy(scene["target"]).to(dev)
    torch.cuda.synchronize()
    normals, normals_ms, normals_dev_ms, d_nrm = None, 0.0, 0.0, None
    if args.estimation == "p2plane":
        # the host-buffer call (the drop-in boundary: upload, grid, kernel, download into a fresh array) ...
        from open3d_amd import geometry
        pc = geometry.PointCloud(scene["target"])
        t0 = time.perf_counter()
        pc.EstimateNormals(geometry.KDTreeSearchParamHybrid(4.0 * s, 30), ctx=ctx)
        normals_ms = 1e3 * (time.perf_counter() - t0)
        normals = pc.normals_
        # ... and the same call on buffers resident in HBM, like the ICP loop below (grid build + kernel); bit-identical
        d_nrm = torch.empty((n, 3), dtype=torch.float64, device=dev)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.check(ctx._lib.o3d_estimate_normals(ctx.handle, d_tgt.data_ptr(), n, _lib.SEARCH_HYBRID, 30, 4.0 * s, _lib.MEM_DEVICE,
                                                    d_nrm.data_ptr(), 0))
            torch.cuda.synchronize()
            normals_dev_ms = 1e3 * (time.perf_counter() - t0)
        if not bool((d_nrm.cpu() == torch.from_numpy(normals)).all()):
            raise SystemExit("EstimateNormals: device-resident and host-buffer calls disagree")

    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.debug_set(use_tile_search=(args.search == "f32"))
    engine = HipEngine(ctx)
    icp = ShardedICP(engine, None, dev, force_exchange=args.force_exchange)
    est = reg.TransformationEstimationPointToPlane() if args.estimation == "p2plane" else reg.TransformationEstimationPointToPoint()
    blocks = max(1, args.blocks)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def fresh_registration(src_t, tgt_t, nrm_t, r, k_iters, n_global, timed_profile=True):
        """One registration from scratch with ICPConvergenceCriteria(0, 0, k_iters): set-up (grid build, source sort: not
        timed, reported separately), then t_loop = the first correspondence pass (Registration.cpp:154) through
        the last of the k_iters iterations (:155-166), bracketed by barrier + synchronize.  Returns (t_loop seconds,
        kernel ms of every pass by HIP events, set-up ms)."""
        icp.setup(src_t, tgt_t, nrm_t, r, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, k_iters), n_global)
        sms = engine.sess.setup_ms()
        barrier()
        engine.sess.profile(bool(timed_profile))
        t0 = time.perf_counter()
        icp.step()                      # pass 0 and the first estimate
        for _ in range(k_iters):
            icp.step()
        barrier()
        t = time.perf_counter() - t0
        ms = engine.sess.profile_read_all() if timed_profile else np.zeros(0)
        engine.sess.profile(False)
        return t, ms, sms

    # W untimed warm-up steps (a rehearsal registration: code objects, caches, clocks), then the timed region: `blocks`
    # registrations from scratch of K iterations each.  value = K / median t_loop  (iters/s = M / t_loop).
    fresh_registration(d_src, d_tgt, d_nrm, radius, max(args.warmup, 1), n, timed_profile=False)
    block_s, block_ms, setup_ms = [], [], None
    for _ in range(blocks):
        t, ms, setup_ms = fresh_registration(d_src, d_tgt, d_nrm, radius, args.steps, n)
        block_s.append(t)
        block_ms.append(ms)
    res = icp.finish(want_corr=False)
    if dist.is_initialized():
        t = torch.tensor(block_s, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        block_s = [float(v) for v in t.cpu()]
    dt = float(np.median(block_s))
    n_corr_global = res.n_corr_
    fitness = res.fitness_
    per_pass = np.median(np.stack(block_ms), axis=0) if len(block_ms[0]) == args.steps + 1 else np.zeros(args.steps + 1)
    pass0_ms = float(per_pass[0])                 # k_icp_search_coop over everything (nothing to certify yet)
    kern_ms = float(np.mean(per_pass[1:]))        # k_icp_pass, averaged over the K launches of a block
    kern_n = blocks * args.steps
    one_launch = args.search == "f32"

    # The converged regime for comparison with earlier rounds (their `value`): the same registration kept going, K more
    # iterations timed once its first W + 3 have passed (every match certified, no search).
    steady = None
    if not args.timed_only:
        total = args.warmup + 3 + args.steps
        icp.setup(d_src, d_tgt, d_nrm, radius, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, total), n)
        for _ in range(1 + args.warmup + 3):
            icp.step()
        barrier()
        engine.sess.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            icp.step()
        barrier()
        st = time.perf_counter() - t0
        sms_ = engine.sess.profile_read_all()
        engine.sess.profile(False)
        if dist.is_initialized():
            tt = torch.tensor([st], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            st = float(tt.cpu()[0])
        steady = {"iters_per_s": args.steps / st, "ms_per_step": 1e3 * st / args.steps,
                  "pass_kernel_ms": float(np.median(sms_)) if len(sms_) else None,
                  "what": "iterations %d..%d of one registration with ICPConvergenceCriteria(0,0,%d): the source no longer moves, "
                          "every previous match is certified (rounds 1-2 reported this as `value`)" % (args.warmup + 4, total, total)}
        icp.finish(want_corr=False)

    # What a caller waits for with the reference's default criteria (1e-6, 1e-6, 30), wall-timed around set-up -> loop ->
    # results with every buffer resident in HBM (the PCIe-inclusive figure for host buffers is in DESIGN.md).
    def default_call(src_t, tgt_t, nrm_t, r):
        corr = torch.empty(src_t.shape[0], dtype=torch.int32, device=dev)
        barrier()
        t0 = time.perf_counter()
        icp.setup(src_t, tgt_t, nrm_t, r, np.eye(4), est, reg.ICPConvergenceCriteria(), n)
        icp.run()
        r_ = engine.sess.finish(corr_out=corr)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0), r_
    default_ms, default_res, fresh30_ms = None, None, None
    if not args.timed_only:
        default_call(d_src, d_tgt, d_nrm, radius)
        default_ms, default_res = default_call(d_src, d_tgt, d_nrm, radius)
    if args.timed_only:
        pass
    elif args.steps != 30:
        t30 = [fresh_registration(d_src, d_tgt, d_nrm, radius, 30, n, timed_profile=False)[0] for _ in range(3)]
        fresh30_ms = 1e3 * float(np.median(t30))
    else:
        fresh30_ms = 1e3 * dt

    # A scene that keeps moving: rotation 1 degree, translation 6 spacings, search radius 4 spacings -- most
    # passes search a large part of the cloud instead of certifying it.  Generated on the device (same generator).
    hard = None
    if world == 1 and args.hard_scene and args.estimation == "p2plane" and not args.timed_only:
        hs = synthetic.make_pair_device(ctx, n, n, cfg_id=33, theta_deg=1.0, tau_in_spacings=(4.0, -3.5, 2.9), want_normals=False)
        h_nrm = torch.empty((n, 3), dtype=torch.float64, device=dev)
        ctx.check(ctx._lib.o3d_estimate_normals(ctx.handle, hs["target"].data_ptr(), n, _lib.SEARCH_HYBRID, 30, 4.0 * s, _lib.MEM_DEVICE,
                                                h_nrm.data_ptr(), 0))
        hr = 4.0 * s
        fresh_registration(hs["source"], hs["target"], h_nrm, hr, 30, n, timed_profile=False)
        ht, hms = [], []
        for _ in range(3):
            t, ms, _sm = fresh_registration(hs["source"], hs["target"], h_nrm, hr, 30, n)
            ht.append(t)
            hms.append(ms)
        hres = icp.finish(want_corr=False)
        hpp = np.median(np.stack(hms), axis=0)
        # share of the queries certified without a search in every pass (statistics on: an untimed run, polled per pass)
        ctx.debug_set(use_tile_search=True, collect_tile_stats=True)
        icp.setup(hs["source"], hs["target"], h_nrm, hr, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, 30), n)
        ctx.tile_stats(reset=True)
        cert_frac = []
        for _ in range(31):
            icp.step()
            cert_frac.append(round(ctx.tile_stats(reset=True)["certified"] / n, 4))
        icp.finish(want_corr=False)
        ctx.debug_set(use_tile_search=(args.search == "f32"), collect_tile_stats=False)
        hbytes = n * 52 + hres.fitness_ * n * 48
        hdt = float(np.median(ht))
        hdef_ms, hdef = default_call(hs["source"], hs["target"], h_nrm, hr)
        hard = {"what": "10 M x 10 M point-to-plane, rotation 1 deg about (1,2,3), translation (4,-3.5,2.9) spacings (|t| = 6.1), "
                        "max_correspondence_distance = 4 spacings, ICPConvergenceCriteria(0,0,30), registration from scratch",
                "iters_per_s": 30 / hdt, "t_loop_ms": 1e3 * hdt, "fitness": hres.fitness_, "inlier_rmse": hres.inlier_rmse_,
                "max_abs_T_error_vs_truth": float(np.abs(hres.transformation_ - hs["T_true"]).max()),
                "kernel_ms_per_pass": [round(float(v), 4) for v in hpp],
                "certified_fraction_per_pass": cert_frac,
                "frac_of_hbm_roofline": hbytes * 31 / (float(hpp.sum()) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "default_criteria_call_ms": hdef_ms, "default_criteria_iterations": hdef.iterations_}
        del hs, h_nrm

    if rank == 0:
        iters_per_s = args.steps / dt
        corr_per_s = (args.steps + 1) * n / dt
        # algorithmic bytes of one pass (one k_icp_pass launch) on this rank: per source point 24 B read + 24 B
        # written (in-place transform) + 4 B correspondence; per correspondence 24 B target point + 24 B normal
        n_shard = hi - lo
        nc_shard = fitness * n_shard
        per_corr = 48 if args.estimation == "p2plane" else 24
        bytes_launch = n_shard * 52 + nc_shard * per_corr
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": "icp_iterations_per_sec_%s_%dM" % (args.estimation, n // 1_000_000) if n >= 1_000_000 else
                      "icp_iterations_per_sec_%s_%d" % (args.estimation, n),
            "value": iters_per_s,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "correspondences_per_sec": corr_per_s,
            "config": {
                "workload": "config 3 headline%s: %s ICP, %d source x %d target points on the synthetic height "
                            "field (L=%.1f m, spacing %.2f mm), max_correspondence_distance = %d spacings, "
                            "ICPConvergenceCriteria(0,0,%d), source sharded over %d GPU(s), target grid replicated"
                            % ("" if args.estimation == "p2plane" else " size, other estimation",
                               "point-to-plane" if args.estimation == "p2plane" else "point-to-point", n, n, scene["L"], 1e3 * s,
                               2 if args.estimation == "p2plane" else 3, args.steps, world),
                "timed_region": "value = K / t_loop of a registration FROM SCRATCH (SURVEY 8d): after %d untimed warm-up steps of a "
                                "rehearsal registration, %d registrations are set up anew (grid build and source sort untimed, "
                                "reported below) and each is timed from its first correspondence pass (Registration.cpp:154) "
                                "through the last of its K = %d iterations -- K + 1 passes, the first two of which search the whole "
                                "cloud; median over the registrations" % (max(args.warmup, 1), blocks, args.steps),
                "estimation": args.estimation,
                "fitness": fitness,
                "transformation_rowmajor": [float(v) for v in np.asarray(res.transformation_).reshape(-1)],
                "inlier_rmse": res.inlier_rmse_,
                "n_corr": n_corr_global,
                "grid_build_ms": float(setup_ms[0]),
                "source_sort_ms": float(setup_ms[1]),
                "estimate_normals_ms_incl_pcie": normals_ms,
                "estimate_normals_ms_device_resident": normals_dev_ms,
                "blocks": blocks,
                "iters_per_s_min_median_max": [args.steps / max(block_s), iters_per_s, args.steps / min(block_s)],
                "kernel_ms_per_pass": [round(float(v), 4) for v in per_pass],
                "fresh_call_0_0_30_ms": fresh30_ms,
                "fresh_call_0_0_30_iters_per_s": 30.0 / (fresh30_ms * 1e-3) if fresh30_ms else None,
                "default_criteria_call_ms": default_ms,
                "default_criteria_call": "ICPConvergenceCriteria(1e-6, 1e-6, 30): %d iterations; wall time of set-up (grid build, source "
                                         "sort) + loop + results, buffers resident in HBM" % (default_res.iterations_ if default_res else -1),
                "steady_state": steady,
                "hard_scene": hard,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_icp_pass: ONE launch per ICP pass -- every wavefront streams its segment of the source (update in place, "
                           "certified previous matches, MFMA sums), searches what it could not certify, and the last workgroups to "
                           "arrive add the partial rows and solve the 6x6 system.  Average over the K launches of a timed registration "
                           "(the first of them searches the whole cloud, the rest certify); HIP events on the launch stream")
                          if args.search == "f32" else "k_icp_search + k_icp_accumulate_mfma (all-fp64 path)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None,
                "kernel_ms": kern_ms,
                "launches_timed": kern_n,
                "algorithmic_bytes_per_launch": bytes_launch,
                # the other kernel of the loop: the initial correspondence pass (nothing to certify: searches everything)
                "search_pass": {"kernel": "k_icp_search_coop (pass 0 of a registration)", "kernel_ms": pass0_ms,
                                "achieved": bytes_launch / (pass0_ms * 1e-3) / 1e9 if pass0_ms > 0 else None,
                                "frac": bytes_launch / (pass0_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if pass0_ms > 0 else None},
                "steady_state_frac": (bytes_launch / (steady["pass_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS)
                                     if steady and steady.get("pass_kernel_ms") else None,
            },
        }
        # HBM traffic per k_icp_pass launch from the PMC passes of this same command (scripts/prof_r03.sh; counters cannot be
        # read from inside the process): 2 x FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads; applied to all
        # reads, so an upper bound) + WRITE_SIZE, averaged like `achieved` over the K launches of a registration -- one that
        # searches the whole cloud, K - 1 that certify.  Only quoted for the workload it was collected on.
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and n == 10_000_000 and args.estimation == "p2plane" and args.search == "f32" and os.path.exists(tpath):
            with open(tpath) as f:
                tr = json.load(f)
            per = {k: (2.0 * v["fetch_size_kb"] + v["write_size_kb"]) * 1024.0 for k, v in tr["k_icp_pass"].items()}
            out["roofline"]["traffic"] = (per["search_all"] + (args.steps - 1) * per["steady"]) / args.steps
            out["roofline"]["traffic_per_kind"] = per
            out["roofline"]["search_pass"]["traffic"] = (2.0 * tr["k_icp_search_coop"]["fetch_size_kb"]
                                                         + tr["k_icp_search_coop"]["write_size_kb"]) * 1024.0
            if "exact_request_bytes" in tr:  # the same passes counted by request size (no correction factor involved)
                ex = tr["exact_request_bytes"]
                per_x = {k: ex["k_icp_pass_" + k]["read_bytes"] + ex["k_icp_pass_" + k]["write_bytes"] for k in ("search_all", "steady")}
                out["roofline"]["traffic_by_request_size"] = (per_x["search_all"] + (args.steps - 1) * per_x["steady"]) / args.steps
            out["roofline"]["traffic_source"] = tr["source"]
        if world == 1 and not args.no_cpu_baseline:
            n_sample = min(args.cpu_sample, n)
            est_type = 2 if args.estimation == "p2plane" else 1
            out["cpu_baseline"], ref = cpu_baseline(scene, normals, radius, n_sample, args.cpu_iters, est_type)
            # ... and with one thread (the reference's own CMakeLists.txt builds without OpenMP): a smaller sample, scaled
            out["cpu_baseline"]["single_thread"] = cpu_baseline(scene, normals, radius, min(args.cpu_sample_1t, n), 1, est_type,
                                                                serial=True)[0]
            # the oracle run is not thrown away: the GPU must reproduce it (checker only -- nothing of it is timed or shipped)
            out["config"]["parity_%s" % ("%dM" % (n // 1_000_000) if n >= 1_000_000 else str(n))] = parity_check(
                ctx, reg, d_src, d_tgt, d_nrm, radius, est, args.cpu_iters, n_sample, ref)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

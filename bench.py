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


def cpu_baseline(scene, normals, radius, n_sample, iters, est_type=2):
    """Oracle (FLANN-shaped KD-tree restatement, OpenMP) on a bounded sample of the same workload.  Returns the baseline
    record and the oracle's result (per-pass log, final correspondences) for the parity check."""
    import oracle
    oracle.build()
    threads = oracle.max_threads()
    src = scene["source"][:n_sample]
    t0 = time.time()
    res = oracle.registration_icp(src, scene["target"], radius, None, est_type, normals, rel_fitness=0.0, rel_rmse=0.0, max_iter=iters,
                                  want_log=True)
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks", type=int, default=5,
                    help="the timed region is repeated this many times back to back (value = median block), steps iterations each")
    ap.add_argument("--search", default="f32", choices=["f32", "f64"],
                    help="f32: fp32 prefilter search + fp64 confirmation + exact fixup (default); f64: per-lane fp64 search")
    ap.add_argument("--force-exchange", action="store_true",
                    help="with one rank under torch.distributed.run: still run pass -> RCCL all-gather -> solve (rehearsal of N > 1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or (args.force_exchange and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
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
    total_iters = args.warmup + blocks * args.steps
    crit = reg.ICPConvergenceCriteria(0.0, 0.0, total_iters)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # An untimed rehearsal of the first iterations of this registration (code objects, caches, clocks), then the real thing.
    icp.setup(d_src, d_tgt, d_nrm, radius, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, 4), n)
    for _ in range(5):
        icp.step()
    barrier()
    icp.setup(d_src, d_tgt, d_nrm, radius, np.eye(4), est, crit, n)
    setup_ms = engine.sess.setup_ms()
    barrier()
    engine.sess.profile(True)
    # pass 0 (the initial correspondence search, Registration.cpp:154) + warmup iterations: not part of `value`, but timed
    # for the whole-call figure in config (a loop that starts from scratch pays for them)
    t_first = time.perf_counter()
    icp.step()
    for _ in range(args.warmup):
        icp.step()
    barrier()
    head_s = time.perf_counter() - t_first
    head_ms = engine.sess.profile_read_all()
    engine.sess.profile_read2()  # clears the record: what follows is the timed region only
    # the timed region: `blocks` x `steps` iterations, each block bracketed by barrier + synchronize
    block_s = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            icp.step()
        barrier()
        block_s.append(time.perf_counter() - t0)
    if dist.is_initialized():
        t = torch.tensor(block_s, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        block_s = [float(v) for v in t.cpu()]
    dt = float(np.median(block_s))
    pass_ms = engine.sess.profile_read_all()
    (first_ms, second_ms), kern_n = engine.sess.profile_read2()  # f32 path: k_icp_cert, k_icp_search_coop; f64 path: search, accumulate
    search_ms = first_ms
    kern_ms = first_ms + second_ms
    one_launch = args.search == "f32" and first_ms < 0.02 < second_ms  # k_icp_pass: nothing between the first two events
    res = icp.finish(want_corr=False)
    n_corr_global = res.n_corr_
    fitness = res.fitness_

    if rank == 0:
        iters_per_s = args.steps / dt
        corr_per_s = args.steps * n / dt
        # algorithmic bytes of one pass (one k_icp_search_coop launch) on this rank: per source point 24 B read + 24 B
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
                               2 if args.estimation == "p2plane" else 3, total_iters, world),
                "estimation": args.estimation,
                "fitness": fitness,
                "inlier_rmse": res.inlier_rmse_,
                "n_corr": n_corr_global,
                "grid_build_ms": float(setup_ms[0]),
                "source_sort_ms": float(setup_ms[1]),
                "estimate_normals_ms_incl_pcie": normals_ms,
                "estimate_normals_ms_device_resident": normals_dev_ms,
                "blocks": blocks,
                "iters_per_s_min_median_max": [args.steps / max(block_s), iters_per_s, args.steps / min(block_s)],
                "kernel_ms_per_pass_min_max": [float(pass_ms.min()), float(pass_ms.max())] if len(pass_ms) else None,
                # a registration that starts from scratch: its first passes search everything (nothing to certify yet)
                "first_passes_kernel_ms": [round(float(v), 4) for v in head_ms[: 1 + args.warmup]],
                "first_passes_wall_ms": 1e3 * head_s,
                "fresh_call_30_iterations_kernel_ms_estimate": float(sum(head_ms[:2]) + 29 * np.median(pass_ms)) if len(head_ms) >= 2 and len(pass_ms) else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_icp_pass: ONE launch per ICP pass -- every wavefront streams its segment of the source (update in place, "
                           "certified previous matches, MFMA sums: the k_icp_cert half), searches what it could not certify (the "
                           "k_icp_search_coop half) and the last workgroups to arrive add the partial rows and solve the 6x6 system.  "
                           "HIP events on the launch stream"
                           if one_launch else
                           "k_icp_cert (update in place + certified previous matches + MFMA sums) and k_icp_search_coop (exact "
                           "correspondence search of what is left + MFMA sums): the two kernels of an ICP pass (O3D_ONE_LAUNCH=0); "
                           "the second one also ends the pass (its last workgroups add the partial rows and solve the 6x6 system: no third launch).  HIP events on the launch stream")
                          if args.search == "f32" else "k_icp_search + k_icp_accumulate_mfma (all-fp64 path)",
                # (one launch: the first interval of the pass profile is empty, the second is the whole pass)
                "cert_kernel_ms" if args.search == "f32" else "search_kernel_ms": None if one_launch else search_ms,
                ("pass_kernel_ms" if one_launch else "search_kernel_ms") if args.search == "f32" else "second_kernel_ms": second_ms,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None,
                "kernel_ms": kern_ms,
                "launches_timed": kern_n,
                "algorithmic_bytes_per_launch": bytes_launch,
            },
        }
        # HBM traffic per pass from the PMC passes of this same command (scripts/prof_pmc.sh; counters cannot be read from
        # inside the process): 2 x FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads; applied to all reads,
        # so an upper bound) + WRITE_SIZE, summed over the two kernels of a pass.  Only quoted for the workload it was collected on.
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and n == 10_000_000 and args.estimation == "p2plane" and args.search == "f32" and os.path.exists(tpath):
            with open(tpath) as f:
                tr = json.load(f)
            out["roofline"]["traffic"] = (2.0 * tr["fetch_size_kb"] + tr["write_size_kb"]) * 1024.0
            out["roofline"]["traffic_source"] = tr["source"]
        if world == 1 and not args.no_cpu_baseline:
            n_sample = min(args.cpu_sample, n)
            est_type = 2 if args.estimation == "p2plane" else 1
            out["cpu_baseline"], ref = cpu_baseline(scene, normals, radius, n_sample, args.cpu_iters, est_type)
            # the oracle run is not thrown away: the GPU must reproduce it (checker only -- nothing of it is timed or shipped)
            out["config"]["parity_%s" % ("%dM" % (n // 1_000_000) if n >= 1_000_000 else str(n))] = parity_check(
                ctx, reg, d_src, d_tgt, d_nrm, radius, est, args.cpu_iters, n_sample, ref)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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


def cpu_baseline(scene, normals, radius, n_sample, iters):
    """Oracle (FLANN-shaped KD-tree restatement, OpenMP) on a bounded sample of the same workload."""
    import oracle
    oracle.build()
    threads = oracle.max_threads()
    src = scene["source"][:n_sample]
    t0 = time.time()
    res = oracle.registration_icp(src, scene["target"], radius, None, 2, normals, rel_fitness=0.0, rel_rmse=0.0, max_iter=iters)
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
    }


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
    total_iters = args.warmup + args.steps
    icp.setup(d_src, d_tgt, d_nrm, radius, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, total_iters), n)
    setup_ms = engine.sess.setup_ms()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # pass 0 (the initial correspondence search, Registration.cpp:154) + warmup iterations, untimed
    icp.step()
    for _ in range(args.warmup):
        icp.step()
    barrier()
    engine.sess.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        icp.step()
    barrier()
    dt = time.perf_counter() - t0
    (search_ms, second_ms), kern_n = engine.sess.profile_read2()  # second: the accumulation kernel of the f64 path only
    # single-kernel pass: nothing runs between the second and third event (their distance is event overhead)
    kern_ms = search_ms if args.search == "f32" else search_ms + second_ms
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
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
                "workload": "config 3 headline: point-to-plane ICP, %d source x %d target points on the synthetic height "
                            "field (L=%.1f m, spacing %.2f mm), max_correspondence_distance = 2 spacings, "
                            "ICPConvergenceCriteria(0,0,%d), source sharded over %d GPU(s), target grid replicated"
                            % (n, n, scene["L"], 1e3 * s, total_iters, world),
                "estimation": args.estimation,
                "fitness": fitness,
                "inlier_rmse": res.inlier_rmse_,
                "n_corr": n_corr_global,
                "grid_build_ms": float(setup_ms[0]),
                "source_sort_ms": float(setup_ms[1]),
                "estimate_normals_ms_incl_pcie": normals_ms,
                "estimate_normals_ms_device_resident": normals_dev_ms,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_icp_search_coop (transform + exact correspondence search + MFMA sums: the one kernel of an ICP "
                          "pass; k_icp_finalize_solve follows), HIP events on the launch stream"
                          if args.search == "f32" else "k_icp_search + k_icp_accumulate_mfma (all-fp64 path)",
                "search_kernel_ms": search_ms,
                "second_kernel_ms": second_ms,
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
        # HBM traffic per launch from the PMC passes of this same command (scripts/prof_pmc.sh; counters cannot be read
        # from inside the process): 2 x FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads; nearly all reads
        # here are 16-byte loads) + WRITE_SIZE.  Only quoted for the workload it was collected on.
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and n == 10_000_000 and args.estimation == "p2plane" and args.search == "f32" and os.path.exists(tpath):
            with open(tpath) as f:
                tr = json.load(f)
            out["roofline"]["traffic"] = (2.0 * tr["fetch_size_kb"] + tr["write_size_kb"]) * 1024.0
            out["roofline"]["traffic_source"] = tr["source"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, normals, radius, min(args.cpu_sample, n), args.cpu_iters)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

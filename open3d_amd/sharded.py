"""Source-sharded ICP over several GPUs.

One process per GPU.  Source points are split into contiguous index ranges, the target grid is replicated, every
rank runs the fused pass on its shard, and ONE small exchange per iteration -- an all-gather of 32 fp64 partial sums
per rank over RCCL/xGMI (torch.distributed backend "nccl") -- gives every rank the same G x 32 table, which each rank
adds in rank order and solves redundantly.  All ranks therefore hold bit-identical transformations and take the same
convergence decision; no further exchange is needed.  The all-gather (rather than an all-reduce) keeps the result
independent of RCCL's reduction tree.

The compute engine is the HIP library (IcpSession).  Tests inject another engine with the same four methods to run the
orchestration over gloo on CPUs.
"""
import numpy as np

NSUMS = 32


def shard_bounds(n, world_size, rank):
    """Contiguous range [lo, hi) of rank's source points: [g*N/G, (g+1)*N/G)."""
    lo = (n * rank) // world_size
    hi = (n * (rank + 1)) // world_size
    return lo, hi


class HipEngine:
    """IcpSession adapter working on torch CUDA tensors."""

    def __init__(self, ctx=None):
        from .pipelines.registration import IcpSession
        self.sess = IcpSession(ctx)
        self._bound = None

    def bind_stream(self):
        """The pass and solve kernels, and the collective torch.distributed issues between them, must be ordered: put
        the context on torch's current stream (re-bound whenever that stream has changed since the last call)."""
        import torch
        ctx = self.sess.ctx
        with torch.cuda.device(ctx.device):
            h = int(torch.cuda.current_stream().cuda_stream)
        if self._bound != h:
            ctx.set_stream(h)
            self._bound = h

    def setup(self, src, tgt, tgt_normals, max_dist, init, estimation, criteria, n_src_global):
        self.bind_stream()
        self.sess.setup(src, tgt, tgt_normals, max_dist, init, estimation, criteria, n_src_global)

    def pass_into(self, local_sums):
        self.bind_stream()
        self.sess.enqueue_pass_to(local_sums.data_ptr())

    def solve_from(self, all_sums, n_ranks):
        self.sess.enqueue_solve(all_sums.data_ptr(), n_ranks)

    def step_unsharded(self):
        self.sess.enqueue_step()

    def poll(self):
        return self.sess.poll()

    def finish(self, want_corr=True):
        return self.sess.finish(want_corr=want_corr)


class ShardedICP:
    """RegistrationICP (reference Registration.cpp:129-168) with the source cloud sharded over a process group."""

    def __init__(self, engine, group=None, device=None, force_exchange=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")
        self.local = torch.zeros(NSUMS, dtype=torch.float64, device=self.device)
        self.all = torch.zeros(NSUMS * self.world, dtype=torch.float64, device=self.device)
        self.max_iter = 0
        # run the pass -> all-gather -> solve sequence even with one rank (rehearses the N > 1 path on one GPU)
        self.force_exchange = bool(force_exchange) and dist.is_initialized()

    def setup(self, src_shard, tgt, tgt_normals, max_dist, init, estimation, criteria, n_src_global):
        self.max_iter = criteria.max_iteration_
        self.engine.setup(src_shard, tgt, tgt_normals, max_dist, init, estimation, criteria, n_src_global)

    def _all_gather(self):
        """The iteration's one exchange.  RCCL ("nccl") gathers device tensors on the stream; gloo cannot gather device tensors,
        so under gloo (CPU tests, and the rehearsal of several ranks on ONE GPU, bench.py --dist-backend gloo) a device tensor
        goes through the host -- correct, synchronous, not a measurement."""
        if self.local.is_cuda and self.dist.get_backend(self.group) == "gloo":
            loc = self.local.cpu()
            allc = self.torch.empty(NSUMS * self.world, dtype=self.torch.float64)
            self.dist.all_gather_into_tensor(allc, loc, group=self.group)
            self.all.copy_(allc)
        else:
            self.dist.all_gather_into_tensor(self.all, self.local, group=self.group)

    def step(self):
        """One LOOP-A pass + estimate on every rank (enqueue only; no host synchronisation)."""
        if self.world == 1 and not self.force_exchange and hasattr(self.engine, "step_unsharded"):
            self.engine.step_unsharded()
            return
        self.engine.pass_into(self.local)
        if self.world > 1 or self.force_exchange:
            self._all_gather()
            self.engine.solve_from(self.all, self.world)
        else:
            self.engine.solve_from(self.local, 1)

    def run(self, check_every=6):
        self.step()
        it = 0
        while it < self.max_iter:
            upto = min(it + check_every, self.max_iter)
            while it < upto:
                self.step()
                it += 1
            if it < self.max_iter and self.engine.poll()[0]:
                break
        return self

    def finish(self, want_corr=True):
        return self.engine.finish(want_corr=want_corr)


def pair_indices(n_pairs, world_size, rank):
    """Pairs of a batch dealt round-robin: rank g registers pairs g, g + G, g + 2G, ... (BASELINE config 5)."""
    return list(range(rank, n_pairs, world_size))


class ShardedBatch:
    """Batches of independent fragment pairs over a process group: no data-path collective.

    Every rank registers its own pairs on its own GPU (RegistrationICPBatch by default); only the results -- 19 doubles
    per pair: transformation, fitness, inlier_rmse, iterations -- are gathered at the end so that every rank can return
    the whole list in pair order.  `run_local(indices) -> [RegistrationResult]` is injectable for the CPU tests."""

    ROW = 19

    def __init__(self, run_local, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.run_local = run_local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")

    def run(self, n_pairs):
        from .pipelines.registration import RegistrationResult
        mine = pair_indices(n_pairs, self.world, self.rank)
        local = self.run_local(mine)
        assert len(local) == len(mine)
        per = (n_pairs + self.world - 1) // self.world  # rows per rank, padded
        buf = np.zeros((max(per, 1), self.ROW))
        for k, r in enumerate(local):
            buf[k, :16] = np.asarray(r.transformation_, dtype=np.float64).reshape(16)
            buf[k, 16:] = (r.fitness_, r.inlier_rmse_, getattr(r, "iterations_", 0))
        t = self.torch.from_numpy(buf).to(self.device)
        if self.world > 1:
            allt = self.torch.zeros((self.world * t.shape[0], self.ROW), dtype=t.dtype, device=self.device)
            self.dist.all_gather_into_tensor(allt, t, group=self.group)  # results only: rank-major concatenation
            table = allt.cpu().numpy().reshape(self.world, t.shape[0], self.ROW)
        else:
            table = t.cpu().numpy()[None]
        out = []
        for p in range(n_pairs):
            row = table[p % self.world, p // self.world]
            r = RegistrationResult(row[:16].reshape(4, 4).copy())
            r.fitness_, r.inlier_rmse_, r.iterations_ = float(row[16]), float(row[17]), int(row[18])
            out.append(r)
        return out


def hip_batch_runner(sources, targets, max_correspondence_distance, inits=None, estimation=None, criteria=None, n_workers=0, ctx=None):
    """run_local for ShardedBatch on the HIP library: RegistrationICPBatch over this rank's pairs."""
    from .pipelines.registration import RegistrationICPBatch

    def run(indices):
        if not indices:
            return []
        return RegistrationICPBatch([sources[i] for i in indices], [targets[i] for i in indices], max_correspondence_distance,
                                    None if inits is None else [inits[i] for i in indices], estimation, criteria,
                                    n_workers=n_workers, want_corr=False, ctx=ctx)
    return run


class HipNormalsEngine:
    """o3d_estimate_normals_shard / _assemble on host (numpy) buffers."""

    def __init__(self, ctx=None):
        from . import _lib
        self._lib = _lib
        self.ctx = ctx or _lib.default_context()

    def shard(self, points, normals, search_param, shard, n_shards, fast_normal_computation=True):
        """-> (part rows x 3 float64, ng)."""
        import ctypes as C
        lib, c = self._lib, self.ctx
        c.check(c._lib.o3d_set_fast_normal_computation(c.handle, int(bool(fast_normal_computation))))
        n = len(points)
        rows = (n + n_shards - 1) // n_shards  # upper bound of ceil(ng / n_shards): ng <= n
        part = np.zeros((max(rows, 1), 3))
        ng = C.c_int64()
        c.check(c._lib.o3d_estimate_normals_shard(c.handle, lib.ptr(points), n, search_param.search_type, search_param.knn_,
                                                  search_param.radius_, lib.MEM_HOST, lib.ptr(normals), shard, n_shards,
                                                  lib.ptr(part), C.byref(ng)))
        rows_ng = (ng.value + n_shards - 1) // n_shards
        return part[:rows_ng], ng.value

    def assemble(self, points, all_parts, n_shards):
        lib, c = self._lib, self.ctx
        out = np.zeros((len(points), 3))
        parts = np.ascontiguousarray(all_parts, dtype=np.float64)
        c.check(c._lib.o3d_estimate_normals_assemble(c.handle, lib.ptr(points), len(points), lib.ptr(parts), n_shards, lib.MEM_HOST,
                                                     lib.ptr(out)))
        return out


class ShardedNormals:
    """PointCloud::EstimateNormals (EstimateNormals.cpp:320-351) with the per-point loop split over a process group: every
    rank builds the grid over the whole cloud, computes the normals of its contiguous range of cell-ordered points, the
    parts are all-gathered in rank order (one collective of 24 B per point) and scattered to the caller's order on every
    rank.  Result bit-identical to the single-GPU call."""

    def __init__(self, engine=None, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.engine = engine or HipNormalsEngine()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")

    def run(self, cloud, search_param, fast_normal_computation=True):
        pts = np.ascontiguousarray(cloud.points_, dtype=np.float64)
        ref = np.ascontiguousarray(cloud.normals_, dtype=np.float64) if cloud.HasNormals() else None
        if fast_normal_computation:  # (engines injected by the CPU tests know only the default)
            part, ng = self.engine.shard(pts, ref, search_param, self.rank, self.world)
        else:
            part, ng = self.engine.shard(pts, ref, search_param, self.rank, self.world, fast_normal_computation=False)
        rows = part.shape[0]  # ceil(ng / world): the same on every rank
        t = self.torch.from_numpy(np.ascontiguousarray(part)).to(self.device)
        if self.world > 1:
            allt = self.torch.zeros((self.world * rows, 3), dtype=t.dtype, device=self.device)
            if rows > 0:
                self.dist.all_gather_into_tensor(allt, t, group=self.group)
            parts = allt.cpu().numpy()
        else:
            parts = t.cpu().numpy()
        cloud.normals_ = self.engine.assemble(pts, parts, self.world)
        return cloud


class ShardedMultiScaleICP:
    """Coarse-to-fine registration (BASELINE configs 3 and 4: VoxelDownSample -> EstimateNormals -> point-to-plane ICP per
    scale) over a process group, one process per GPU.  Per scale
      * VoxelDownSample of both clouds runs replicated on every rank (a global sort by voxel key; its output
        order -- ascending voxel index -- is deterministic, so every rank holds the same arrays),
      * EstimateNormals of the target is split over the ranks (ShardedNormals: one all-gather of the parts),
      * RegistrationICP is split by source shard (ShardedICP: one all-gather of 32 sums per iteration), starting from the
        previous scale's transformation; the shards' correspondences are gathered so that every rank returns the whole array.
    The same loop as pipelines.registration.MultiScaleICP, and the same results (correspondences bit for bit).
    `downsample(cloud, voxel) -> cloud`, `icp_engine()` and `normals_engine` are injectable for the CPU tests."""

    def __init__(self, icp_engine=None, normals_engine=None, downsample=None, group=None, device=None, ctx=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")
        self.ctx = ctx
        self.icp_engine = icp_engine or (lambda: HipEngine(ctx))
        self.normals_engine = normals_engine
        self.downsample = downsample or (lambda cloud, v: cloud.VoxelDownSample(v, ctx=ctx))

    def _gather_corr(self, corr_shard, n_total):
        """Shards of the correspondence array in rank order -> the whole array on every rank."""
        if self.world == 1:
            return np.asarray(corr_shard, dtype=np.int32)
        per = (n_total + self.world - 1) // self.world + 1     # every shard is at most this long ([g*N/G, (g+1)*N/G))
        buf = np.full(per, -2, np.int32)
        buf[:len(corr_shard)] = corr_shard
        t = self.torch.from_numpy(buf).to(self.device)
        allt = self.torch.empty(per * self.world, dtype=t.dtype, device=self.device)
        self.dist.all_gather_into_tensor(allt, t, group=self.group)
        table = allt.cpu().numpy().reshape(self.world, per)
        out = np.empty(n_total, np.int32)
        for g in range(self.world):
            lo, hi = shard_bounds(n_total, self.world, g)
            out[lo:hi] = table[g, :hi - lo]
        return out

    def run(self, source, target, voxel_sizes, max_correspondence_distances=None, init=None, estimation=None, criteria=None,
            normal_radius_factor=2.0, normal_max_nn=30, want_corr=True):
        from .geometry import KDTreeSearchParamHybrid, PointCloud
        from .pipelines import registration as reg
        from . import _lib
        estimation = estimation or reg.TransformationEstimationPointToPlane()
        p2l = estimation.GetTransformationEstimationType() == _lib.EST_POINT_TO_PLANE
        T = np.eye(4) if init is None else np.asarray(init, dtype=np.float64)
        results = []
        for k, v in enumerate(voxel_sizes):
            src_v = self.downsample(source, v) if v > 0 else source
            tgt_v = self.downsample(target, v) if v > 0 else target
            if p2l and not (v <= 0 and tgt_v.HasNormals()):
                tgt_v = PointCloud(tgt_v.points_)
                ShardedNormals(self.normals_engine or HipNormalsEngine(self.ctx), self.group, self.device).run(
                    tgt_v, KDTreeSearchParamHybrid(normal_radius_factor * v, normal_max_nn))
            r = max_correspondence_distances[k] if max_correspondence_distances is not None else 2.0 * v
            crit = criteria[k] if isinstance(criteria, (list, tuple)) else (criteria or reg.ICPConvergenceCriteria())
            src_pts = np.ascontiguousarray(src_v.points_, dtype=np.float64)
            n_src = len(src_pts)
            lo, hi = shard_bounds(n_src, self.world, self.rank)
            icp = ShardedICP(self.icp_engine(), self.group, self.device)
            icp.setup(src_pts[lo:hi], np.ascontiguousarray(tgt_v.points_, dtype=np.float64),
                      np.ascontiguousarray(tgt_v.normals_, dtype=np.float64) if p2l else None, r, T, estimation, crit, n_src)
            icp.run()
            res = icp.finish(want_corr=want_corr)
            if want_corr:
                res.corr_ = self._gather_corr(res.corr_, n_src)
            res.n_source_, res.n_target_ = n_src, len(tgt_v.points_)
            results.append(res)
            T = res.transformation_
        return results[-1], results

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

    def setup(self, src, tgt, tgt_normals, max_dist, init, estimation, criteria, n_src_global):
        self.sess.setup(src, tgt, tgt_normals, max_dist, init, estimation, criteria, n_src_global)

    def pass_into(self, local_sums):
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

    def step(self):
        """One LOOP-A pass + estimate on every rank (enqueue only; no host synchronisation)."""
        if self.world == 1 and not self.force_exchange and hasattr(self.engine, "step_unsharded"):
            self.engine.step_unsharded()
            return
        self.engine.pass_into(self.local)
        if self.world > 1 or self.force_exchange:
            self.dist.all_gather_into_tensor(self.all, self.local, group=self.group)
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

"""CPU suite: the source-sharded multi-rank ICP driver (open3d_amd/sharded.py) over gloo, world_size 2.

The HIP engine cannot run here (no GPU), so the test injects an engine with the same four methods built on the CPU
oracle; what is under test is the product's orchestration: contiguous source shards, one all-gather of 32 fp64 sums
per iteration, rank-ordered summation, identical decisions on every rank, shard-order concatenation of the result.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Stand-in for HipEngine with the semantics of k_icp_search/accumulate + k_icp_solve (device side of the loop)."""

    def setup(self, src, tgt, tgt_normals, max_dist, init, estimation, criteria, n_src_global):
        import oracle
        self.orc = oracle
        self.src = np.array(src, dtype=np.float64)
        self.tgt, self.nrm = np.asarray(tgt), tgt_normals
        self.r = max_dist
        self.T = np.array(init, dtype=np.float64)
        self.update = np.array(init, dtype=np.float64)
        self.apply = not oracle.is_identity(init)
        self.est = estimation.GetTransformationEstimationType()
        self.scaling = getattr(estimation, "with_scaling_", False)
        self.crit = criteria
        self.n_global = n_src_global
        self.passes = self.iters = 0
        self.done = False
        self.fit = self.rmse = 0.0
        self.corr = np.full(len(self.src), -1, np.int32)

    def pass_into(self, local):
        if self.done:
            return
        if self.apply:
            self.src = self.orc.transform_points(self.update, self.src)
        corr, nc, sd2, _, _ = self.orc.correspondences(self.src, self.tgt, self.r)
        self.corr = corr
        q = np.zeros(32)
        m = corr >= 0
        if nc:
            if self.est == 2:
                _, JTJ, JTr, r2 = self.orc.p2plane_compute(self.src, self.tgt, self.nrm, corr)
                q[:21] = JTJ[np.triu_indices(6)]
                q[21:27] = JTr
                q[27] = r2
            else:
                s, t = self.src[m], self.tgt[corr[m]]
                q[0:3], q[3:6] = s.sum(0), t.sum(0)
                q[6:15] = (t.T @ s).reshape(9)
                q[15] = (s * s).sum()
        q[28], q[29] = sd2, nc
        local.copy_(__import__("torch").from_numpy(q))

    def solve_from(self, all_sums, n_ranks):
        if self.done:
            return
        a = all_sums.numpy().reshape(n_ranks, 32)
        q = a[0].copy()
        for r in range(1, n_ranks):
            q = q + a[r]
        nc = q[29]
        fit = nc / self.n_global if nc > 0 else 0.0
        rmse = np.sqrt(q[28] / nc) if nc > 0 else 0.0
        pf, pr = self.fit, self.rmse
        self.fit, self.rmse = fit, rmse
        p = self.passes
        self.passes += 1
        if p >= 1:
            self.iters = p
            if abs(pf - fit) < self.crit.relative_fitness_ and abs(pr - rmse) < self.crit.relative_rmse_:
                self.done = True
                return
        if p >= self.crit.max_iteration_:
            self.done = True
            return
        upd = np.eye(4)
        if nc > 0:
            if self.est == 2:
                A = np.zeros((6, 6))
                A[np.triu_indices(6)] = q[:21]
                A = A + A.T - np.diag(np.diag(A))
                upd = self.orc.vec6_to_mat4(self.orc.ldlt_solve(A, -q[21:27]))
            else:
                ms, md = q[0:3] / nc, q[3:6] / nc
                sig = q[6:15].reshape(3, 3) / nc - np.outer(md, ms)
                U, sv, Vt = np.linalg.svd(sig)
                S = np.ones(3)
                if np.linalg.det(U) * np.linalg.det(Vt) < 0:
                    S[2] = -1
                R = U @ np.diag(S) @ Vt
                upd[:3, :3] = R
                upd[:3, 3] = md - R @ ms
        self.update = upd
        self.T = upd @ self.T
        self.apply = True

    def poll(self):
        return self.done, self.passes

    def finish(self, want_corr=True):
        from open3d_amd.pipelines.registration import RegistrationResult
        res = RegistrationResult(self.T)
        res.corr_ = self.corr
        res.fitness_, res.inlier_rmse_, res.iterations_ = self.fit, self.rmse, self.iters
        return res


def _worker(rank, world, port, est_name, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from conftest import make_scene
    from open3d_amd.pipelines import registration as reg
    from open3d_amd.sharded import ShardedICP, shard_bounds
    import oracle
    oracle.set_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = make_scene(5001, 6000, cfg_id=31)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    lo, hi = shard_bounds(len(sc["source"]), world, rank)
    est = reg.TransformationEstimationPointToPlane() if est_name == "p2plane" else reg.TransformationEstimationPointToPoint()
    r = 2 * s if est_name == "p2plane" else 3 * s
    icp = ShardedICP(OracleEngine(), None, torch.device("cpu"))
    icp.setup(sc["source"][lo:hi], sc["target"], nrm, r, np.eye(4), est, reg.ICPConvergenceCriteria(), len(sc["source"]))
    icp.run(check_every=3)
    res = icp.finish()
    np.savez(os.path.join(tmpdir, "rank%d.npz" % rank), T=res.transformation_, corr=res.corr_, lo=lo, hi=hi,
             scalars=np.array([res.fitness_, res.inlier_rmse_, res.iterations_]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("est_name", ["p2plane", "p2p"])
def test_sharded_icp_two_ranks_matches_single_process(oracle, tmp_path, est_name):
    import torch.multiprocessing as mp
    from conftest import make_scene
    world = 2
    port = 29500 + (os.getpid() % 2000) + (7 if est_name == "p2p" else 0)
    mp.spawn(_worker, args=(world, port, est_name, str(tmp_path)), nprocs=world, join=True)
    sc = make_scene(5001, 6000, cfg_id=31)
    s = sc["spacing"]
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    ref = oracle.registration_icp(sc["source"], sc["target"], 2 * s if est_name == "p2plane" else 3 * s, None,
                                  2 if est_name == "p2plane" else 1, nrm)
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    # every rank holds the same transformation and scalars, bit for bit
    assert np.array_equal(outs[0]["T"], outs[1]["T"]) and np.array_equal(outs[0]["scalars"], outs[1]["scalars"])
    assert (outs[0]["lo"], outs[0]["hi"], outs[1]["lo"], outs[1]["hi"]) == (0, 2500, 2500, 5001)
    corr = np.concatenate([o["corr"] for o in outs])  # rank order = ascending source index
    assert np.array_equal(corr, ref.corr)
    fit, rmse, iters = outs[0]["scalars"]
    assert int(iters) == ref.iterations
    assert abs(fit - ref.fitness) <= 1e-12 and abs(rmse - ref.inlier_rmse) <= 1e-9 * ref.inlier_rmse
    assert np.max(np.abs(outs[0]["T"] - ref.transformation)) < 1e-8


def test_shard_bounds_cover_everything():
    from open3d_amd.sharded import shard_bounds
    for n in (0, 1, 7, 10_000_000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def _batch_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from conftest import make_scene
    from open3d_amd.pipelines.registration import RegistrationResult
    from open3d_amd.sharded import ShardedBatch, pair_indices
    import oracle
    oracle.set_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ran = []

    def run_local(indices):  # stand-in for hip_batch_runner: the oracle's RegistrationICP on this rank's pairs
        out = []
        for p in indices:
            sc = make_scene(600, 700, cfg_id=300 + p)
            o = oracle.registration_icp(sc["source"], sc["target"], 3 * sc["spacing"], None, 1, None)
            r = RegistrationResult(o.transformation)
            r.fitness_, r.inlier_rmse_, r.iterations_ = o.fitness, o.inlier_rmse, o.iterations
            out.append(r)
            ran.append(p)
        return out

    n_pairs = 5  # odd: the ranks hold different numbers of pairs
    res = ShardedBatch(run_local, None, torch.device("cpu")).run(n_pairs)
    assert ran == pair_indices(n_pairs, world, rank)
    np.savez(os.path.join(tmpdir, "batch%d.npz" % rank), T=np.stack([r.transformation_ for r in res]),
             s=np.array([[r.fitness_, r.inlier_rmse_, r.iterations_] for r in res]), ran=np.array(ran))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_pairs_dealt_over_two_ranks(oracle, tmp_path):
    """BASELINE config 5: independent pairs spread over ranks, no data-path collective; every rank returns the whole list
    in pair order, identical to registering the pairs one by one."""
    import torch.multiprocessing as mp
    from conftest import make_scene
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_batch_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "batch%d.npz" % r)) for r in range(world)]
    assert np.array_equal(outs[0]["T"], outs[1]["T"]) and np.array_equal(outs[0]["s"], outs[1]["s"])
    assert sorted(list(outs[0]["ran"]) + list(outs[1]["ran"])) == [0, 1, 2, 3, 4]
    assert list(outs[0]["ran"]) == [0, 2, 4] and list(outs[1]["ran"]) == [1, 3]
    for p in range(5):
        sc = make_scene(600, 700, cfg_id=300 + p)
        o = oracle.registration_icp(sc["source"], sc["target"], 3 * sc["spacing"], None, 1, None)
        assert np.array_equal(outs[0]["T"][p], o.transformation)
        assert outs[0]["s"][p][0] == o.fitness and outs[0]["s"][p][1] == o.inlier_rmse and int(outs[0]["s"][p][2]) == o.iterations
    from open3d_amd.sharded import pair_indices
    for n in (0, 1, 5, 4096):
        for w in (1, 2, 8):
            got = sorted(sum((pair_indices(n, w, r) for r in range(w)), []))
            assert got == list(range(n))


class OracleNormalsEngine:
    """Stand-in for HipNormalsEngine: same two methods, the oracle's EstimateNormals, 'cell order' = finite points in
    index order."""

    def shard(self, points, normals, sp, shard, n_shards):
        import oracle
        full = oracle.estimate_normals(points, sp.search_type, sp.knn_, sp.radius_, normals)
        fin = np.flatnonzero(np.isfinite(points).all(1))
        ng = len(fin)
        rows = (ng + n_shards - 1) // n_shards
        part = np.zeros((rows, 3))
        mine = fin[shard * rows:(shard + 1) * rows]
        part[:len(mine)] = full[mine]
        return part, ng

    def assemble(self, points, all_parts, n_shards):
        fin = np.flatnonzero(np.isfinite(points).all(1))
        out = np.tile([0.0, 0.0, 1.0], (len(points), 1))
        out[fin] = all_parts[:len(fin)]
        return out


def _normals_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from conftest import make_scene
    from open3d_amd import geometry
    from open3d_amd.sharded import ShardedNormals
    import oracle
    oracle.set_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = make_scene(10, 3001, cfg_id=71)
    pts = sc["target"].copy()
    pts[5] = np.nan
    pc = geometry.PointCloud(pts)
    ShardedNormals(OracleNormalsEngine(), None, torch.device("cpu")).run(pc, geometry.KDTreeSearchParamHybrid(4 * sc["spacing"], 30))
    np.save(os.path.join(tmpdir, "nrm%d.npy" % rank), pc.normals_)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_normals_two_ranks(oracle, tmp_path):
    """EstimateNormals split over two ranks (gloo): one all-gather of the parts, every rank ends with the single-process
    normals, bit for bit (odd point count: the ranks' parts differ in length, the last one is padded)."""
    import torch.multiprocessing as mp
    from conftest import make_scene
    world = 2
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_normals_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sc = make_scene(10, 3001, cfg_id=71)
    pts = sc["target"].copy()
    pts[5] = np.nan
    want = oracle.estimate_normals(pts, oracle.HYBRID, 30, 4 * sc["spacing"])
    a, b = (np.load(os.path.join(str(tmp_path), "nrm%d.npy" % r)) for r in range(world))
    assert np.array_equal(a, b) and np.array_equal(a, want)


def _multiscale_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from conftest import make_scene
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    from open3d_amd.sharded import ShardedMultiScaleICP
    import oracle
    oracle.set_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = make_scene(9000, 9001, cfg_id=91, theta_deg=1.0)
    s = sc["spacing"]

    def downsample(cloud, v):  # stand-in for PointCloud.VoxelDownSample on the GPU: the oracle's, same output order
        return geometry.PointCloud(oracle.voxel_down_sample(cloud.points_, v)[0])

    ms = ShardedMultiScaleICP(OracleEngine, OracleNormalsEngine(), downsample, None, torch.device("cpu"))
    final, per_scale = ms.run(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), [4 * s, 2 * s, 0.0],
                              max_correspondence_distances=[8 * s, 4 * s, 2 * s], estimation=reg.TransformationEstimationPointToPlane(),
                              criteria=reg.ICPConvergenceCriteria(), normal_radius_factor=2.0)
    np.savez(os.path.join(tmpdir, "ms%d.npz" % rank), T=np.stack([r.transformation_ for r in per_scale]),
             n=np.array([[r.n_source_, r.n_target_, r.iterations_] for r in per_scale]), corr=final.corr_,
             corr1=per_scale[1].corr_)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_multiscale_two_ranks_matches_single_process(oracle, tmp_path):
    """BASELINE configs 3 / 4 as a process-group pipeline: VoxelDownSample replicated, EstimateNormals and RegistrationICP
    sharded, three scales (the last on the raw clouds).  Both ranks must end with the same per-scale transformations,
    bit for bit, and with the correspondences of the single-process chain of the same calls."""
    import torch.multiprocessing as mp
    from conftest import make_scene
    world = 2
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_multiscale_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (np.load(os.path.join(str(tmp_path), "ms%d.npz" % r)) for r in range(world))
    assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["n"], b["n"]) and np.array_equal(a["corr"], b["corr"])
    # the same chain in one process, straight on the oracle
    sc = make_scene(9000, 9001, cfg_id=91, theta_deg=1.0)
    s = sc["spacing"]
    T = np.eye(4)
    for k, (v, r) in enumerate([(4 * s, 8 * s), (2 * s, 4 * s), (0.0, 2 * s)]):
        src = oracle.voxel_down_sample(sc["source"], v)[0] if v > 0 else sc["source"]
        tgt = oracle.voxel_down_sample(sc["target"], v)[0] if v > 0 else sc["target"]
        nrm = oracle.estimate_normals(tgt, oracle.HYBRID, 30, 2.0 * v)
        ref = oracle.registration_icp(src, tgt, r, T, 2, nrm)
        assert (a["n"][k] == [len(src), len(tgt), ref.iterations]).all(), k
        assert np.max(np.abs(a["T"][k] - ref.transformation)) < 1e-8, k
        if k == 1:
            assert np.array_equal(a["corr1"], ref.corr)
        T = ref.transformation
    assert np.array_equal(a["corr"], ref.corr)
    assert np.max(np.abs(a["T"][-1] - sc["T_true"])) < 5e-3   # and it is the motion the scene was made with (noisy clouds)

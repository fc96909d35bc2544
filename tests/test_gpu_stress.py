"""Stress cases for the pruned cell search: queries hugging cell faces, large coordinate offsets, clustered targets."""
import numpy as np
import pytest
from conftest import make_scene, rel_err

pytestmark = pytest.mark.gpu


def _reg():
    from open3d_amd import geometry, pipelines
    return geometry, pipelines.registration


def _check(ctx, oracle, src, tgt, r, brute=False):
    geometry, reg = _reg()
    got = reg.EvaluateRegistration(geometry.PointCloud(src), geometry.PointCloud(tgt), r, ctx=ctx)
    corr, nc, fit, rmse = oracle.evaluate_registration(src, tgt, r, brute=brute)
    bad = np.nonzero(got.corr_ != corr)[0]
    assert len(bad) == 0, "first mismatches at %s: got %s want %s" % (bad[:5], got.corr_[bad[:5]], corr[bad[:5]])
    assert got.fitness_ == fit


@pytest.mark.parametrize("offset", [0.0, 1.0e4, -3.3e5])
@pytest.mark.parametrize("seed", [0, 1])
def test_volume_cloud_random_radius(ctx, oracle, offset, seed):
    rng = np.random.default_rng(seed)
    tgt = rng.random((40000, 3)) * np.array([2.0, 1.0, 0.5]) + offset
    src = rng.random((30000, 3)) * np.array([2.2, 1.1, 0.6]) - 0.05 + offset
    for r in (0.02, 0.051, 0.3):
        _check(ctx, oracle, src, tgt, r)


@pytest.mark.parametrize("n_tgt", [262144, 262145, 4096, 4097])
def test_sort_variant_boundaries(ctx, oracle, n_tgt):
    """Target sizes either side of a sort tile (4096 keys) and of the 64-tile limit below which a radix pass derives its
    digit bases inside the scatter kernel; volume and sheet clouds (small and multi-million-cell prefix tables)."""
    rng = np.random.default_rng(n_tgt)
    tgt = rng.random((n_tgt, 3)) * np.array([4.0, 2.0, 1.0])
    src = tgt[rng.integers(0, n_tgt, 20000)] + rng.normal(0, 0.004, (20000, 3))
    _check(ctx, oracle, src, tgt, 0.02)                      # ~ 1.3 M cells at 262144 points
    sheet = tgt * np.array([1.0, 1.0, 0.01])
    _check(ctx, oracle, src * np.array([1.0, 1.0, 0.01]), sheet, 0.003)  # 1333 x 667 x few cells: > 4 M table entries


def test_queries_on_cell_faces(ctx, oracle):
    """Targets on a lattice whose pitch equals the cell size; queries exactly on and a hair either side of faces."""
    rng = np.random.default_rng(5)
    r = 0.125
    g = np.arange(0, 20) * r
    tgt = np.stack(np.meshgrid(g, g, g[:6], indexing="ij"), -1).reshape(-1, 3)
    tgt = tgt + rng.normal(0, 1e-9, tgt.shape)
    base = tgt[rng.integers(0, len(tgt), 20000)]
    eps = rng.choice([0.0, 1e-16, -1e-16, 1e-12, -1e-12, 1e-7, -1e-7, 0.4999 * r, -0.4999 * r, 0.5 * r], size=base.shape)
    src = base + eps
    _check(ctx, oracle, src, tgt, r)
    _check(ctx, oracle, src, tgt, r * 0.999999)


def test_clustered_and_duplicate_targets(ctx, oracle):
    rng = np.random.default_rng(9)
    centers = rng.random((50, 3)) * 10
    tgt = (centers[:, None, :] + rng.normal(0, 0.01, (50, 400, 3))).reshape(-1, 3)
    tgt = np.concatenate([tgt, tgt[::7]])  # exact duplicates at higher indices: ties -> lowest index
    src = (centers[:, None, :] + rng.normal(0, 0.03, (50, 300, 3))).reshape(-1, 3)
    _check(ctx, oracle, src, tgt, 0.05, brute=True)
    _check(ctx, oracle, src, tgt, 1.0, brute=True)


def test_coarsened_grid_still_exact(ctx, oracle):
    """Tiny radius in a big sparse volume: the dense table budget forces cells much larger than the radius."""
    rng = np.random.default_rng(2)
    tgt = rng.random((20000, 3)) * 1000.0
    src = tgt[rng.integers(0, len(tgt), 5000)] + rng.normal(0, 0.01, (5000, 3))
    _check(ctx, oracle, src, tgt, 0.05)
    # scattered this thinly, nearly every point would need a brick of its own: the bricked table (test_gpu_sparse.py) is
    # tried, found larger than its budget, and the dense table with coarser cells is built instead
    info = ctx.grid_info()
    assert info["brick_side"] == 0 and info["cell_side"] > 1.0, info


def test_surface_1m_matches_kdtree_oracle(ctx, oracle):
    """1 M x 1 M points of the benchmark surface (config 2 size): bit-exact correspondences."""
    from conftest import make_scene
    sc = make_scene(1_000_000, 1_000_000, cfg_id=2, L=6.32)
    _check(ctx, oracle, sc["source"], sc["target"], 2 * sc["spacing"])


def test_headline_size_10m_every_point(ctx, oracle):
    """BASELINE headline size (10 M x 10 M, point-to-plane, 3 iterations): ALL of it against the oracle -- the KD-tree
    restatement does its four passes over the 10 M points in a few seconds on the box's cores.  (1) every pass's
    correspondence count identical, fitness / rmse / transformation within 1e-6; the final correspondence array bit-exact
    on all 10 M points.  (2) Permuting the source permutes the result and leaves fitness / rmse / transformation unchanged
    (the Hilbert processing order and the fixed-order reductions are internal).  (3) The summary numbers follow from the
    correspondence array.  (4) A fresh EvaluateRegistration at the result equals the loop's last search."""
    from open3d_amd import geometry, synthetic
    from open3d_amd.pipelines import registration as reg
    n = 10_000_000
    sc = synthetic.make_pair(n, n, cfg_id=3)
    s = sc["spacing"]
    r = 2 * s
    src, tgt = sc["source"], sc["target"]
    nrm = synthetic.surface_normals(tgt[:, 0], tgt[:, 1])
    crit = reg.ICPConvergenceCriteria(0.0, 0.0, 3)
    est = reg.TransformationEstimationPointToPlane()
    target = geometry.PointCloud(tgt, normals=nrm)
    sess = reg.IcpSession(ctx).setup(src, tgt, nrm, r, np.eye(4), est, crit)
    sess.run()
    got = sess.finish()
    log = sess.log()
    assert got.iterations_ == 3 and got.fitness_ > 0.99

    # (1) the whole run against the oracle
    ref = oracle.registration_icp(src, tgt, r, None, 2, nrm, rel_fitness=0.0, rel_rmse=0.0, max_iter=3, want_log=True)
    assert ref.iterations == 3 and len(log) == len(ref.log) == 4
    for p in range(4):
        assert log[p, 2] == ref.log[p, 2], "n_corr differs at pass %d" % p
        assert rel_err(log[p, 0], ref.log[p, 0]) < 1e-6 and rel_err(log[p, 1], ref.log[p, 1]) < 1e-6
        assert rel_err(log[p, 4:20], ref.log[p, 4:20]) < 1e-6
    bad = np.nonzero(got.corr_ != ref.corr)[0]
    assert len(bad) == 0, "%d of %d correspondences differ, first at %s" % (len(bad), n, bad[:5])
    assert rel_err(got.transformation_, ref.transformation) < 1e-6
    assert rel_err(got.fitness_, ref.fitness) < 1e-6 and rel_err(got.inlier_rmse_, ref.inlier_rmse) < 1e-6

    # (3) consistency of the summary with the correspondences
    m = got.corr_ >= 0
    T = got.transformation_
    moved = src[m] @ T[:3, :3].T + T[:3, 3]
    d2 = ((moved - tgt[got.corr_[m]]) ** 2).sum(1)
    assert d2.max() < oracle.radius_threshold(r) * (1 + 1e-9)
    assert got.fitness_ == m.sum() / n
    assert abs(got.inlier_rmse_ - np.sqrt(d2.mean())) < 1e-9 * got.inlier_rmse_

    # (4) the loop's last search == a fresh evaluation at its result
    ev = reg.EvaluateRegistration(geometry.PointCloud(src), geometry.PointCloud(tgt), r, T, ctx=ctx)
    assert np.array_equal(ev.corr_, got.corr_)

    # (2) permutation of the source
    rng = np.random.default_rng(0)
    perm = rng.permutation(n)
    got_p = reg.RegistrationICP(geometry.PointCloud(src[perm]), target, r, np.eye(4), est, crit, ctx=ctx)
    assert np.array_equal(got_p.corr_, got.corr_[perm])
    assert got_p.fitness_ == got.fitness_
    assert np.abs(got_p.transformation_ - T).max() < 1e-12 and abs(got_p.inlier_rmse_ - got.inlier_rmse_) < 1e-15


def test_config2_full_size_end_to_end(ctx, oracle):
    """BASELINE config 2 at full size: 1 M x 1 M, EstimateNormals(Hybrid(4 spacings, 30)) on the GPU, point-to-plane ICP with
    the default criteria.  (a) the GPU's normals against the oracle's (1e-9; the closed form's sign is the same on both
    sides); (b) the strict check: the oracle's ICP fed with THE GPU'S normals must give the GPU's correspondences bit for
    bit, the same number of iterations and T / fitness / rmse within 1e-6; (c) the two chains end to end (each with its own
    normals, which differ in the last digits): same iteration count, T within 1e-6, correspondences equal except where a
    1e-9 change of the pose flips a near-tie -- at most a handful of the million."""
    from open3d_amd import geometry
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(1_000_000, 1_000_000, cfg_id=2, L=6.32)
    s = sc["spacing"]
    r = 2 * s
    pc = geometry.PointCloud(sc["target"])
    pc.EstimateNormals(geometry.KDTreeSearchParamHybrid(4.0 * s, 30), ctx=ctx)
    gpu_n = pc.normals_
    ref_n = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    assert np.abs(gpu_n - ref_n).max() < 1e-9                                                   # (a)
    est = reg.TransformationEstimationPointToPlane()
    got = reg.RegistrationICP(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=gpu_n), r, np.eye(4), est,
                              reg.ICPConvergenceCriteria(), ctx=ctx)
    ref = oracle.registration_icp(sc["source"], sc["target"], r, None, 2, gpu_n)                 # (b)
    assert got.iterations_ == ref.iterations
    bad = np.nonzero(got.corr_ != ref.corr)[0]
    assert len(bad) == 0, "%d correspondences differ, first at %s" % (len(bad), bad[:5])
    assert rel_err(got.transformation_, ref.transformation) < 1e-6
    assert rel_err(got.fitness_, ref.fitness) < 1e-6 and rel_err(got.inlier_rmse_, ref.inlier_rmse) < 1e-6
    assert rel_err(got.transformation_, sc["T_true"]) < 1e-3      # and it is the pose the scene was built with
    chain = oracle.registration_icp(sc["source"], sc["target"], r, None, 2, ref_n)               # (c)
    assert chain.iterations == got.iterations_
    assert rel_err(got.transformation_, chain.transformation) < 1e-6
    assert np.count_nonzero(got.corr_ != chain.corr) <= 8


@pytest.mark.gpu
def test_bench_exchange_path_under_torchrun_one_rank():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, RCCL process group), rehearsed with one rank on
    the one GPU of the box: pass -> all-gather of the 32 sums over RCCL -> solve must give the numbers of the unsharded loop."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:   # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    common = ["--gpus", "1", "--points", "200000", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    a = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py")] + common + ["--force-exchange"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert b.returncode == 0, b.stderr[-2000:]
    ja, jb = (json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]) for r in (a, b))
    assert ja["n_gpus"] == 1 and ja["value"] > 0
    for k in ("fitness", "inlier_rmse", "n_corr"):
        assert ja["config"][k] == pytest.approx(jb["config"][k], rel=1e-12), k


def test_seeded_search_walk_on_lattice_with_ties(ctx, oracle):
    """The previous pass's match seeds the next search (k_icp_search_coop).  A walk of caller-given updates over a lattice
    target -- exact ties between 2, 4 and 8 targets, matches that move to another cell, jumps that make the seed useless,
    queries that lose and regain their match -- must give the oracle's correspondences after every single pass."""
    from open3d_amd.pipelines import registration as reg
    g = np.arange(24, dtype=np.float64)
    X, Y, Z = np.meshgrid(g, g, np.arange(3, dtype=np.float64), indexing="ij")
    tgt = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    rng = np.random.default_rng(12)
    base = rng.integers(2, 20, (6000, 3)).astype(np.float64)
    base[:, 2] = rng.integers(0, 3, 6000)
    off = np.array([[0.5, 0, 0], [0.5, 0.5, 0], [0.5, 0.5, 0.5], [0.25, 0, 0], [0, 0, 0], [0.5, 0.25, 0.125]])
    src = base + off[rng.integers(0, len(off), 6000)]
    r = 1.0

    def shift(x, y, z):
        T = np.eye(4)
        T[:3, 3] = (x, y, z)
        return T

    walk = [np.eye(4), shift(1, 0, 0), shift(0, -1, 0), shift(0.25, 0, 0), shift(-0.25, 0.5, 0), shift(0, 0, 0.5),
            shift(7, 3, 0), shift(-7, -3, -0.5), shift(0, 0, 40.0), shift(0, 0, -40.0), shift(0.5, 0.5, 0.5), shift(-1.5, -0.5, -0.5),
            # small moves: matches certified without a search (k_icp_search_coop) until a bound runs out or a tie forms
            shift(1e-9, 0, 0), shift(0, 2e-5, -1e-5), shift(1e-3, -1e-3, 0), shift(0.03, 0, 0.01), shift(0.1, 0.05, 0),
            shift(0.119, -0.049, -0.01), shift(-0.25, 0, 0), shift(0, 0, 0), shift(0, 0, 30.0), shift(1e-6, 0, 0), shift(0, 0, -30.0)]
    for use_tile in (True, False):
        ctx.debug_set(use_tile_search=use_tile, collect_tile_stats=use_tile)
        sess = reg.IcpSession(ctx).setup(src, tgt, None, r, np.eye(4), reg.TransformationEstimationPointToPoint(),
                                         reg.ICPConvergenceCriteria(0, 0, 100))
        cur = src.copy()
        for k, U in enumerate(walk):
            if k > 0:
                sess.set_update(U)
                cur = oracle.transform_points(U, cur)
            if use_tile:
                ctx.tile_stats(reset=True)
            sess.enqueue_pass()
            got = sess.fetch()
            corr, nc, sd2, fit, rmse = oracle.correspondences(cur, tgt, r)
            assert np.array_equal(got.corr_, corr), "pass %d (tile search %s)" % (k, use_tile)
            assert got.fitness_ == fit and rel_err(got.inlier_rmse_, rmse) < 1e-12 if nc else got.fitness_ == 0.0
            if use_tile and k > 0 and not U[:3, 3].any():
                # an update that moves nothing: every entry stands as the pass before left it -- ties included, which no
                # bound can certify -- and nothing is searched
                assert ctx.tile_stats(reset=True)["certified"] == len(src), k
    ctx.debug_set(use_tile_search=True, collect_tile_stats=False)


def test_staged_copies_back_to_back(ctx):
    """xfer.hip: large copies of caller buffers are staged through pinned double buffers by helper threads.  Copies that
    follow each other directly (source then target of one call) must not disturb one another: a copy returns with its last
    DMAs still in flight, and the next one reuses the same pinned buffers."""
    import ctypes as C
    for mb in (13, 33, 100, 250):       # below / above the staging thresholds, odd sizes
        bad = C.c_int64(-1)
        ctx.check(ctx._lib.o3d_debug_copy_check(ctx.handle, (mb << 20) + 24, 3, C.byref(bad)))
        assert bad.value == 0, "%d MiB: %d words differ" % (mb, bad.value)


def test_certified_matches_skip_the_search_and_change_nothing(ctx, oracle):
    """Near convergence k_icp_search_coop certifies most previous matches without a search (distance to the match below the
    carried lower bound on every other target point).  The loop with certificates must give, pass for pass, what the loop
    without them gives -- the same correspondence count, the same sums up to their order of addition -- and the final
    correspondences must equal the oracle's; the counters must show that the certificates were actually used, also for
    the queries WITHOUT a match of a partially overlapping pair."""
    import os
    from open3d_amd.pipelines import registration as reg
    sc = make_scene(120_000, 100_000, cfg_id=21, theta_deg=0.3)
    s = sc["spacing"]
    src = sc["source"].copy()
    src[:30000, 0] += 240.0                   # a quarter of the source lies far outside the target's grid: no match, ever
    src[30000:40000, 2] += 3.0 * s            # ... and some hover 1.5 radii above the surface: inside the grid, no match
    nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s)
    est, crit = reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0, 0, 24)
    logs, finals, stats = {}, {}, {}
    for mode in ("cert", "nocert"):
        os.environ.pop("O3D_NO_CERT", None)
        if mode == "nocert":
            os.environ["O3D_NO_CERT"] = "1"
        try:
            ctx.debug_set(collect_tile_stats=True)
            ctx.tile_stats(reset=True)
            sess = reg.IcpSession(ctx).setup(src, sc["target"], nrm, 2 * s, np.eye(4), est, crit)
            sess.run()
            finals[mode] = sess.finish()
            logs[mode] = sess.log()
            stats[mode] = ctx.tile_stats(reset=True)
        finally:
            os.environ.pop("O3D_NO_CERT", None)
            ctx.debug_set(collect_tile_stats=False)
    a, b = logs["cert"], logs["nocert"]
    assert a.shape == b.shape and a.shape[0] == 25
    # n_corr of every pass is the same number; the fp64 sums are added in another order (two kernels' partial rows)
    assert np.array_equal(a[:, 2], b[:, 2]) and np.abs(a[:, 3] - b[:, 3]).max() <= 1e-12 * a[:, 3].max()
    assert np.abs(a[:, 4:20] - b[:, 4:20]).max() < 1e-12                             # the transformation before every pass
    assert np.array_equal(finals["cert"].corr_, finals["nocert"].corr_)
    ref = oracle.registration_icp(src, sc["target"], 2 * s, None, 2, nrm, rel_fitness=0.0, rel_rmse=0.0, max_iter=24)
    assert np.array_equal(finals["cert"].corr_, ref.corr)
    assert stats["nocert"]["certified"] == 0
    assert stats["cert"]["certified"] > 8 * len(src), stats["cert"]       # most queries in most of the later passes
    assert stats["cert"]["item_trips"] < 0.6 * stats["nocert"]["item_trips"]

"""Stress cases for the pruned cell search: queries hugging cell faces, large coordinate offsets, clustered targets."""
import numpy as np
import pytest

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


def test_surface_1m_matches_kdtree_oracle(ctx, oracle):
    """1 M x 1 M points of the benchmark surface (config 2 size): bit-exact correspondences."""
    from conftest import make_scene
    sc = make_scene(1_000_000, 1_000_000, cfg_id=2, L=6.32)
    _check(ctx, oracle, sc["source"], sc["target"], 2 * sc["spacing"])

"""Several contexts in one process: per-context counters, contexts on a device other than the caller's current one."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cert_policy")]


def _reg():
    from open3d_amd import geometry, pipelines
    return geometry, pipelines.registration


def test_bound_violation_counter_is_per_context(oracle):
    """The safety net of the fp32 prefilter (a winner whose fp64 confirmation fails) is counted in the CONTEXT's device
    memory: a violation injected into context A must fail A's finish -- and only A's -- while B, set up and running on the
    same GPU in between, returns its result; A works again after the failure has been reported once."""
    from open3d_amd import _lib
    geometry, reg = _reg()
    a, b = _lib.Context(0), _lib.Context(0)
    try:
        sc = make_scene(20000, 20000, cfg_id=5)
        r = 2 * sc["spacing"]
        ref = oracle.evaluate_registration(sc["source"], sc["target"], r)[0]
        est, crit = reg.TransformationEstimationPointToPoint(), reg.ICPConvergenceCriteria(0, 0, 2)
        sa = reg.IcpSession(a).setup(sc["source"], sc["target"], None, r, np.eye(4), est, crit)
        sb = reg.IcpSession(b).setup(sc["source"], sc["target"], None, r, np.eye(4), est, crit)
        sa.enqueue_step()
        sb.enqueue_step()
        a.check(a._lib.o3d_debug_inject_bound_violation(a.handle, 3))
        got_b = sb.finish()                      # B finishes first: must not consume (or see) A's violation
        assert np.array_equal(got_b.corr_, ref)
        with pytest.raises(_lib.O3DError) as ei:
            sa.finish()
        assert ei.value.status == 8 and "3 fp32 prefilter" in str(ei.value)
        # reported once, then cleared; and a new setup starts from zero
        got_a = reg.EvaluateRegistration(geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"]), r, ctx=a)
        assert np.array_equal(got_a.corr_, ref)
    finally:
        a.close()
        b.close()


def test_tile_stats_are_per_context():
    from open3d_amd import _lib
    _, reg = _reg()
    a, b = _lib.Context(0), _lib.Context(0)
    try:
        sc = make_scene(30000, 30000, cfg_id=6)
        r = 2 * sc["spacing"]
        est, crit = reg.TransformationEstimationPointToPoint(), reg.ICPConvergenceCriteria(0, 0, 2)
        a.debug_set(collect_tile_stats=True)
        b.debug_set(collect_tile_stats=True)
        sa = reg.IcpSession(a).setup(sc["source"], sc["target"], None, r, np.eye(4), est, crit)
        sb = reg.IcpSession(b).setup(sc["source"][:10000], sc["target"], None, r, np.eye(4), est, crit)
        sa.enqueue_step()
        sb.enqueue_step()
        sb.enqueue_step()
        ta, tb = a.tile_stats(), b.tile_stats()
        assert ta["valid_lanes"] == 30000 and tb["valid_lanes"] == 20000
    finally:
        a.close()
        b.close()


def test_context_on_other_device_than_current(oracle):
    """RegistrationICP and RegistrationICPBatch (worker threads) on Context(1) while the caller's current device is 0: every
    event, copy and launch of the call must land on the context's device.  Needs two GPUs; skipped on a one-GPU box."""
    import torch
    from open3d_amd import _lib
    if _lib.load().o3d_device_count() < 2:
        pytest.skip("one GPU visible")
    geometry, reg = _reg()
    torch.cuda.set_device(0)
    c1 = _lib.Context(1)
    try:
        sc = make_scene(30000, 30000, cfg_id=9)
        r = 2 * sc["spacing"]
        nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * sc["spacing"])
        src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
        est = reg.TransformationEstimationPointToPlane()
        got = reg.RegistrationICP(src, tgt, r, np.eye(4), est, reg.ICPConvergenceCriteria(), ctx=c1)  # default criteria: polls
        ref = oracle.registration_icp(sc["source"], sc["target"], r, None, 2, nrm)
        assert np.array_equal(got.corr_, ref.corr)
        assert torch.cuda.current_device() == 0
        many = reg.RegistrationICPBatch([src] * 6, [tgt] * 6, r, None, est, reg.ICPConvergenceCriteria(), n_workers=3, want_corr=True, ctx=c1)
        for m in many:
            assert np.array_equal(m.corr_, ref.corr)
    finally:
        c1.close()

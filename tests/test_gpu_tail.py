"""The end of a pass inside the search kernel (icp.hip: pass_tail -- the last workgroups to arrive add the partial rows and
run the estimate) against the separate k_icp_finalize_solve launch it replaces (O3D_NO_TAIL=1): which workgroup arrives last
changes from run to run, what is added in which order must not.  Bar: the per-pass log (fitness, rmse, count, sum of squares,
transformation, all 32 sums), the final transformation and the correspondences are IDENTICAL bit for bit, for grids of one
group member each (8 workgroups), partly filled groups and the full 1024-workgroup grid, with and without certificates."""
import os

import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _run(ctx, sc, nrm, est, r, iters, env):
    from open3d_amd.pipelines import registration as reg
    old = {k: os.environ.get(k) for k in ("O3D_NO_TAIL", "O3D_NO_CERT", "O3D_CERT")}
    try:
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nrm, r, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, iters))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    sess.run()
    res = sess.finish()
    return res, sess.log()


@pytest.mark.parametrize("n", [1_000, 70_000, 300_000, 1_500_000])
@pytest.mark.parametrize("est_name", ["p2p", "p2plane"])
def test_pass_tail_equals_separate_finalize_solve(ctx, n, est_name):
    from open3d_amd.pipelines import registration as reg
    from open3d_amd import synthetic
    sc = make_scene(n, n - n // 10, cfg_id=300 + n % 97, theta_deg=0.3)
    s = sc["spacing"]
    nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
    est, r = ((reg.TransformationEstimationPointToPoint(True), 3 * s) if est_name == "p2p"
              else (reg.TransformationEstimationPointToPlane(), 2 * s))
    iters = 60 if n >= 1_000_000 else 12     # (a longer soak on the full 1024-workgroup grid)
    for cert_env in ({"O3D_CERT": "1"}, {"O3D_NO_CERT": "1"}):
        want, want_log = _run(ctx, sc, nrm, est, r, iters, dict(cert_env, O3D_NO_TAIL="1"))
        for rep in range(3):   # who arrives last differs between repetitions
            got, got_log = _run(ctx, sc, nrm, est, r, iters, cert_env)
            assert got.iterations_ == want.iterations_ == iters
            assert np.array_equal(got_log, want_log)
            assert np.array_equal(got.transformation_, want.transformation_)
            assert np.array_equal(got.corr_, want.corr_)
            assert got.fitness_ == want.fitness_ and got.inlier_rmse_ == want.inlier_rmse_


def test_pass_tail_in_the_stepwise_sharded_form(ctx):
    """o3d_icp_pass_to (sums only, the estimate is the caller's): the tail writes the same 32 sums as k_icp_finalize."""
    import torch
    from open3d_amd.pipelines import registration as reg
    from open3d_amd import synthetic
    sc = make_scene(120_000, 100_000, cfg_id=17)
    s = sc["spacing"]
    nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
    out = []
    for env in ({"O3D_NO_TAIL": "1"}, {}):
        old = os.environ.get("O3D_NO_TAIL")
        os.environ.pop("O3D_NO_TAIL", None)
        os.environ.update(env)
        try:
            sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nrm, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                             reg.ICPConvergenceCriteria(0.0, 0.0, 5))
        finally:
            os.environ.pop("O3D_NO_TAIL", None)
            if old is not None:
                os.environ["O3D_NO_TAIL"] = old
        sums = torch.zeros(5, 32, dtype=torch.float64, device="cuda")
        for it in range(5):
            sess.enqueue_pass_to(sums[it].data_ptr())
            sess.enqueue_solve(sums[it].data_ptr(), 1)
        torch.cuda.synchronize()
        out.append((sums.cpu().numpy(), sess.finish().transformation_))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_certificate_size_policy(ctx):
    """Without O3D_CERT / O3D_NO_CERT certificates are used from 200 k source points on, and the result does not depend on
    it: below, the counters show none; above, most of the later passes' queries."""
    from open3d_amd.pipelines import registration as reg
    from open3d_amd import synthetic
    for n, expect in ((150_000, False), (250_000, True)):
        sc = make_scene(n, n, cfg_id=5, theta_deg=0.2)
        s = sc["spacing"]
        nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
        est = reg.TransformationEstimationPointToPlane()
        ctx.debug_set(collect_tile_stats=True)
        try:
            ctx.tile_stats(reset=True)
            got, _ = _run(ctx, sc, nrm, est, 2 * s, 10, {})
            certified = ctx.tile_stats(reset=True)["certified"]
            other, _ = _run(ctx, sc, nrm, est, 2 * s, 10, {"O3D_NO_CERT": "1"} if expect else {"O3D_CERT": "1"})
        finally:
            ctx.debug_set(collect_tile_stats=False)
        assert (certified > 3 * n) if expect else (certified == 0), certified
        assert np.array_equal(got.corr_, other.corr_)
        assert np.abs(got.transformation_ - other.transformation_).max() < 1e-12


@pytest.mark.gpu
def test_repeated_registrations_on_one_context_repeat_their_bits(ctx):
    """What a registration leaves behind in the context (work lists, partial rows, tickets, bounds, and whatever sits in the
    XCDs' caches) must not reach the next one: the same registration run again and again on one context, with registrations
    of other lengths in between and passes enqueued beyond the end of the loop, returns the same bits every time.  (An
    experimental one-launch form of the pass failed exactly this and nothing else.)"""
    import torch
    from open3d_amd import synthetic
    from open3d_amd.pipelines import registration as reg
    n = 30000
    sc = synthetic.make_pair(n, n, cfg_id=0, L=6.0)
    s = sc["spacing"]
    nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
    seen = {}
    for k in (1, 2, 3, 4, 3, 3, 2, 4, 1, 3):
        sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nrm, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(),
                                         reg.ICPConvergenceCriteria(0.0, 0.0, k))
        for _ in range(6):  # (more than the loop has: the surplus passes return at once)
            sess.enqueue_step()
        torch.cuda.synchronize()
        lg = np.array(sess.log())
        assert lg.shape[0] == k + 1
        if k in seen:
            assert np.array_equal(lg, seen[k]), "registration with %d iterations changed on repetition" % k
        seen[k] = lg
    for k in (2, 3, 4):  # a longer registration begins like a shorter one
        assert np.array_equal(seen[k][: k, 20:], seen[k - 1][: k, 20:])


@pytest.mark.gpu
@pytest.mark.parametrize("est_name,n", [("p2plane", 30000), ("p2p", 30000), ("p2plane", 450000)])
def test_one_launch_pass_equals_two_launches(est_name, n):
    """k_icp_pass (certificates and search of a pass in one launch) against k_icp_cert + k_icp_search_coop
    (O3D_ONE_LAUNCH=0): the same partial rows in the same order, so every logged sum and the transformation must agree bit
    for bit -- on a fresh context and on later registrations on the same one, whose first listed pass has a certificate
    half that stands aside and must still overwrite the previous registration's rows for every XCD to see.  The switch is
    read once per process: the two forms run in two child processes."""
    import subprocess
    import sys
    import tempfile
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg
ctx = _lib.Context(0)
n = %d
sc = synthetic.make_pair(n, n, cfg_id=0, L=6.0 if n < 100000 else 20.0)
s = sc["spacing"]
plane = %r == "p2plane"
nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1]) if plane else None
est = reg.TransformationEstimationPointToPlane() if plane else reg.TransformationEstimationPointToPoint()
out = []
for k in (1, 2, 5, 3, 5):
    sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nrm, (2 if plane else 3) * s, np.eye(4), est,
                                     reg.ICPConvergenceCriteria(0.0, 0.0, k))
    for _ in range(7):
        sess.enqueue_step()
    torch.cuda.synchronize()
    out.append(np.array(sess.log()).ravel())
    res = sess.finish()
    out.append(np.asarray(res.corr_, dtype=np.float64))
np.save(sys.argv[1], np.concatenate(out))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n, est_name)
    got = []
    with tempfile.TemporaryDirectory() as tmp:
        for mode in ("1", "0"):
            env = dict(os.environ, O3D_ONE_LAUNCH=mode, O3D_CERT="1")
            path = os.path.join(tmp, "r%s.npy" % mode)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=90)
            got.append(np.load(path))
    assert got[0].shape == got[1].shape and np.array_equal(got[0], got[1])


@pytest.mark.gpu
def test_tail_check_switch_single_and_gang(ctx, oracle):
    """O3D_TAIL_CHECK=1 (ADVICE, round 2): after every pass that ends inside its kernel the partial rows are added once
    more by the separate-launch reduction and compared bit for bit with what pass_tail wrote -- the run-time check of the
    invariant documented at store_partial_row (a partial row written with a plain store would surface here as an
    O3D_ERR_STATE).  Repeated registrations on ONE context for point-to-point, point-to-plane and coloured ICP with
    certificates on (later registrations begin with a certificate half that stands aside and must overwrite the previous
    registration's rows), and the lock-step gang path of the batch entry point; the results must equal the unchecked runs."""
    import test_gpu_companions as comp
    from open3d_amd import geometry, synthetic
    from open3d_amd.pipelines import registration as reg
    keys = ("O3D_TAIL_CHECK", "O3D_CERT", "O3D_NO_CERT")
    old = {k: os.environ.get(k) for k in keys}

    def run_all():
        out = []
        n = 30000
        sc = synthetic.make_pair(n, n, cfg_id=0, L=6.0)
        s = sc["spacing"]
        nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
        src, tgt = geometry.PointCloud(sc["source"]), geometry.PointCloud(sc["target"], normals=nrm)
        for est, r in ((reg.TransformationEstimationPointToPlane(), 2 * s), (reg.TransformationEstimationPointToPoint(False), 3 * s)):
            for k in (3, 1, 4, 3):
                got = reg.RegistrationICP(src, tgt, r, np.eye(4), est, reg.ICPConvergenceCriteria(0.0, 0.0, k), ctx=ctx)
                out.append((got.transformation_.copy(), got.fitness_, got.inlier_rmse_))
        csc = comp._colored_scene(20000, 20400, cfg_id=77)
        cn = oracle.estimate_normals(csc["target"], oracle.HYBRID, 30, 4 * csc["spacing"])
        cs = geometry.PointCloud(csc["source"], colors=csc["source_colors"])
        ct = geometry.PointCloud(csc["target"], normals=cn, colors=csc["target_colors"])
        for k in (4, 2, 4):
            got = reg.RegistrationColoredICP(cs, ct, 2.5 * csc["spacing"], np.eye(4), reg.ICPConvergenceCriteria(0.0, 0.0, k), ctx=ctx)
            out.append((got.transformation_.copy(), got.fitness_, got.inlier_rmse_))
        many = reg.RegistrationICPBatch([src] * 5, [tgt] * 5, 2 * s, None, reg.TransformationEstimationPointToPlane(),
                                        reg.ICPConvergenceCriteria(0.0, 0.0, 6), n_workers=1, ctx=ctx)
        out += [(m.transformation_.copy(), m.fitness_, m.inlier_rmse_) for m in many]
        return out

    try:
        os.environ.pop("O3D_NO_CERT", None)
        os.environ["O3D_CERT"] = "1"
        os.environ.pop("O3D_TAIL_CHECK", None)
        plain = run_all()
        os.environ["O3D_TAIL_CHECK"] = "1"
        checked = run_all()  # (raises O3DError with the first differing sum if the tail and the separate launch disagree)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert len(plain) == len(checked)
    for (Ta, fa, ra), (Tb, fb, rb) in zip(plain, checked):
        assert np.array_equal(Ta, Tb) and fa == fb and ra == rb

"""The small dense algebra of the estimate on the GPU (icp.hip: svd3, ldlt_solve6, vec6_to_mat4, umeyama_from_sums) checked
against numpy.linalg DIRECTLY -- no oracle in the loop: the device code and oracle/o3d_oracle.c are siblings, so a shared
misreading of Eigen would pass every GPU-vs-oracle test.  Reference: TransformationEstimation.cpp:49-60 (Eigen::umeyama),
:74-98 with utility/Eigen.cpp:85,116-130 (6x6 LDLT solve, Rz*Ry*Rx).  Driven through o3d_compute_transformation on crafted
correspondence sets, including the degenerate ones: reflection (det(U) det(V) < 0), rank-2 and rank-1 covariance, an exactly
singular JTJ."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reg():
    from open3d_amd import geometry, pipelines
    return geometry, pipelines.registration


def _rot(axis, deg):
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    th = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def umeyama_numpy(src, dst, with_scaling):
    """Umeyama 1991 / Eigen::umeyama written with numpy.linalg.svd."""
    n = len(src)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    S, D = src - mu_s, dst - mu_d
    sigma = D.T @ S / n
    U, d, Vt = np.linalg.svd(sigma)
    sgn = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        sgn[2] = -1.0
    R = U @ np.diag(sgn) @ Vt
    c = float((d * sgn).sum() / ((S ** 2).sum() / n)) if with_scaling else 1.0
    T = np.eye(4)
    T[:3, :3] = c * R
    T[:3, 3] = mu_d - c * R @ mu_s
    return T, d


def _p2p(ctx, src, dst, with_scaling):
    geometry, reg = _reg()
    n = len(src)
    pairs = np.stack([np.arange(n), np.arange(n)], 1).astype(np.int32)
    est = reg.TransformationEstimationPointToPoint(with_scaling)
    return est.ComputeTransformation(geometry.PointCloud(src), geometry.PointCloud(dst), pairs, ctx=ctx)


@pytest.mark.parametrize("with_scaling", [False, True])
def test_umeyama_generic_against_numpy_svd(ctx, with_scaling):
    rng = np.random.default_rng(1)
    src = rng.normal(0, 1, (5000, 3)) * np.array([3.0, 1.0, 0.3]) + np.array([10.0, -4.0, 2.0])
    R, t, c = _rot([1, 2, 3], 37.0), np.array([0.5, -2.0, 1.0]), (1.7 if with_scaling else 1.0)
    dst = c * src @ R.T + t + rng.normal(0, 0.01, src.shape)
    got = _p2p(ctx, src, dst, with_scaling)
    ref, _ = umeyama_numpy(src, dst, with_scaling)
    assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max()
    assert np.abs(got[:3, :3] - c * R).max() < 1e-3          # and it is the motion the data was made with


@pytest.mark.parametrize("with_scaling", [False, True])
def test_umeyama_reflection_case(ctx, with_scaling):
    """dst is a MIRROR image of src: the unconstrained optimum has det = -1, so det(U) det(V) < 0 and the smallest singular
    direction is flipped (S(2) = -1)."""
    rng = np.random.default_rng(2)
    src = rng.normal(0, 1, (4000, 3)) * np.array([2.0, 1.0, 0.5])
    M = np.diag([1.0, 1.0, -1.0]) @ _rot([0.3, -1, 0.5], 20.0)
    dst = src @ M.T + np.array([1.0, 2.0, 3.0]) + rng.normal(0, 1e-3, src.shape)
    ref, d = umeyama_numpy(src, dst, with_scaling)
    assert d[2] > 1e-3 * d[0]                                  # full rank: the flip is a decision, not noise
    U, _, Vt = np.linalg.svd((dst - dst.mean(0)).T @ (src - src.mean(0)))
    assert np.linalg.det(U) * np.linalg.det(Vt) < 0            # the case this test is about
    got = _p2p(ctx, src, dst, with_scaling)
    assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max()
    Rg = got[:3, :3] / np.cbrt(np.linalg.det(got[:3, :3]))
    assert np.linalg.det(Rg) > 0 and np.abs(Rg @ Rg.T - np.eye(3)).max() < 1e-12


def test_umeyama_rank2_planar(ctx):
    """All points of both sets in a plane: the covariance has rank 2; the rotation is still unique."""
    rng = np.random.default_rng(3)
    src = np.zeros((3000, 3))
    src[:, :2] = rng.normal(0, 1, (3000, 2)) * np.array([2.0, 0.7])
    Rz = _rot([0, 0, 1], 25.0)
    dst = src @ Rz.T + np.array([0.3, -0.2, 0.0])
    dst[:, :2] += rng.normal(0, 1e-3, (3000, 2))               # stays exactly planar
    ref, d = umeyama_numpy(src, dst, False)
    assert d[2] < 1e-12 * d[0]
    got = _p2p(ctx, src, dst, False)
    assert np.abs(got - ref).max() < 1e-9
    assert abs(np.linalg.det(got[:3, :3]) - 1.0) < 1e-12


def test_umeyama_rank1_collinear(ctx):
    """All points on a line: rank-1 covariance, the rotation about the line is free.  What is determined -- where the
    source points land -- must agree with numpy's solution, and the result must be a proper rotation."""
    rng = np.random.default_rng(4)
    u = np.array([1.0, 2.0, -0.5]) / np.linalg.norm([1.0, 2.0, -0.5])
    tpar = rng.normal(0, 2, 2000)
    src = np.array([3.0, 1.0, -2.0]) + tpar[:, None] * u
    v = _rot([0.2, 1, 0.1], 40.0) @ u
    dst = np.array([-1.0, 0.5, 4.0]) + (tpar + rng.normal(0, 1e-3, 2000))[:, None] * v
    ref, d = umeyama_numpy(src, dst, False)
    assert d[1] < 1e-10 * d[0]
    got = _p2p(ctx, src, dst, False)
    R = got[:3, :3]
    assert np.isfinite(got).all()
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-9 and abs(np.linalg.det(R) - 1.0) < 1e-9
    land_got = src @ R.T + got[:3, 3]
    land_ref = src @ ref[:3, :3].T + ref[:3, 3]
    assert np.abs(land_got - land_ref).max() < 1e-8


def _p2plane_system(src, tgt, nrm):
    J = np.concatenate([np.cross(src, nrm), nrm], axis=1)      # TransformationEstimation.cpp:79-86
    r = ((src - tgt) * nrm).sum(1)
    return J.T @ J, J.T @ r


def _vec6_to_T(x):
    cx, sx, cy, sy, cz, sz = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), np.cos(x[2]), np.sin(x[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx                                   # utility/Eigen.cpp:89-97
    T[:3, 3] = x[3:]
    return T


def _p2plane(ctx, src, tgt, nrm):
    geometry, reg = _reg()
    n = len(src)
    pairs = np.stack([np.arange(n), np.arange(n)], 1).astype(np.int32)
    return reg.TransformationEstimationPointToPlane().ComputeTransformation(geometry.PointCloud(src), geometry.PointCloud(tgt, normals=nrm),
                                                                           pairs, ctx=ctx)


def test_point_to_plane_against_numpy_solve(ctx):
    rng = np.random.default_rng(5)
    tgt = rng.normal(0, 1, (6000, 3)) * np.array([2.0, 1.5, 0.8])
    nrm = rng.normal(0, 1, tgt.shape)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    src = (tgt - np.array([0.02, -0.01, 0.03])) @ _rot([1, -1, 2], 1.5) + rng.normal(0, 1e-3, tgt.shape)
    A, b = _p2plane_system(src, tgt, nrm)
    x = np.linalg.solve(A, -b)
    got = _p2plane(ctx, src, tgt, nrm)
    assert np.abs(got - _vec6_to_T(x)).max() < 1e-10
    # badly scaled but regular: far from the origin the rotational columns dwarf the translational ones (pivoting matters)
    off = np.array([800.0, -300.0, 150.0])
    A, b = _p2plane_system(src + off, tgt + off, nrm)
    x = np.linalg.solve(A, -b)
    got = _p2plane(ctx, src + off, tgt + off, nrm)
    assert np.abs(got - _vec6_to_T(x)).max() < 1e-7


def test_point_to_plane_exactly_singular_system(ctx):
    """Every normal is (0, 0, 1) and every point has integer coordinates: JTJ is exactly singular (rotation about z and the
    translations in x, y are unconstrained; numpy.linalg.solve refuses it).  Eigen's LDLT (utility/Eigen.cpp:85, all checks
    off) returns a particular solution with the unconstrained components at zero: the update must be finite, rigid, must
    solve the consistent normal equations, and must leave the unconstrained motions alone."""
    rng = np.random.default_rng(6)
    tgt = rng.integers(-20, 21, (3000, 3)).astype(np.float64)
    nrm = np.tile(np.array([0.0, 0.0, 1.0]), (len(tgt), 1))
    src = tgt.copy()
    src[:, 2] += rng.integers(-1, 2, len(tgt))                 # residuals along z only, integers
    A, b = _p2plane_system(src, tgt, nrm)
    assert np.linalg.matrix_rank(A) == 3
    with pytest.raises(np.linalg.LinAlgError):
        np.linalg.solve(A, -b)
    got = _p2plane(ctx, src, tgt, nrm)
    assert np.isfinite(got).all()
    R = got[:3, :3]
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1.0) < 1e-12
    x_min = np.linalg.lstsq(A, -b, rcond=None)[0]              # the minimum-norm solution of the same consistent system
    # recover x from T = Rz Ry Rx | t
    x_got = np.array([np.arctan2(R[2, 1], R[2, 2]), -np.arcsin(R[2, 0]), np.arctan2(R[1, 0], R[0, 0]), *got[:3, 3]])
    assert np.abs(A @ x_got + b).max() < 1e-8 * max(np.abs(b).max(), 1.0)
    assert np.abs(x_got[[2, 3, 4]]).max() == 0.0               # rotation about z, translation in x and y: untouched
    assert np.abs(x_got[[0, 1, 5]] - x_min[[0, 1, 5]]).max() < 1e-9

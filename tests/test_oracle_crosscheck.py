"""CPU suite: the oracle (oracle/o3d_oracle.c) against independent numpy / scipy computations.

The reference ships no tests or golden vectors, so these cross-checks -- together with the
reference call sites cited in the oracle -- are what pins the restatement ("parity unpinned" by reference fixtures).
"""
import numpy as np
import pytest

from conftest import make_scene


def _np_brute(src, tgt, thr):
    d = src[:, None, :] - tgt[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]  # ((dx*dx)+dy*dy)+dz*dz, no FMA
    j = np.argmin(d2, axis=1)  # first minimum = lowest index on ties
    m = d2[np.arange(len(src)), j]
    return np.where(m < thr, j, -1).astype(np.int32), m


def test_radius_threshold_known_answers(oracle):
    # (double)(float)(r*r), geometry/KDTreeFlann.cpp:148 -- values from SURVEY.md section 8c
    assert oracle.radius_threshold(0.05) == 0.0024999999441206455
    assert oracle.radius_threshold(0.075) == 0.005625000223517418
    assert oracle.radius_threshold(0.05) < 0.05 * 0.05 < 0.075 * 0.075 < oracle.radius_threshold(0.075)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_correspondences_vs_numpy_bruteforce(oracle, seed):
    sc = make_scene(1500, 2000, cfg_id=100 + seed)
    r = 2.5 * sc["spacing"]
    want, d2 = _np_brute(sc["source"], sc["target"], oracle.radius_threshold(r))
    for brute in (False, True):
        corr, nc, sd2, fit, rmse = oracle.correspondences(sc["source"], sc["target"], r, brute=brute)
        assert np.array_equal(corr, want)
        assert nc == (want >= 0).sum() and fit == nc / len(want)
        assert abs(sd2 - d2[want >= 0].sum()) <= 1e-12 * sd2
        assert abs(rmse - np.sqrt(d2[want >= 0].sum() / nc)) <= 1e-12 * rmse


def test_correspondences_ties_lowest_index_and_nan(oracle):
    rng = np.random.default_rng(0)
    tgt = rng.random((300, 3))
    tgt = np.concatenate([tgt, tgt[:40]])  # duplicates: the lower index must win
    src = tgt[:40] + 1e-3
    src[3] = np.nan
    for brute in (False, True):
        corr = oracle.correspondences(src, tgt, 0.1, brute=brute)[0]
        assert corr[3] == -1
        assert np.all(corr[np.arange(40) != 3] < 300)


def test_correspondences_vs_scipy_ckdtree_100k(oracle):
    from scipy.spatial import cKDTree
    sc = make_scene(100_000, 100_000, cfg_id=1)
    r = 3 * sc["spacing"]
    corr, nc, sd2, fit, rmse = oracle.correspondences(sc["source"], sc["target"], r)
    d, j = cKDTree(sc["target"]).query(sc["source"], k=1, distance_upper_bound=np.sqrt(oracle.radius_threshold(r)))
    want = np.where(np.isfinite(d), j, -1)
    assert (corr != want).mean() < 1e-5  # scipy rounds distances differently: only ulp-level ties may differ
    assert abs(fit - np.isfinite(d).mean()) < 1e-5


def test_kdtree_knn_radius_vs_bruteforce(oracle):
    rng = np.random.default_rng(5)
    pts = rng.random((3000, 3))
    for q in pts[:25]:
        for st, knn, rad in [(oracle.KNN, 30, 0.0), (oracle.HYBRID, 12, 0.15), (oracle.RADIUS, 0, 0.12)]:
            ia, da = oracle.neighbors(pts, q, st, knn, rad, brute=False)
            ib, db = oracle.neighbors(pts, q, st, knn, rad, brute=True)
            if st == oracle.RADIUS:  # brute returns its set in (d2, idx) order as well
                d2 = ((q - pts) ** 2)
                d2 = (d2[:, 0] + d2[:, 1]) + d2[:, 2]
                want = np.nonzero(d2 < oracle.radius_threshold(rad))[0]
                assert set(ia.tolist()) == set(want.tolist())
                assert np.all(np.diff(da) >= 0)
            else:
                assert np.array_equal(ia, ib) and np.array_equal(da, db)


def test_transform_points_operation_order(oracle):
    rng = np.random.default_rng(2)
    T = np.eye(4)
    T[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    T[:3, 3] = rng.normal(size=3)
    p = rng.normal(size=(1000, 3)) * 10
    out = oracle.transform_points(T, p)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    rows = [((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3] * 1.0 for r in range(4)]
    want = np.stack([rows[0] / rows[3], rows[1] / rows[3], rows[2] / rows[3]], axis=1)
    assert np.array_equal(out, want)


def test_is_identity_semantics(oracle):
    assert oracle.is_identity(np.eye(4))
    T = np.eye(4); T[0, 3] = 1e-13
    assert oracle.is_identity(T)
    T[0, 3] = 1e-11
    assert not oracle.is_identity(T)
    T = np.eye(4); T[1, 1] = 1 + 1e-11
    assert not oracle.is_identity(T)


def test_ldlt_and_vec6(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        J = rng.normal(size=(50, 6))
        A, b = J.T @ J, rng.normal(size=6)
        x = oracle.ldlt_solve(A, b)
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    # singular direction: Eigen's LDLT solve zeroes the component instead of dividing by 0
    A = np.diag([1.0, 2.0, 0.0, 3.0, 4.0, 5.0])
    x = oracle.ldlt_solve(A, np.ones(6))
    assert np.allclose(x, [1, 0.5, 0, 1 / 3, 0.25, 0.2])
    v = rng.normal(size=6) * 0.3
    T = oracle.vec6_to_mat4(v)
    cx, sx, cy, sy, cz, sz = np.cos(v[0]), np.sin(v[0]), np.cos(v[1]), np.sin(v[1]), np.cos(v[2]), np.sin(v[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    assert np.allclose(T[:3, :3], Rz @ Ry @ Rx, atol=1e-15)  # utility/Eigen.cpp:92-94
    assert np.array_equal(T[:3, 3], v[3:]) and np.array_equal(T[3], [0, 0, 0, 1])


def test_svd3_and_umeyama(oracle):
    rng = np.random.default_rng(4)
    for _ in range(20):
        A = rng.normal(size=(3, 3))
        U, S, V = oracle.svd3(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-13)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-13) and np.allclose(V.T @ V, np.eye(3), atol=1e-13)
        assert np.allclose(S, np.linalg.svd(A, compute_uv=False), atol=1e-13)
    U, S, V = oracle.svd3(np.diag([2.0, 1.0, 0.0]))  # rank deficient
    assert np.allclose(U.T @ U, np.eye(3), atol=1e-13) and np.allclose(S, [2, 1, 0])

    def umeyama(src, dst, scaling):  # Eigen/src/Geometry/Umeyama.h
        ms, md = src.mean(0), dst.mean(0)
        s, d = src - ms, dst - md
        sig = d.T @ s / len(src)
        U, sv, Vt = np.linalg.svd(sig)
        S = np.ones(3)
        if np.linalg.det(U) * np.linalg.det(Vt) < 0:
            S[2] = -1
        R = U @ np.diag(S) @ Vt
        c = (sv * S).sum() / (s ** 2).sum(1).mean() if scaling else 1.0
        T = np.eye(4)
        T[:3, :3] = c * R
        T[:3, 3] = md - c * R @ ms
        return T

    src = rng.normal(size=(400, 3))
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    R *= np.sign(np.linalg.det(R))
    dst = 1.3 * src @ R.T + np.array([0.3, -2.0, 1.0]) + rng.normal(size=src.shape) * 0.01
    corr = np.arange(400, dtype=np.int32)
    corr[::9] = -1
    m = corr >= 0
    for scaling in (False, True):
        T = oracle.p2p_compute(src, dst, corr, scaling)
        assert np.allclose(T, umeyama(src[m], dst[corr[m]], scaling), atol=1e-12)
    assert np.array_equal(oracle.p2p_compute(src, dst, np.full(400, -1, np.int32)), np.eye(4))  # :51-52


def test_point_to_plane_system(oracle):
    rng = np.random.default_rng(6)
    src, tgt = rng.normal(size=(300, 3)), rng.normal(size=(350, 3))
    nrm = rng.normal(size=(350, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    corr = rng.integers(-1, 350, 300).astype(np.int32)
    T, JTJ, JTr, r2 = oracle.p2plane_compute(src, tgt, nrm, corr)
    m = corr >= 0
    vs, vt, nt = src[m], tgt[corr[m]], nrm[corr[m]]
    r = ((vs - vt) * nt).sum(1)
    J = np.concatenate([np.cross(vs, nt), nt], axis=1)  # TransformationEstimation.cpp:83-85
    assert np.allclose(JTJ, J.T @ J, rtol=1e-12) and np.allclose(JTr, J.T @ r, rtol=1e-12, atol=1e-12)
    assert abs(r2 - (r * r).sum()) < 1e-10
    x = np.linalg.solve(J.T @ J, -(J.T @ r))  # utility/Eigen.cpp:122
    assert np.allclose(T, oracle.vec6_to_mat4(x), atol=1e-10)
    assert abs(oracle.compute_rmse(src, tgt, nrm, corr, 2) - np.sqrt((r * r).mean())) < 1e-12
    assert abs(oracle.compute_rmse(src, tgt, None, corr, 1) - np.sqrt(((vs - vt) ** 2).sum(1).mean())) < 1e-12


def test_voxel_down_sample_vs_numpy(oracle):
    rng = np.random.default_rng(7)
    pts = rng.random((4000, 3)) * np.array([2.0, 1.0, 0.5]) - 0.7
    nrm = rng.normal(size=pts.shape)
    nrm[5] = np.nan
    col = rng.random(pts.shape)
    v = 0.11
    op, on, oc, ov = oracle.voxel_down_sample(pts, v, nrm, col)
    vmin = pts.min(0) - v * 0.5  # PointCloud.cpp:279
    idx = np.floor((pts - vmin) / v).astype(np.int64)
    keys, inv = np.unique(idx[:, ::-1], axis=0, return_inverse=True)  # sorted by (z, y, x)
    inv = inv.reshape(-1)
    assert np.array_equal(ov, keys[:, ::-1].astype(np.int32))
    cnt = np.bincount(inv)
    for a, src in ((op, pts), (oc, col)):
        want = np.stack([np.bincount(inv, src[:, d]) for d in range(3)], 1) / cnt[:, None]
        assert np.allclose(a, want, rtol=1e-13)
    good = ~np.isnan(nrm).any(1)
    want = np.stack([np.bincount(inv[good], nrm[good, d], minlength=len(cnt)) for d in range(3)], 1) / cnt[:, None]
    assert np.allclose(on, want, rtol=1e-12, atol=1e-15)
    with pytest.raises(RuntimeError, match="voxel_size <= 0"):
        oracle.voxel_down_sample(pts, 0.0)
    with pytest.raises(RuntimeError, match="too small"):
        oracle.voxel_down_sample(pts * 1e6, 1e-5)


def test_estimate_normals_vs_numpy_eigh(oracle):
    rng = np.random.default_rng(8)
    pts = rng.random((1500, 3)) * np.array([1.0, 1.0, 0.05])
    nrm, cnt = oracle.estimate_normals(pts, oracle.KNN, 20, 0.0, want_counts=True)
    assert np.all(cnt == 20)
    bad = 0
    for i in range(0, 1500, 10):
        d2 = ((pts - pts[i]) ** 2).sum(1)
        nb = np.argsort(d2, kind="stable")[:20]
        w, v = np.linalg.eigh(np.cov(pts[nb].T, bias=True))
        if w[1] - w[0] < 1e-6 * w[2]:
            continue
        bad += abs(abs(v[:, 0] @ nrm[i]) - 1.0) > 1e-8
    assert bad == 0
    # fewer than 3 neighbours -> (0, 0, 1); flip toward an existing normal (EstimateNormals.cpp:336-349)
    assert np.all(oracle.estimate_normals(pts, oracle.KNN, 2, 0.0) == [0, 0, 1])
    prior = rng.normal(size=pts.shape)
    flipped = oracle.estimate_normals(pts, oracle.KNN, 20, 0.0, normals=prior)
    assert np.all((flipped * prior).sum(1) >= 0)
    assert np.allclose(np.abs((flipped * nrm).sum(1)), 1.0, atol=1e-12)
    assert np.array_equal(oracle.fast_eigen3x3(np.zeros((3, 3))), [0, 0, 0])
    assert np.array_equal(oracle.fast_eigen3x3(np.diag([3.0, 1.0, 2.0])), [0, 1, 0])


def test_config1_point_to_point_100k(oracle):
    """BASELINE.json config 1: point-to-point ICP, 100k-point synthetic surface with a known rigid transform, CPU."""
    sc = make_scene(100_000, 100_000, cfg_id=1, L=extent(100_000), theta_deg=0.05)
    s = sc["spacing"]
    res = oracle.registration_icp(sc["source"], sc["target"], 3 * s, None, 1, want_log=True)
    assert res.fitness > 0.999 and res.iterations <= 30
    assert res.log[-1, 1] < res.log[0, 1]  # rmse went down
    # recovered pose near the ground truth: point-to-point ICP between two independent samplings slides along the
    # (locally almost flat) surface, so in-plane translation is only good to about a spacing; height is tight
    assert np.max(np.abs(res.transformation[:3, 3] - sc["T_true"][:3, 3])) < 1.0 * s
    assert abs(res.transformation[2, 3] - sc["T_true"][2, 3]) < 0.2 * s
    assert np.max(np.abs(res.transformation[:3, :3] - sc["T_true"][:3, :3])) < 1e-3
    # the loop bookkeeping of Registration.cpp:155-166
    assert len(res.log) == res.iterations + 1
    with pytest.raises(RuntimeError, match="Invalid max_correspondence_distance"):
        oracle.registration_icp(sc["source"][:10], sc["target"][:10], 0.0)
    with pytest.raises(RuntimeError, match="require pre-computed normal vectors"):
        oracle.registration_icp(sc["source"][:10], sc["target"][:10], 1.0, None, 2, None)


def extent(n):
    from open3d_amd import synthetic
    return synthetic.extent_for(n)


def test_information_matrix(oracle):
    rng = np.random.default_rng(9)
    tgt = rng.random((500, 3))
    src = tgt[:300] + rng.normal(0, 1e-3, (300, 3))
    G = oracle.information_matrix(src, tgt, 0.05, np.eye(4))
    corr = oracle.evaluate_registration(src, tgt, 0.05)[0]
    want = np.zeros((6, 6))
    for j in corr[corr >= 0]:
        x, y, z = tgt[j]
        for row in ([0, z, -y, 1, 0, 0], [-z, 0, x, 0, 1, 0], [y, -x, 0, 0, 0, 1]):  # Registration.cpp:326-346
            want += np.outer(row, row)
    assert np.allclose(G, want, rtol=1e-12)


def test_colored_icp_pieces_vs_numpy(oracle):
    """ColoredICP.cpp:68-177 restated in numpy from the formulas (lstsq with the orthogonality row; explicit 2-row Jacobian)."""
    sc = make_scene(1500, 1800, cfg_id=303)
    s = sc["spacing"]
    tgt, src = sc["target"], sc["source"]
    nrm = oracle.estimate_normals(tgt, oracle.HYBRID, 30, 4 * s)
    rng = np.random.default_rng(5)
    tcol = 0.5 + 0.3 * np.stack([np.sin(2 * tgt[:, 0]), np.cos(1.5 * tgt[:, 1]), np.sin(tgt[:, 0] + tgt[:, 1])], 1)
    scol = rng.uniform(0.2, 0.8, (len(src), 3))
    radius = 5 * s
    grad = oracle.color_gradient(tgt, nrm, tcol, radius, 30)
    it = tcol.sum(1) / 3.0
    thr = oracle.radius_threshold(radius)
    checked = 0
    for k in rng.integers(0, len(tgt), 60):
        d2 = ((tgt - tgt[k]) ** 2).sum(1)
        order = np.lexsort((np.arange(len(tgt)), d2))
        order = order[d2[order] < thr][:30]
        if len(order) < 4:
            assert not grad[k].any()
            continue
        nn = len(order)
        adj = order[1:]
        proj = tgt[adj] - ((tgt[adj] - tgt[k]) @ nrm[k])[:, None] * nrm[k]
        A = np.vstack([proj - tgt[k], (nn - 1) * nrm[k]])
        b = np.concatenate([it[adj] - it[k], [0.0]])
        x = np.linalg.solve(A.T @ A, A.T @ b)
        assert np.allclose(grad[k], x, rtol=1e-8, atol=1e-10 * np.abs(x).max())
        checked += 1
    assert checked > 30

    corr = oracle.correspondences(src, tgt, 3 * s)[0]
    lam = 0.9
    T, JTJ, JTr = oracle.colored_compute(src, scol, tgt, nrm, tcol, grad, corr, lam)
    J_all, r_all = [], []
    slg, slp = np.sqrt(lam), np.sqrt(1 - lam)
    for i in np.nonzero(corr >= 0)[0]:
        t = corr[i]
        vs, vt, nt, dit = src[i], tgt[t], nrm[t], grad[t]
        J_all.append(slg * np.concatenate([np.cross(vs, nt), nt]))
        r_all.append(slg * (vs - vt) @ nt)
        vs_proj = vs - ((vs - vt) @ nt) * nt
        is0 = dit @ (vs_proj - vt) + it[t]
        ditM = -dit @ (np.eye(3) - np.outer(nt, nt))
        J_all.append(slp * np.concatenate([np.cross(vs, ditM), ditM]))
        r_all.append(slp * (scol[i].sum() / 3.0 - is0))
    J, r = np.array(J_all), np.array(r_all)
    assert np.allclose(JTJ, J.T @ J, rtol=1e-10) and np.allclose(JTr, J.T @ r, rtol=1e-9, atol=1e-12)
    x = np.linalg.solve(J.T @ J, -(J.T @ r))
    assert np.allclose(T, oracle.vec6_to_mat4(x), rtol=1e-8, atol=1e-12)
    assert np.isclose(oracle.colored_rmse(src, scol, tgt, nrm, tcol, grad, corr, lam), (r * r).sum(), rtol=1e-10)


def test_fpfh_vs_numpy(oracle):
    """Feature.cpp:39-159 restated with numpy for a few points (brute-force neighbour lists)."""
    sc = make_scene(10, 1500, cfg_id=404)
    pts = sc["target"]
    nrm = oracle.estimate_normals(pts, oracle.KNN, 20)
    k = 12
    feat, spfh = oracle.fpfh(pts, nrm, oracle.KNN, k, 0.0, want_spfh=True)

    def knn(i):
        d2 = ((pts - pts[i]) ** 2).sum(1)
        order = np.lexsort((np.arange(len(pts)), d2))[:k]
        return order, d2[order]

    def pair(p1, n1, p2, n2):
        d = p2 - p1
        L = np.linalg.norm(d)
        if L == 0:
            return np.zeros(4)
        a1, a2 = n1 @ d / L, n2 @ d / L
        if np.arccos(abs(a1)) > np.arccos(abs(a2)):
            n1, n2, d, f2 = n2, n1, -d, -a2
        else:
            f2 = a1
        v = np.cross(d, n1)
        if np.linalg.norm(v) == 0:
            return np.zeros(4)
        v = v / np.linalg.norm(v)
        w = np.cross(n1, v)
        return np.array([np.arctan2(w @ n2, n1 @ n2), v @ n2, f2, L])

    def spfh_of(i):
        idx, _ = knn(i)
        h = np.zeros(33)
        for j in idx[1:]:
            f = pair(pts[i], nrm[i], pts[j], nrm[j])
            b = [int(np.clip(np.floor(11 * (f[0] + np.pi) / (2 * np.pi)), 0, 10)),
                 int(np.clip(np.floor(11 * (f[1] + 1) * 0.5), 0, 10)) + 11,
                 int(np.clip(np.floor(11 * (f[2] + 1) * 0.5), 0, 10)) + 22]
            for bb in b:
                h[bb] += 100.0 / (len(idx) - 1)
        return h

    for i in (0, 7, 333, 1499):
        assert np.allclose(spfh[i], spfh_of(i), atol=1e-9)
        idx, d2 = knn(i)
        acc = np.zeros(33)
        for j, d in zip(idx[1:], d2[1:]):
            if d > 0:
                acc += spfh[j] / d
        s3 = acc.reshape(3, 11).sum(1)
        scale = np.where(s3 != 0, 100.0 / np.where(s3 != 0, s3, 1), 0.0)
        want = (acc.reshape(3, 11) * scale[:, None]).reshape(33) + spfh[i]
        assert np.allclose(feat[i], want, rtol=1e-10, atol=1e-9)


def test_feature_nearest_and_checkers_against_numpy(oracle):
    """RANSAC pieces of the oracle (Registration.cpp:223,237; CorrespondenceChecker.cpp:38-82) against plain numpy.
    Integer-valued features make every summation order exact, so indices (ties -> lowest index = numpy's first argmin)
    and distances must agree exactly."""
    rng = np.random.default_rng(11)
    for dim in (33, 5, 8):
        sf = rng.integers(0, 6, (dim, 300)).astype(np.float64)
        tf = rng.integers(0, 6, (dim, 700)).astype(np.float64)
        tf[:, 500:600] = tf[:, 0:100]
        d = ((sf.T[:, None, :] - tf.T[None, :, :]) ** 2).sum(2)
        idx, d2 = oracle.feature_nearest(sf, tf)
        assert np.array_equal(idx, d.argmin(1)) and np.array_equal(d2, d.min(1))
        q = np.array([7, 299, -1, 300], np.int32)
        idx, d2 = oracle.feature_nearest(sf, tf, q)
        assert np.array_equal(idx[:2], d.argmin(1)[[7, 299]]) and idx[2] == -1 and idx[3] == -1
    # continuous features: same winner wherever the margin is above rounding
    sf, tf = rng.random((33, 200)), rng.random((33, 400))
    d = ((sf.T[:, None, :] - tf.T[None, :, :]) ** 2).sum(2)
    idx, d2 = oracle.feature_nearest(sf, tf)
    assert np.array_equal(idx, d.argmin(1)) and np.allclose(d2, d.min(1), rtol=1e-13)

    src, tgt = rng.random((50, 3)), rng.random((60, 3))
    sn = rng.normal(size=(50, 3)); sn /= np.linalg.norm(sn, axis=1, keepdims=True)
    tn = rng.normal(size=(60, 3)); tn /= np.linalg.norm(tn, axis=1, keepdims=True)
    from open3d_amd.synthetic import rigid_transform
    T = rigid_transform(20.0, (0.1, -0.2, 0.05))
    seen = np.zeros(3, int)
    for _ in range(300):
        c = np.stack([rng.integers(0, 50, 4), rng.integers(0, 60, 4)], 1).astype(np.int32)
        ds = np.linalg.norm(src[c[:, 0]][:, None] - src[c[:, 0]][None], axis=2)
        dt = np.linalg.norm(tgt[c[:, 1]][:, None] - tgt[c[:, 1]][None], axis=2)
        iu = np.triu_indices(4, 1)
        edge = not np.any((ds[iu] < dt[iu] * 0.5) | (dt[iu] < ds[iu] * 0.5))
        moved = src[c[:, 0]] @ T[:3, :3].T + T[:3, 3]
        dist = not np.any(np.linalg.norm(tgt[c[:, 1]] - moved, axis=1) > 0.6)
        nrm = not np.any(((sn[c[:, 0]] @ T[:3, :3].T) * tn[c[:, 1]]).sum(1) < np.cos(1.4))
        assert oracle.ransac_check(src, tgt, c, T, edge_sim=0.5) == edge
        assert oracle.ransac_check(src, tgt, c, T, dist_thr=0.6) == dist
        assert oracle.ransac_check(src, tgt, c, T, normal_angle=1.4, src_normals=sn, tgt_normals=tn) == nrm
        assert oracle.ransac_check(src, tgt, c, T, 0.5, 0.6, 1.4, sn, tn) == (edge and dist and nrm)
        assert oracle.ransac_check(src, tgt, c, T, normal_angle=1.4)   # no normals: passes (:67-72)
        seen += [edge, dist, nrm]
    assert np.all(seen > 0) and np.all(seen < 300)   # every checker gave both answers


def test_self_adjoint_eigen_solver_restatement(oracle):
    """The oracle's restatement of Eigen::SelfAdjointEigenSolver<Matrix3d> (fast_normal_computation=false,
    EstimateNormals.cpp:247-251; Eigen itself is absent: parity unpinned) against numpy's eigh: eigenvalues to rounding,
    eigenvectors orthonormal with A v = w v, ascending order; and the solver's fixed points."""
    rng = np.random.default_rng(0)
    for t in range(4000):
        a = rng.standard_normal((3, 3)) * 10 ** rng.uniform(-6, 3)
        kind = t % 5
        if kind == 1:
            a[:, 2] *= 1e-6
        if kind == 2:
            a = np.diag(rng.standard_normal(3))
        if kind == 3:
            a[2] = a[1]
        A = a @ a.T
        ev, Q, ok = oracle.self_adjoint_eigen3x3(A)
        w = np.linalg.eigvalsh(A)
        sc = max(np.abs(w).max(), 1e-300)
        assert ok and np.all(np.diff(ev) >= 0)
        assert np.abs(ev - w).max() / sc < 1e-13
        assert np.abs(A @ Q - Q * ev).max() / sc < 1e-13
        assert np.abs(Q.T @ Q - np.eye(3)).max() < 1e-13
    ev, Q, ok = oracle.self_adjoint_eigen3x3(np.zeros((3, 3)))
    assert ok and np.array_equal(ev, [0, 0, 0]) and np.array_equal(Q, np.eye(3))
    ev, Q, ok = oracle.self_adjoint_eigen3x3(np.diag([3.0, 1.0, 2.0]))
    assert np.array_equal(ev, [1, 2, 3]) and np.array_equal(Q[:, 0], [0, 1, 0])
    # a normal through ComputeNormal with the solver: a plane z = 0 gives +-(0, 0, 1)
    pts = np.concatenate([rng.random((40, 2)), np.zeros((40, 1))], axis=1)
    n_eig = oracle.estimate_normals(pts, oracle.KNN, 10, 0.0, fast=False)
    assert np.allclose(np.abs(n_eig[:, 2]), 1.0, atol=1e-12)


def test_voxel_radix_order_equals_qsort_order(oracle):
    """The oracle orders its voxels with a stable radix sort of the packed voxel index where that fits 64 bits (the full-size
    GPU tests run it on 100 M points); the order, the sums and the outputs must be those of its qsort path -- also with NaN
    normals, and with a lattice too fine for 64 bits or a NaN point, where it falls back to qsort by itself."""
    rng = np.random.default_rng(11)
    pts = rng.random((200_000, 3)) * np.array([5.0, 3.0, 1.0]) - 2.0
    nrm = rng.standard_normal(pts.shape)
    nrm[::97] = np.nan
    col = rng.random(pts.shape)
    L = oracle.lib()
    try:
        for v in (0.3, 0.05, 0.004, 0.0007, 3e-9):
            L.orc_voxel_force_qsort(0)
            a = oracle.voxel_down_sample(pts, v, nrm, col)
            L.orc_voxel_force_qsort(1)
            b = oracle.voxel_down_sample(pts, v, nrm, col)
            assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)), v
            # numpy's own statement of PointCloud.cpp:288-303 for the points: lexsort by (z, y, x), stable
            vmin = pts.min(0) - 0.5 * v
            key = np.floor((pts - vmin) / v).astype(np.int64)
            order = np.lexsort((key[:, 0], key[:, 1], key[:, 2]))
            ks = key[order]
            first = np.ones(len(ks), bool)
            first[1:] = (ks[1:] != ks[:-1]).any(1)
            assert np.array_equal(a[3], ks[first].astype(np.int32))
            cnt = np.diff(np.append(np.nonzero(first)[0], len(ks)))
            assert np.allclose(a[0], np.add.reduceat(pts[order], np.nonzero(first)[0]) / cnt[:, None], rtol=1e-13, atol=1e-13)
    finally:
        L.orc_voxel_force_qsort(0)

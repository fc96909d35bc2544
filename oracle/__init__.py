"""ctypes front-end of the CPU oracle (oracle/o3d_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (open3d_amd/) never imports it.
Parity status: "parity unpinned" by the reference's own tests (it has none);
see the header of o3d_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


def build(force=False):
    """Compile liboracle.so / liboracle_serial.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "o3d_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


def lib(serial=False):
    name = "liboracle_serial.so" if serial else "liboracle.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_radius_threshold.restype = C.c_double
        L.orc_radius_threshold.argtypes = [C.c_double]
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_build.argtypes = [c_dp, C.c_int64]
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_compute_rmse.restype = C.c_double
        L.orc_voxel_down_sample.restype = C.c_int64
        _LIBS[name] = L
    return _LIBS[name]


def _d(a):
    return a.ctypes.data_as(c_dp)


def _i(a):
    return a.ctypes.data_as(c_ip)


def _pts(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def _mat(T):
    """numpy 4x4 (row-major view) -> column-major flat (Eigen::Matrix4d storage)."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).reshape(16)


def _unmat(flat):
    return np.array(flat, dtype=np.float64).reshape(4, 4).T.copy()


def radius_threshold(r):
    return lib().orc_radius_threshold(float(r))


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return lib().orc_max_threads()


def correspondences(src, tgt, max_dist, brute=False):
    """Registration.cpp:41-91 -> (corr int32[ns], n_corr, sum_d2, fitness, rmse)."""
    L = lib()
    src, tgt = _pts(src), _pts(tgt)
    corr = np.empty(len(src), np.int32)
    nc = C.c_int64()
    sd2, fit, rmse = C.c_double(), C.c_double(), C.c_double()
    kd = None if brute or len(tgt) == 0 else L.orc_kdtree_build(_d(tgt), len(tgt))
    L.orc_correspondences(_d(src), C.c_int64(len(src)), _d(tgt), C.c_int64(len(tgt)), C.c_void_p(kd),
                          C.c_double(max_dist), int(brute or kd is None), _i(corr), C.byref(nc), C.byref(sd2),
                          C.byref(fit), C.byref(rmse))
    if kd:
        L.orc_kdtree_free(kd)
    return corr, nc.value, sd2.value, fit.value, rmse.value


def transform_points(T, pts):
    out = _pts(pts).copy()
    lib().orc_transform_points(_d(_mat(T)), _d(out), C.c_int64(len(out)))
    return out


def p2plane_compute(src, tgt, nrm, corr):
    """TransformationEstimation.cpp:74-98 -> (T 4x4, JTJ 6x6, JTr 6, r2)."""
    src, tgt, nrm = _pts(src), _pts(tgt), _pts(nrm)
    corr = np.ascontiguousarray(corr, np.int32)
    T = np.zeros(16)
    JTJ = np.zeros(36)
    JTr = np.zeros(6)
    r2 = C.c_double()
    lib().orc_p2plane_compute(_d(src), _d(tgt), _d(nrm), _i(corr), C.c_int64(len(src)), _d(T), _d(JTJ), _d(JTr), C.byref(r2))
    return _unmat(T), JTJ.reshape(6, 6), JTr, r2.value


def p2p_compute(src, tgt, corr, with_scaling=False):
    """TransformationEstimation.cpp:49-60 -> T 4x4."""
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, np.int32)
    T = np.zeros(16)
    lib().orc_p2p_compute(_d(src), _d(tgt), _i(corr), C.c_int64(len(src)), int(with_scaling), _d(T))
    return _unmat(T)


def compute_rmse(src, tgt, nrm, corr, est_type):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, np.int32)
    n = _d(_pts(nrm)) if nrm is not None else None
    return lib().orc_compute_rmse(_d(src), _d(tgt), n, _i(corr), C.c_int64(len(src)), int(est_type))


def ldlt_solve(A, b):
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(len(b))
    lib().orc_ldlt_solve(len(b), _d(A), _d(b), _d(x))
    return x


def vec6_to_mat4(v):
    v = np.ascontiguousarray(v, np.float64)
    T = np.zeros(16)
    lib().orc_vec6_to_mat4(_d(v), _d(T))
    return _unmat(T)


def svd3(A):
    A = np.ascontiguousarray(A, np.float64)
    U, S, V = np.zeros(9), np.zeros(3), np.zeros(9)
    lib().orc_svd3(_d(A), _d(U), _d(S), _d(V))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def is_identity(T):
    return bool(lib().orc_is_identity(_d(_mat(T))))


class ICPResult:
    pass


def registration_icp(src, tgt, max_dist, init=None, est_type=1, tgt_normals=None, with_scaling=False,
                     rel_fitness=1e-6, rel_rmse=1e-6, max_iter=30, brute=False, want_log=False, serial=False):
    """Registration.cpp:129-168.  est_type 1 = PointToPoint, 2 = PointToPlane."""
    L = lib(serial)
    src, tgt = _pts(src), _pts(tgt)
    init = np.eye(4) if init is None else init
    nrm = _pts(tgt_normals) if tgt_normals is not None else None
    T = np.zeros(16)
    corr = np.empty(len(src), np.int32)
    nc = C.c_int64()
    fit, rmse, tl = C.c_double(), C.c_double(), C.c_double()
    iters = C.c_int()
    log = np.zeros((max_iter + 1, 20)) if want_log else None
    rc = L.orc_registration_icp(_d(src), C.c_int64(len(src)), _d(tgt), _d(nrm) if nrm is not None else None,
                                C.c_int64(len(tgt)), C.c_double(max_dist), _d(_mat(init)), int(est_type),
                                int(with_scaling), C.c_double(rel_fitness), C.c_double(rel_rmse), int(max_iter),
                                int(brute), _d(T), _i(corr), C.byref(nc), C.byref(fit), C.byref(rmse), C.byref(iters),
                                _d(log) if want_log else None, C.byref(tl))
    if rc == 1:
        raise RuntimeError("[Open3D ERROR] Invalid max_correspondence_distance.")
    if rc == 2:
        raise RuntimeError("[Open3D ERROR] TransformationEstimationPointToPlane and TransformationEstimationColoredICP "
                           "require pre-computed normal vectors for target PointCloud.")
    r = ICPResult()
    r.transformation = _unmat(T)
    r.corr = corr
    r.n_corr = nc.value
    r.fitness = fit.value
    r.inlier_rmse = rmse.value
    r.iterations = iters.value
    r.loop_seconds = tl.value
    r.log = log[: iters.value + 1] if want_log else None
    return r


def evaluate_registration(src, tgt, max_dist, T=None, brute=False):
    """Registration.cpp:118-127 -> (corr, n_corr, fitness, rmse)."""
    L = lib()
    src, tgt = _pts(src), _pts(tgt)
    T = np.eye(4) if T is None else T
    corr = np.empty(len(src), np.int32)
    nc = C.c_int64()
    fit, rmse = C.c_double(), C.c_double()
    L.orc_evaluate_registration(_d(src), C.c_int64(len(src)), _d(tgt), C.c_int64(len(tgt)), C.c_double(max_dist),
                                _d(_mat(T)), int(brute), _i(corr), C.byref(nc), C.byref(fit), C.byref(rmse))
    return corr, nc.value, fit.value, rmse.value


def information_matrix(src, tgt, max_dist, T, brute=False):
    src, tgt = _pts(src), _pts(tgt)
    G = np.zeros(36)
    lib().orc_information_matrix(_d(src), C.c_int64(len(src)), _d(tgt), C.c_int64(len(tgt)), C.c_double(max_dist),
                                 _d(_mat(T)), int(brute), _d(G))
    return G.reshape(6, 6)


def color_gradient(pts, normals, colors, radius, max_nn=30):
    """InitializePointCloudForColoredICP, ColoredICP.cpp:68-123 -> gradients [n,3]."""
    pts, normals, colors = _pts(pts), _pts(normals), _pts(colors)
    g = np.zeros_like(pts)
    lib().orc_color_gradient(_d(pts), _d(normals), _d(colors), C.c_int64(len(pts)), C.c_double(radius), int(max_nn), _d(g))
    return g


def colored_compute(src, src_col, tgt, tgt_nrm, tgt_col, grad, corr, lambda_geometric=0.968):
    """TransformationEstimationForColoredICP::ComputeTransformation, ColoredICP.cpp:125-177 -> (T, JTJ, JTr)."""
    a = [_pts(x) for x in (src, src_col, tgt, tgt_nrm, tgt_col, grad)]
    corr = np.ascontiguousarray(corr, np.int32)
    T, JTJ, JTr = np.zeros(16), np.zeros(36), np.zeros(6)
    lib().orc_colored_compute(*[_d(x) for x in a], _i(corr), C.c_int64(len(a[0])), C.c_double(lambda_geometric), _d(T), _d(JTJ), _d(JTr))
    return _unmat(T), JTJ.reshape(6, 6), JTr


def colored_rmse(src, src_col, tgt, tgt_nrm, tgt_col, grad, corr, lambda_geometric=0.968):
    """ColoredICP.cpp:179-203 (sum of squared residuals)."""
    a = [_pts(x) for x in (src, src_col, tgt, tgt_nrm, tgt_col, grad)]
    corr = np.ascontiguousarray(corr, np.int32)
    f = lib().orc_colored_rmse
    f.restype = C.c_double
    return f(*[_d(x) for x in a], _i(corr), C.c_int64(len(a[0])), C.c_double(lambda_geometric))


def registration_colored_icp(src, src_col, tgt, tgt_nrm, tgt_col, max_dist, init=None, lambda_geometric=0.968, rel_fitness=1e-6,
                             rel_rmse=1e-6, max_iter=30):
    """RegistrationColoredICP, ColoredICP.cpp:205-211 -> ICPResult (+ .gradients)."""
    src, src_col, tgt, tgt_nrm, tgt_col = [_pts(x) for x in (src, src_col, tgt, tgt_nrm, tgt_col)]
    init = np.eye(4) if init is None else init
    T = np.zeros(16)
    corr = np.empty(max(len(src), 1), np.int32)
    nc = C.c_int64()
    fit, rmse = C.c_double(), C.c_double()
    iters = C.c_int()
    grad = np.zeros_like(tgt)
    st = lib().orc_registration_colored_icp(_d(src), _d(src_col), C.c_int64(len(src)), _d(tgt), _d(tgt_nrm), _d(tgt_col),
                                            C.c_int64(len(tgt)), C.c_double(max_dist), _d(_mat(init)), C.c_double(lambda_geometric),
                                            C.c_double(rel_fitness), C.c_double(rel_rmse), int(max_iter), _d(T), _i(corr),
                                            C.byref(nc), C.byref(fit), C.byref(rmse), C.byref(iters), _d(grad))
    if st != 0:
        raise ValueError("orc_registration_colored_icp status %d" % st)
    r = ICPResult()
    r.transformation, r.corr, r.n_corr = _unmat(T), corr[: len(src)], nc.value
    r.fitness, r.inlier_rmse, r.iterations, r.gradients = fit.value, rmse.value, iters.value, grad
    return r


def fpfh(pts, normals, search_type=0, knn=30, radius=0.0, want_spfh=False):
    """ComputeFPFHFeature, Feature.cpp:113-159 -> features [n, 33] (and the SPFH)."""
    pts, normals = _pts(pts), _pts(normals)
    f = np.zeros((len(pts), 33))
    sp = np.zeros((len(pts), 33))
    lib().orc_fpfh(_d(pts), _d(normals), C.c_int64(len(pts)), int(search_type), int(knn), C.c_double(radius), _d(f), _d(sp))
    return (f, sp) if want_spfh else f


def neighbor_query(mode, cloud, queries, knn=1, radius=0.0):
    """PointCloud.cpp:112-132 (mode 0, knn 1), :521-544 (mode 0, knn 2 on itself), :438-449 (mode 1), :406-411 (mode 2)."""
    cloud, queries = _pts(cloud), _pts(queries)
    out = np.zeros(len(queries))
    cnt = np.zeros(max(len(queries), 1), np.int32)
    lib().orc_neighbor_query(int(mode), _d(cloud), C.c_int64(len(cloud)), _d(queries), C.c_int64(len(queries)), int(knn),
                             C.c_double(radius), _d(out), _i(cnt))
    return cnt[: len(queries)] if mode == 2 else out


def feature_nearest(src_feat, tgt_feat, query_ids=None):
    """Registration.cpp:223,237: nearest target column (flann::L2<double>, ties -> lowest index) of the queried source
    columns.  Features are dim x n (Feature::data_).  Returns (index int32[n_q], d2[n_q])."""
    sf = np.ascontiguousarray(np.asarray(src_feat, np.float64).T)
    tf = np.ascontiguousarray(np.asarray(tgt_feat, np.float64).T)
    q = None if query_ids is None else np.ascontiguousarray(query_ids, np.int32)
    nq = len(sf) if q is None else len(q)
    idx = np.full(max(nq, 1), -1, np.int32)
    d2 = np.zeros(max(nq, 1))
    lib().orc_feature_nearest(_d(sf), C.c_int64(len(sf)), _d(tf), C.c_int64(len(tf)), int(sf.shape[1]) if sf.ndim == 2 else 0,
                              _i(q) if q is not None else None, C.c_int64(nq), _i(idx), _d(d2))
    return idx[:nq], d2[:nq]


def ransac_check(src, tgt, corres, T, edge_sim=-1.0, dist_thr=-1.0, normal_angle=-1.0, src_normals=None, tgt_normals=None):
    """CorrespondenceChecker.cpp:38-82 on one correspondence set (n x 2 {source, target})."""
    c = np.ascontiguousarray(corres, np.int32)
    f = lib().orc_ransac_check
    f.restype = C.c_int
    return bool(f(_d(_pts(src)), _d(_pts(src_normals)) if src_normals is not None else None, _d(_pts(tgt)),
                  _d(_pts(tgt_normals)) if tgt_normals is not None else None, _i(c), len(c), _d(_mat(T)), C.c_double(edge_sim),
                  C.c_double(dist_thr), C.c_double(normal_angle)))


def ransac_corr_evaluate(src, tgt, corres, T, max_dist):
    """EvaluateRANSACBasedOnCorrespondence, Registration.cpp:93-116 -> (fitness, rmse, inlier mask)."""
    c = np.ascontiguousarray(corres, np.int32)
    mask = np.zeros(max(len(c), 1), np.int32)
    e = C.c_double()
    f = lib().orc_ransac_corr_evaluate
    f.restype = C.c_int64
    good = f(_d(_pts(src)), _d(_pts(tgt)), _i(c), C.c_int64(len(c)), _d(_mat(T)), C.c_double(max_dist), C.byref(e), _i(mask))
    if good == 0:
        return 0.0, 0.0, mask[: len(c)].astype(bool)
    return good / len(c), float(np.sqrt(e.value / good)), mask[: len(c)].astype(bool)


def voxel_down_sample(pts, voxel, normals=None, colors=None):
    """PointCloud.cpp:273-306 -> (points, normals|None, colors|None, voxel_index int32[M,3]); sorted by (z,y,x)."""
    pts = _pts(pts)
    n = len(pts)
    nrm = _pts(normals) if normals is not None else None
    col = _pts(colors) if colors is not None else None
    op = np.empty((max(n, 1), 3))
    on = np.empty((max(n, 1), 3)) if nrm is not None else None
    oc = np.empty((max(n, 1), 3)) if col is not None else None
    ov = np.empty((max(n, 1), 3), np.int32)
    m = lib().orc_voxel_down_sample(_d(pts), _d(nrm) if nrm is not None else None, _d(col) if col is not None else None,
                                    C.c_int64(n), C.c_double(voxel), _d(op), _d(on) if on is not None else None,
                                    _d(oc) if oc is not None else None, _i(ov))
    if m == -1:
        raise RuntimeError("[Open3D ERROR] [VoxelDownSample] voxel_size <= 0.")
    if m == -2:
        raise RuntimeError("[Open3D ERROR] [VoxelDownSample] voxel_size is too small.")
    return op[:m].copy(), (on[:m].copy() if on is not None else None), (oc[:m].copy() if oc is not None else None), ov[:m].copy()


KNN, RADIUS, HYBRID = 0, 1, 2


def estimate_normals(pts, search_type=KNN, knn=30, radius=0.0, normals=None, want_counts=False, fast=True):
    """EstimateNormals.cpp:320-351 -> normals [n,3] (and neighbour counts).  fast=False: fast_normal_computation=false,
    Eigen's SelfAdjointEigenSolver (:247-251)."""
    pts = _pts(pts)
    n = len(pts)
    has = normals is not None
    out = _pts(normals).copy() if has else np.zeros((n, 3))
    cnt = np.zeros(n, np.int32)
    if n:
        fn = lib().orc_estimate_normals if fast else lib().orc_estimate_normals_eigen
        fn(_d(pts), C.c_int64(n), int(search_type), int(knn), C.c_double(radius), _d(out), int(has), _i(cnt))
    return (out, cnt) if want_counts else out


def neighbors(pts, q, search_type=KNN, knn=30, radius=0.0, brute=False, cap=4096):
    pts = _pts(pts)
    q = np.ascontiguousarray(q, np.float64)
    n_out = cap if search_type == RADIUS else max(knn, 1)
    idx = np.zeros(n_out, np.int32)
    d2 = np.zeros(n_out)
    k = lib().orc_neighbors(_d(pts), C.c_int64(len(pts)), _d(q), int(search_type), int(knn), C.c_double(radius), int(brute),
                            _i(idx), _d(d2), int(cap))
    k = min(k, n_out)
    return idx[:k].copy(), d2[:k].copy()


def compute_normal(pts, indices):
    pts = _pts(pts)
    indices = np.ascontiguousarray(indices, np.int32)
    nv = np.zeros(3)
    cov = np.zeros(9)
    lib().orc_compute_normal(_d(pts), _i(indices), len(indices), _d(nv), _d(cov))
    return nv, cov.reshape(3, 3)


def self_adjoint_eigen3x3(cov):
    """Eigen::SelfAdjointEigenSolver<Matrix3d>::compute as restated in the oracle -> (eigenvalues ascending, eigenvectors as
    columns, converged)."""
    cov = np.ascontiguousarray(cov, np.float64).reshape(3, 3)
    ev = np.zeros(3)
    q = np.zeros(9)
    lib().orc_self_adjoint_eigen3x3.restype = C.c_int
    ok = lib().orc_self_adjoint_eigen3x3(_d(cov), _d(ev), _d(q))
    return ev, q.reshape(3, 3).T.copy(), bool(ok)


def fast_eigen3x3(cov):
    cov = np.ascontiguousarray(cov, np.float64)
    nv = np.zeros(3)
    lib().orc_fast_eigen3x3(_d(cov), _d(nv))
    return nv

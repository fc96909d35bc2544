"""Host-side mirror of open3d::pipelines::registration (reference pipelines/registration/Registration.h:54-152,
TransformationEstimation.h:43-130): same names, argument meaning, defaults and error behaviour; the work is done
by libo3d_amd.so through the C ABI in include/o3d_icp.h."""
import ctypes as C

import numpy as np

from .. import _lib
from ..geometry import PointCloud


class ICPConvergenceCriteria:
    """Registration.h:54-76."""

    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness_ = float(relative_fitness)
        self.relative_rmse_ = float(relative_rmse)
        self.max_iteration_ = int(max_iteration)


class RegistrationResult:
    """Registration.h:107-126.  correspondence_set_ is an (n_corr, 2) int32 array of (source, target) pairs in
    ascending source index (the reference's serial order)."""

    def __init__(self, transformation=None):
        self.transformation_ = np.eye(4) if transformation is None else np.array(transformation, dtype=np.float64)
        self.correspondence_set_ = np.zeros((0, 2), np.int32)
        self.inlier_rmse_ = 0.0
        self.fitness_ = 0.0
        self.iterations_ = 0  # extra: number of estimation updates applied
        self.corr_ = None     # extra: dense int32[n_src] form (-1 = none)


ESTIMATION_UNSPECIFIED, ESTIMATION_COLORED_ICP = 0, 3  # TransformationEstimation.h:45-50


def _est_call(fn_name, est_type, with_scaling, source, target, corres, ctx):
    ctx = ctx or _lib.default_context()
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    nrm = np.ascontiguousarray(target.normals_, dtype=np.float64) if target.HasNormals() else None
    pairs = np.ascontiguousarray(corres, dtype=np.int32).reshape(-1, 2)
    if fn_name == "transformation":
        T = np.zeros(16)
        ctx.check(ctx._lib.o3d_compute_transformation(ctx.handle, _lib.ptr(src), len(src), _lib.ptr(tgt), _lib.ptr(nrm), len(tgt),
                                                      _lib.ptr(pairs), len(pairs), est_type, int(with_scaling), _lib.MEM_HOST,
                                                      T.ctypes.data_as(_lib.c_dp)))
        return _lib.from_colmajor(T)
    r = C.c_double()
    ctx.check(ctx._lib.o3d_compute_rmse(ctx.handle, _lib.ptr(src), len(src), _lib.ptr(tgt), _lib.ptr(nrm), len(tgt),
                                        _lib.ptr(pairs), len(pairs), est_type, _lib.MEM_HOST, C.byref(r)))
    return r.value


class TransformationEstimation:
    """TransformationEstimation.h:57-80.  Subclasses other than the two below run through the generic path of
    RegistrationICP: correspondences on the GPU, ComputeTransformation(source, target, corres) on the host."""

    def GetTransformationEstimationType(self):
        return ESTIMATION_UNSPECIFIED

    def ComputeRMSE(self, source, target, corres, ctx=None):
        raise NotImplementedError

    def ComputeTransformation(self, source, target, corres, ctx=None):
        raise NotImplementedError


class TransformationEstimationPointToPoint(TransformationEstimation):
    """TransformationEstimation.h:85-107 (with_scaling default false)."""

    def __init__(self, with_scaling=False):
        self.with_scaling_ = bool(with_scaling)

    def GetTransformationEstimationType(self):
        return _lib.EST_POINT_TO_POINT

    def ComputeRMSE(self, source, target, corres, ctx=None):
        """TransformationEstimation.cpp:38-47."""
        return _est_call("rmse", _lib.EST_POINT_TO_POINT, False, source, target, corres, ctx)

    def ComputeTransformation(self, source, target, corres, ctx=None):
        """TransformationEstimation.cpp:49-60."""
        return _est_call("transformation", _lib.EST_POINT_TO_POINT, self.with_scaling_, source, target, corres, ctx)


class TransformationEstimationPointToPlane(TransformationEstimation):
    """TransformationEstimation.h:112-130."""
    with_scaling_ = False

    def GetTransformationEstimationType(self):
        return _lib.EST_POINT_TO_PLANE

    def ComputeRMSE(self, source, target, corres, ctx=None):
        """TransformationEstimation.cpp:62-72."""
        return _est_call("rmse", _lib.EST_POINT_TO_PLANE, False, source, target, corres, ctx)

    def ComputeTransformation(self, source, target, corres, ctx=None):
        """TransformationEstimation.cpp:74-98."""
        return _est_call("transformation", _lib.EST_POINT_TO_PLANE, False, source, target, corres, ctx)


def _pairs(corr):
    """Dense corr -> (n_corr, 2) pairs, ascending source index (o3d_expand_correspondences: one pass in C; the numpy
    nonzero / take / stack chain took 10x as long as the whole device-side ICP call on millions of points)."""
    corr = np.ascontiguousarray(corr, dtype=np.int32)
    out = np.empty((len(corr), 2), np.int32)
    m = _lib.load().o3d_expand_correspondences(corr.ctypes.data, len(corr), out.ctypes.data) if len(corr) else 0
    return out[:m]


def _registration_icp_on_device_clouds(source, target, r, init, estimation, criteria, ctx, want_corr=True):
    """RegistrationICP on DevicePointCloud handles (o3d_cloud_registration_icp): nothing but the result crosses PCIe."""
    est = estimation.GetTransformationEstimationType()
    if est not in (_lib.EST_POINT_TO_POINT, _lib.EST_POINT_TO_PLANE):
        raise _lib.O3DError(9, "device-resident clouds take the point-to-point and point-to-plane estimations")
    n = len(source)
    T = np.zeros(16)
    corr = np.empty(max(n, 1), np.int32) if want_corr else None
    nc = C.c_int64()
    fit, rmse = C.c_double(), C.c_double()
    iters = C.c_int()
    ctx.check(ctx._lib.o3d_cloud_registration_icp(ctx.handle, source._h, target._h, float(r), _lib.colmajor(init).ctypes.data_as(_lib.c_dp),
                                                  est, int(getattr(estimation, "with_scaling_", False)), criteria.relative_fitness_,
                                                  criteria.relative_rmse_, criteria.max_iteration_, T.ctypes.data_as(_lib.c_dp),
                                                  _lib.ptr(corr), C.byref(nc), C.byref(fit), C.byref(rmse), C.byref(iters)))
    res = RegistrationResult(_lib.from_colmajor(T))
    if want_corr:
        res.corr_ = corr[:n]
        res.correspondence_set_ = _pairs(res.corr_)
    res.fitness_, res.inlier_rmse_, res.iterations_, res.n_corr_ = fit.value, rmse.value, iters.value, nc.value
    return res


def RegistrationICP(source, target, max_correspondence_distance, init=None, estimation=None, criteria=None, ctx=None):
    """RegistrationICP, Registration.cpp:129-168.  source / target: PointCloud, or DevicePointCloud handles (both)."""
    ctx = ctx or _lib.default_context()
    init = np.eye(4) if init is None else np.asarray(init, dtype=np.float64)
    estimation = estimation or TransformationEstimationPointToPoint(False)
    criteria = criteria or ICPConvergenceCriteria()
    est = estimation.GetTransformationEstimationType()
    if hasattr(source, "_h") and hasattr(target, "_h"):
        return _registration_icp_on_device_clouds(source, target, max_correspondence_distance, init, estimation, criteria, ctx)
    if est not in (_lib.EST_POINT_TO_POINT, _lib.EST_POINT_TO_PLANE):
        return _registration_icp_host_estimation(source, target, max_correspondence_distance, init, estimation, criteria, ctx)
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    nrm = np.ascontiguousarray(target.normals_, dtype=np.float64) if target.HasNormals() else None
    T = np.zeros(16)
    corr = np.empty(max(len(src), 1), np.int32)
    nc = C.c_int64()
    fit, rmse = C.c_double(), C.c_double()
    iters = C.c_int()
    ctx.check(ctx._lib.o3d_registration_icp(ctx.handle, _lib.ptr(src), len(src), _lib.ptr(tgt), _lib.ptr(nrm), len(tgt),
                                            float(max_correspondence_distance),
                                            _lib.colmajor(init).ctypes.data_as(_lib.c_dp), est,
                                            int(getattr(estimation, "with_scaling_", False)), criteria.relative_fitness_,
                                            criteria.relative_rmse_, criteria.max_iteration_, _lib.MEM_HOST,
                                            T.ctypes.data_as(_lib.c_dp), _lib.ptr(corr), C.byref(nc), C.byref(fit),
                                            C.byref(rmse), C.byref(iters)))
    res = RegistrationResult(_lib.from_colmajor(T))
    res.corr_ = corr[: len(src)].copy()
    res.correspondence_set_ = _pairs(res.corr_)
    res.fitness_ = fit.value
    res.inlier_rmse_ = rmse.value
    res.iterations_ = iters.value
    return res


def _transform_points(points, T):
    """PointCloud::Transform, Geometry3D.cpp:77-89 (points only)."""
    h = points @ T[:3, :3].T + T[:3, 3]
    w = points @ T[3, :3] + T[3, 3]
    return h / w[:, None]


def _registration_icp_host_estimation(source, target, max_correspondence_distance, init, estimation, criteria, ctx):
    """Registration.cpp:129-168 for an estimation whose ComputeTransformation stays on the host (ColoredICP, user
    subclasses): the correspondence search runs on the GPU (o3d_icp_pass / o3d_icp_fetch), the virtual call gets the
    transformed source exactly as the reference passes `pcd`, and its result goes back with o3d_icp_set_update."""
    if not (max_correspondence_distance > 0.0):
        raise _lib.O3DError(1, "[Open3D ERROR] Invalid max_correspondence_distance.")
    if estimation.GetTransformationEstimationType() == ESTIMATION_COLORED_ICP and not target.HasNormals():
        raise _lib.O3DError(2, ctx._lib.o3d_status_string(2).decode())
    sess = IcpSession(ctx)
    big = ICPConvergenceCriteria(0.0, 0.0, 1 << 20)  # the loop bound and the convergence test are applied here
    sess.setup(source.points_, target.points_, None, max_correspondence_distance, init, TransformationEstimationPointToPoint(),
               big)
    pcd = PointCloud(np.array(source.points_, dtype=np.float64))
    for name in ("normals_", "colors_"):
        if getattr(source, name, None) is not None and len(getattr(source, name)):
            setattr(pcd, name, np.array(getattr(source, name), dtype=np.float64))
    T = np.array(init, dtype=np.float64)
    if not _is_identity(T):
        pcd.points_ = _transform_points(pcd.points_, T)
        if pcd.HasNormals():
            pcd.normals_ = pcd.normals_ @ T[:3, :3].T
    sess.enqueue_pass()
    res = sess.fetch()
    res.transformation_ = T.copy()
    for i in range(criteria.max_iteration_):
        update = np.asarray(estimation.ComputeTransformation(pcd, target, res.correspondence_set_), dtype=np.float64)
        T = update @ T
        pcd.points_ = _transform_points(pcd.points_, update)
        if pcd.HasNormals():
            pcd.normals_ = pcd.normals_ @ update[:3, :3].T
        sess.set_update(update)
        sess.enqueue_pass()
        backup = res
        res = sess.fetch()
        res.transformation_ = T.copy()
        res.iterations_ = i + 1
        if (abs(backup.fitness_ - res.fitness_) < criteria.relative_fitness_
                and abs(backup.inlier_rmse_ - res.inlier_rmse_) < criteria.relative_rmse_):
            break
    return res


def _is_identity(T, prec=1e-12):
    """Eigen::Matrix4d::isIdentity() with dummy_precision (Registration.cpp:123,150)."""
    T = np.asarray(T, dtype=np.float64)
    d = np.diag(T)
    off = T - np.diag(d)
    return bool(np.all(np.abs(d - 1.0) <= prec * np.minimum(np.abs(d), 1.0)) and np.all(np.abs(off) <= prec))


def RegistrationICPBatch(sources, targets, max_correspondence_distance, inits=None, estimation=None, criteria=None,
                         n_workers=0, want_corr=False, ctx=None):
    """o3d_registration_icp_batch: RegistrationICP (Registration.cpp:129-168) over independent pairs (BASELINE config 5).
    sources / targets: sequences of PointCloud; inits: optional sequence of 4x4.  Returns a list of RegistrationResult."""
    ctx = ctx or _lib.default_context()
    estimation = estimation or TransformationEstimationPointToPoint(False)
    criteria = criteria or ICPConvergenceCriteria()
    est = estimation.GetTransformationEstimationType()
    n = len(sources)
    if len(targets) != n:
        raise ValueError("sources and targets differ in length")
    src = [np.ascontiguousarray(s.points_, dtype=np.float64) for s in sources]
    tgt = [np.ascontiguousarray(t.points_, dtype=np.float64) for t in targets]
    nrm = [np.ascontiguousarray(t.normals_, dtype=np.float64) if t.HasNormals() else None for t in targets]
    if est == _lib.EST_POINT_TO_PLANE and any(x is None for x in nrm):
        raise _lib.O3DError(2, ctx._lib.o3d_status_string(2).decode())
    VP = C.c_void_p * max(n, 1)
    I64 = C.c_int64 * max(n, 1)
    a_src = VP(*[x.ctypes.data for x in src])
    a_tgt = VP(*[x.ctypes.data for x in tgt])
    a_nrm = VP(*[(x.ctypes.data if x is not None else None) for x in nrm])
    n_src = I64(*[len(x) for x in src])
    n_tgt = I64(*[len(x) for x in tgt])
    init_flat = None
    if inits is not None:
        init_flat = np.ascontiguousarray(np.stack([_lib.colmajor(T) for T in inits]))
    corr = [np.empty(max(len(x), 1), np.int32) for x in src] if want_corr else None
    a_corr = VP(*[c.ctypes.data for c in corr]) if want_corr else None
    T = np.zeros((max(n, 1), 16))
    nc = np.zeros(max(n, 1), np.int64)
    fit, rmse = np.zeros(max(n, 1)), np.zeros(max(n, 1))
    iters, status = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    ctx.check(ctx._lib.o3d_registration_icp_batch(
        ctx.handle, n, a_src, n_src, a_tgt, a_nrm if est == _lib.EST_POINT_TO_PLANE else None, n_tgt,
        float(max_correspondence_distance), _lib.ptr(init_flat), est, int(getattr(estimation, "with_scaling_", False)),
        criteria.relative_fitness_, criteria.relative_rmse_, criteria.max_iteration_, _lib.MEM_HOST, int(n_workers),
        _lib.ptr(T), a_corr, _lib.ptr(nc), _lib.ptr(fit), _lib.ptr(rmse), _lib.ptr(iters), _lib.ptr(status)))
    out = []
    for p in range(n):
        r = RegistrationResult(_lib.from_colmajor(T[p]))
        r.fitness_, r.inlier_rmse_, r.iterations_, r.n_corr_ = float(fit[p]), float(rmse[p]), int(iters[p]), int(nc[p])
        if want_corr:
            r.corr_ = corr[p][: len(src[p])]
            r.correspondence_set_ = _pairs(r.corr_)
        out.append(r)
    return out


class Feature:
    """pipelines/registration/Feature.h: data_ is dim x n (one column per point)."""

    def __init__(self, data=None):
        self.data_ = np.zeros((33, 0)) if data is None else data

    def Dimension(self):
        return self.data_.shape[0]

    def Num(self):
        return self.data_.shape[1]


def ComputeFPFHFeature(input, search_param=None, ctx=None):
    """ComputeFPFHFeature, pipelines/registration/Feature.cpp:113-159."""
    from ..geometry import KDTreeSearchParamKNN
    ctx = ctx or _lib.default_context()
    sp = search_param or KDTreeSearchParamKNN()
    if not input.HasNormals():
        raise _lib.O3DError(2, "[Open3D ERROR] [ComputeFPFHFeature] Failed because input point cloud has no normal.")
    pts = np.ascontiguousarray(input.points_, dtype=np.float64)
    nrm = np.ascontiguousarray(input.normals_, dtype=np.float64)
    out = np.zeros((max(len(pts), 1), 33))
    ctx.check(ctx._lib.o3d_compute_fpfh_feature(ctx.handle, _lib.ptr(pts), _lib.ptr(nrm), len(pts), sp.search_type, sp.knn_, sp.radius_,
                                                _lib.MEM_HOST, _lib.ptr(out)))
    return Feature(out[: len(pts)].T)  # a 33 x n view of the column-major storage


class RANSACConvergenceCriteria:
    """Registration.h:87-102."""

    def __init__(self, max_iteration=1000, max_validation=1000):
        self.max_iteration_ = int(max_iteration)
        self.max_validation_ = int(max_validation)


class CorrespondenceCheckerBasedOnEdgeLength:
    """CorrespondenceChecker.h / .cpp:38-51."""
    require_pointcloud_alignment_ = False

    def __init__(self, similarity_threshold=0.9):
        self.similarity_threshold_ = float(similarity_threshold)


class CorrespondenceCheckerBasedOnDistance:
    """CorrespondenceChecker.cpp:53-63."""
    require_pointcloud_alignment_ = True

    def __init__(self, distance_threshold):
        self.distance_threshold_ = float(distance_threshold)


class CorrespondenceCheckerBasedOnNormal:
    """CorrespondenceChecker.cpp:65-82."""
    require_pointcloud_alignment_ = True

    def __init__(self, normal_angle_threshold):
        self.normal_angle_threshold_ = float(normal_angle_threshold)


def _checker_thresholds(checkers):
    """The device path runs the reference's three checkers; several of one kind combine into the strictest one (they
    are ANDed in Registration.cpp:251-266)."""
    edge, dist, ang = -1.0, -1.0, -1.0
    for c in checkers or []:
        if isinstance(c, CorrespondenceCheckerBasedOnEdgeLength):
            edge = max(edge, c.similarity_threshold_)
        elif isinstance(c, CorrespondenceCheckerBasedOnDistance):
            dist = c.distance_threshold_ if dist < 0 else min(dist, c.distance_threshold_)
        elif isinstance(c, CorrespondenceCheckerBasedOnNormal):
            ang = c.normal_angle_threshold_ if ang < 0 else min(ang, c.normal_angle_threshold_)
        else:
            raise _lib.O3DError(5, "a user-defined CorrespondenceChecker has no device path")
    return edge, dist, ang


def _feat(f):
    """Feature (dim x n) -> the column-major storage of Feature::data_: n rows of dim doubles."""
    d = f.data_ if isinstance(f, Feature) else np.asarray(f)
    return np.ascontiguousarray(np.asarray(d, dtype=np.float64).T)


def FeatureNearest(source_feature, target_feature, query_ids=None, ctx=None):
    """The feature matching of RegistrationRANSACBasedOnFeatureMatching (Registration.cpp:223,237): for each queried source
    column the nearest target column (KDTreeFlann::SearchKNN(.., 1)).  Returns (index int32[n_q], squared distance[n_q])."""
    ctx = ctx or _lib.default_context()
    sf, tf = _feat(source_feature), _feat(target_feature)
    q = None if query_ids is None else np.ascontiguousarray(query_ids, np.int32)
    nq = len(sf) if q is None else len(q)
    idx = np.full(max(nq, 1), -1, np.int32)
    d2 = np.zeros(max(nq, 1))
    ctx.check(ctx._lib.o3d_feature_nearest(ctx.handle, _lib.ptr(sf), len(sf), _lib.ptr(tf), len(tf), sf.shape[1], _lib.ptr(q), nq,
                                           _lib.MEM_HOST, _lib.ptr(idx), _lib.ptr(d2)))
    return idx[:nq], d2[:nq]


def RansacHypotheses(source, target, corres, estimation=None, checkers=None, ctx=None):
    """Registration.cpp:250-268 for a batch of candidate correspondence sets (n_hyp x ransac_n x 2): checkers and
    estimation.ComputeTransformation.  Returns (T[n_hyp, 4, 4], ok[n_hyp] bool)."""
    ctx = ctx or _lib.default_context()
    estimation = estimation or TransformationEstimationPointToPoint(False)
    c = np.ascontiguousarray(corres, np.int32)
    assert c.ndim == 3 and c.shape[2] == 2
    n_hyp, ransac_n = c.shape[0], c.shape[1]
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    snrm = np.ascontiguousarray(source.normals_, dtype=np.float64) if source.HasNormals() else None
    tnrm = np.ascontiguousarray(target.normals_, dtype=np.float64) if target.HasNormals() else None
    edge, dist, ang = _checker_thresholds(checkers)
    T = np.zeros((max(n_hyp, 1), 16))
    ok = np.zeros(max(n_hyp, 1), np.int32)
    ctx.check(ctx._lib.o3d_ransac_hypotheses(
        ctx.handle, _lib.ptr(src), _lib.ptr(snrm), len(src), _lib.ptr(tgt), _lib.ptr(tnrm), len(tgt), _lib.ptr(c), n_hyp, ransac_n,
        estimation.GetTransformationEstimationType(), int(getattr(estimation, "with_scaling_", False)), edge, dist, ang,
        _lib.MEM_HOST, _lib.ptr(T), _lib.ptr(ok)))
    return np.stack([_lib.from_colmajor(T[k]) for k in range(n_hyp)]) if n_hyp else np.zeros((0, 4, 4)), ok[:n_hyp].astype(bool)


def RegistrationRANSACBasedOnFeatureMatching(source, target, source_feature, target_feature, max_correspondence_distance,
                                             estimation=None, ransac_n=4, checkers=None, criteria=None, seed=-1, ctx=None):
    """RegistrationRANSACBasedOnFeatureMatching, Registration.cpp:199-301.  seed < 0: std::random_device like the
    reference (not reproducible); seed >= 0: std::mt19937(seed)."""
    ctx = ctx or _lib.default_context()
    estimation = estimation or TransformationEstimationPointToPoint(False)
    criteria = criteria or RANSACConvergenceCriteria()
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    snrm = np.ascontiguousarray(source.normals_, dtype=np.float64) if source.HasNormals() else None
    tnrm = np.ascontiguousarray(target.normals_, dtype=np.float64) if target.HasNormals() else None
    sf, tf = _feat(source_feature), _feat(target_feature)
    if len(sf) != len(src) or len(tf) != len(tgt) or sf.shape[1] != tf.shape[1]:
        raise _lib.O3DError(1, "feature and point counts / dimensions do not match")
    edge, dist, ang = _checker_thresholds(checkers)
    T = np.zeros(16)
    corr = np.empty(max(len(src), 1), np.int32)
    nc = C.c_int64()
    fit, rmse = C.c_double(), C.c_double()
    counts = np.zeros(2, np.int32)
    ctx.check(ctx._lib.o3d_registration_ransac_feature_matching(
        ctx.handle, _lib.ptr(src), _lib.ptr(snrm), len(src), _lib.ptr(tgt), _lib.ptr(tnrm), len(tgt), _lib.ptr(sf), _lib.ptr(tf),
        sf.shape[1], float(max_correspondence_distance), estimation.GetTransformationEstimationType(),
        int(getattr(estimation, "with_scaling_", False)), int(ransac_n), edge, dist, ang, criteria.max_iteration_,
        criteria.max_validation_, int(seed), _lib.MEM_HOST, T.ctypes.data_as(_lib.c_dp), _lib.ptr(corr), C.byref(nc), C.byref(fit),
        C.byref(rmse), _lib.ptr(counts)))
    res = RegistrationResult(_lib.from_colmajor(T))
    res.corr_ = corr[: len(src)].copy()
    res.correspondence_set_ = _pairs(res.corr_)
    res.fitness_, res.inlier_rmse_, res.n_corr_ = fit.value, rmse.value, nc.value
    res.iterations_, res.validations_ = int(counts[0]), int(counts[1])
    return res


def RegistrationRANSACBasedOnCorrespondence(source, target, corres, max_correspondence_distance, estimation=None, ransac_n=6,
                                            criteria=None, seed=-1, ctx=None):
    """RegistrationRANSACBasedOnCorrespondence, Registration.cpp:170-197.  corres: n x 2 {source index, target index}.
    correspondence_set_ of the result = the listed pairs within the distance under the returned transformation."""
    ctx = ctx or _lib.default_context()
    estimation = estimation or TransformationEstimationPointToPoint(False)
    criteria = criteria or RANSACConvergenceCriteria()
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    tnrm = np.ascontiguousarray(target.normals_, dtype=np.float64) if target.HasNormals() else None
    c = np.ascontiguousarray(np.asarray(corres, np.int32).reshape(-1, 2))
    T = np.zeros(16)
    mask = np.zeros(max(len(c), 1), np.int32)
    n_in = C.c_int64()
    fit, rmse = C.c_double(), C.c_double()
    ctx.check(ctx._lib.o3d_registration_ransac_correspondence(
        ctx.handle, _lib.ptr(src), len(src), _lib.ptr(tgt), _lib.ptr(tnrm), len(tgt), _lib.ptr(c), len(c),
        float(max_correspondence_distance), estimation.GetTransformationEstimationType(),
        int(getattr(estimation, "with_scaling_", False)), int(ransac_n), criteria.max_iteration_, criteria.max_validation_, int(seed),
        _lib.MEM_HOST, T.ctypes.data_as(_lib.c_dp), _lib.ptr(mask), C.byref(n_in), C.byref(fit), C.byref(rmse)))
    res = RegistrationResult(_lib.from_colmajor(T))
    res.fitness_, res.inlier_rmse_, res.n_corr_ = fit.value, rmse.value, n_in.value
    res.inlier_mask_ = mask[: len(c)].astype(bool)
    res.correspondence_set_ = c[res.inlier_mask_]
    return res


def EvaluateHypotheses(source, target, max_correspondence_distance, transformations, want_corr=True, ctx=None):
    """The validation step of RegistrationRANSACBasedOnFeatureMatching (Registration.cpp:269-275) for a list of candidate
    transformations.  Returns (fitness[k], inlier_rmse[k], best RegistrationResult or None): `best` is the hypothesis
    RANSAC's comparison keeps."""
    ctx = ctx or _lib.default_context()
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    n = len(transformations)
    Ts = np.ascontiguousarray(np.stack([_lib.colmajor(T) for T in transformations])) if n else np.zeros((0, 16))
    fit, rmse, nc = np.zeros(max(n, 1)), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.int64)
    corr = np.empty(max(len(src), 1), np.int32) if want_corr else None
    best = C.c_int(-1)
    ctx.check(ctx._lib.o3d_evaluate_hypotheses(ctx.handle, _lib.ptr(src), len(src), _lib.ptr(tgt), len(tgt),
                                               float(max_correspondence_distance), _lib.ptr(Ts), n, _lib.MEM_HOST, _lib.ptr(fit),
                                               _lib.ptr(rmse), _lib.ptr(nc), C.byref(best), _lib.ptr(corr)))
    res = None
    if best.value >= 0:
        res = RegistrationResult(np.asarray(transformations[best.value], dtype=np.float64))
        res.fitness_, res.inlier_rmse_, res.index_ = float(fit[best.value]), float(rmse[best.value]), best.value
        if want_corr:
            res.corr_ = corr[: len(src)].copy()
            res.correspondence_set_ = _pairs(res.corr_)
    return fit[:n], rmse[:n], res


def RegistrationColoredICP(source, target, max_distance, init=None, criteria=None, lambda_geometric=0.968, ctx=None,
                           return_gradients=False):
    """RegistrationColoredICP, ColoredICP.cpp:205-211 (colour gradients :68-123, estimation :125-177), on the device."""
    ctx = ctx or _lib.default_context()
    init = np.eye(4) if init is None else np.asarray(init, dtype=np.float64)
    criteria = criteria or ICPConvergenceCriteria()
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    nrm = np.ascontiguousarray(target.normals_, dtype=np.float64) if target.HasNormals() else None
    scol = np.ascontiguousarray(source.colors_, dtype=np.float64) if source.HasColors() else None
    tcol = np.ascontiguousarray(target.colors_, dtype=np.float64) if target.HasColors() else None
    T = np.zeros(16)
    corr = np.empty(max(len(src), 1), np.int32)
    grad = np.zeros((max(len(tgt), 1), 3)) if return_gradients else None
    nc = C.c_int64()
    fit, rmse = C.c_double(), C.c_double()
    iters = C.c_int()
    ctx.check(ctx._lib.o3d_registration_colored_icp(
        ctx.handle, _lib.ptr(src), _lib.ptr(scol), len(src), _lib.ptr(tgt), _lib.ptr(nrm), _lib.ptr(tcol), len(tgt),
        float(max_distance), _lib.colmajor(init).ctypes.data_as(_lib.c_dp), float(lambda_geometric), criteria.relative_fitness_,
        criteria.relative_rmse_, criteria.max_iteration_, _lib.MEM_HOST, T.ctypes.data_as(_lib.c_dp), _lib.ptr(corr), C.byref(nc),
        C.byref(fit), C.byref(rmse), C.byref(iters), _lib.ptr(grad)))
    res = RegistrationResult(_lib.from_colmajor(T))
    res.corr_ = corr[: len(src)].copy()
    res.correspondence_set_ = _pairs(res.corr_)
    res.fitness_, res.inlier_rmse_, res.iterations_ = fit.value, rmse.value, iters.value
    if return_gradients:
        res.color_gradient_ = grad[: len(tgt)]
    return res


def MultiScaleICP(source, target, voxel_sizes, max_correspondence_distances=None, init=None, estimation=None,
                  criteria=None, normal_radius_factor=2.0, normal_max_nn=30, ctx=None, device_resident=True, want_corr=True):
    """Coarse-to-fine registration (BASELINE configs 3 and 4).  Build-defined: the reference has no multi-scale API; this
    is the loop a caller writes over its three calls -- per scale v: VoxelDownSample(v) on both clouds
    (PointCloud.cpp:273-306), EstimateNormals(Hybrid(normal_radius_factor * v, normal_max_nn)) on the target for
    point-to-plane (EstimateNormals.cpp:320-351), RegistrationICP with max distance 2 v (or the given one) starting from the
    previous scale's transformation (Registration.cpp:129-168).  voxel_size <= 0 uses the clouds as they are.
    Returns (final RegistrationResult, [per-scale RegistrationResult])."""
    from ..geometry import DevicePointCloud, KDTreeSearchParamHybrid
    ctx = ctx or _lib.default_context()
    estimation = estimation or TransformationEstimationPointToPlane()
    T = np.eye(4) if init is None else np.asarray(init, dtype=np.float64)
    p2l = estimation.GetTransformationEstimationType() == _lib.EST_POINT_TO_PLANE
    results = []
    if device_resident and estimation.GetTransformationEstimationType() in (_lib.EST_POINT_TO_POINT, _lib.EST_POINT_TO_PLANE):
        # ONE upload per raw cloud; every scale's clouds are made and consumed in HBM (o3d_cloud_*): same results, bit for bit
        d_src = source if hasattr(source, "_h") else DevicePointCloud.upload(source, ctx)
        d_tgt = target if hasattr(target, "_h") else DevicePointCloud.upload(target, ctx)
        for k, v in enumerate(voxel_sizes):
            src_v = d_src.VoxelDownSample(v, ctx=ctx) if v > 0 else d_src
            tgt_v = d_tgt.VoxelDownSample(v, ctx=ctx) if v > 0 else d_tgt
            if p2l and not (v <= 0 and tgt_v.HasNormals()):
                tgt_v.EstimateNormals(KDTreeSearchParamHybrid(normal_radius_factor * v, normal_max_nn), ctx=ctx)
            r = max_correspondence_distances[k] if max_correspondence_distances is not None else 2.0 * v
            crit = criteria[k] if isinstance(criteria, (list, tuple)) else (criteria or ICPConvergenceCriteria())
            res = _registration_icp_on_device_clouds(src_v, tgt_v, r, T, estimation, crit, ctx, want_corr=want_corr)
            res.n_source_, res.n_target_ = len(src_v), len(tgt_v)
            results.append(res)
            T = res.transformation_
            for c in (src_v, tgt_v):
                if c is not d_src and c is not d_tgt:
                    c.free()
        for c, given in ((d_src, source), (d_tgt, target)):
            if c is not given:
                c.free()
        return results[-1], results
    for k, v in enumerate(voxel_sizes):
        src_v = source.VoxelDownSample(v, ctx=ctx) if v > 0 else source
        tgt_v = target.VoxelDownSample(v, ctx=ctx) if v > 0 else target
        if p2l and not (v <= 0 and tgt_v.HasNormals()):
            tgt_v = PointCloud(tgt_v.points_)
            tgt_v.EstimateNormals(KDTreeSearchParamHybrid(normal_radius_factor * v, normal_max_nn), ctx=ctx)
        r = max_correspondence_distances[k] if max_correspondence_distances is not None else 2.0 * v
        crit = criteria[k] if isinstance(criteria, (list, tuple)) else criteria
        res = RegistrationICP(src_v, tgt_v, r, T, estimation, crit, ctx=ctx)
        res.n_source_, res.n_target_ = len(src_v.points_), len(tgt_v.points_)
        results.append(res)
        T = res.transformation_
    return results[-1], results


def GetInformationMatrixFromPointClouds(source, target, max_correspondence_distance, transformation, ctx=None):
    """GetInformationMatrixFromPointClouds, Registration.cpp:303-355.  Returns the 6x6 GTG."""
    ctx = ctx or _lib.default_context()
    T = np.asarray(transformation, dtype=np.float64)
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    G = np.zeros(36)
    ctx.check(ctx._lib.o3d_information_matrix(ctx.handle, _lib.ptr(src), len(src), _lib.ptr(tgt), len(tgt),
                                              float(max_correspondence_distance), _lib.colmajor(T).ctypes.data_as(_lib.c_dp),
                                              _lib.MEM_HOST, G.ctypes.data_as(_lib.c_dp)))
    return G.reshape(6, 6)


def EvaluateRegistration(source, target, max_correspondence_distance, transformation=None, ctx=None):
    """EvaluateRegistration, Registration.cpp:118-127."""
    ctx = ctx or _lib.default_context()
    T = np.eye(4) if transformation is None else np.asarray(transformation, dtype=np.float64)
    src = np.ascontiguousarray(source.points_, dtype=np.float64)
    tgt = np.ascontiguousarray(target.points_, dtype=np.float64)
    corr = np.empty(max(len(src), 1), np.int32)
    nc = C.c_int64()
    fit, rmse = C.c_double(), C.c_double()
    ctx.check(ctx._lib.o3d_evaluate_registration(ctx.handle, _lib.ptr(src), len(src), _lib.ptr(tgt), len(tgt),
                                                 float(max_correspondence_distance),
                                                 _lib.colmajor(T).ctypes.data_as(_lib.c_dp), _lib.MEM_HOST, _lib.ptr(corr),
                                                 C.byref(nc), C.byref(fit), C.byref(rmse)))
    res = RegistrationResult(T)
    res.corr_ = corr[: len(src)].copy()
    res.correspondence_set_ = _pairs(res.corr_)
    res.fitness_ = fit.value
    res.inlier_rmse_ = rmse.value
    return res


class IcpSession:
    """Stepwise, device-resident form of the same loop (o3d_icp_setup / pass / solve / finish): the caller owns the
    per-iteration exchange.  Buffers may be numpy arrays (host) or torch CUDA tensors (device, float64, contiguous)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self._keep = None
        self.n_src = 0
        self.max_iter = 0

    def setup(self, src, tgt, tgt_normals, max_correspondence_distance, init=None, estimation=None, criteria=None,
              n_src_global=None):
        estimation = estimation or TransformationEstimationPointToPoint(False)
        criteria = criteria or ICPConvergenceCriteria()
        init = np.eye(4) if init is None else np.asarray(init, dtype=np.float64)
        on_device = hasattr(src, "data_ptr")
        mem = _lib.MEM_DEVICE if on_device else _lib.MEM_HOST
        if not on_device:
            src = np.ascontiguousarray(src, dtype=np.float64)
            tgt = np.ascontiguousarray(tgt, dtype=np.float64)
            tgt_normals = np.ascontiguousarray(tgt_normals, dtype=np.float64) if tgt_normals is not None else None
        self._keep = (src, tgt, tgt_normals)
        self.n_src = int(src.shape[0])
        self.max_iter = criteria.max_iteration_
        self.mem = mem
        c = self.ctx
        c.check(c._lib.o3d_icp_setup(c.handle, _lib.ptr(src), self.n_src, _lib.ptr(tgt), _lib.ptr(tgt_normals),
                                     int(tgt.shape[0]), float(max_correspondence_distance),
                                     _lib.colmajor(init).ctypes.data_as(_lib.c_dp),
                                     estimation.GetTransformationEstimationType(),
                                     int(getattr(estimation, "with_scaling_", False)), criteria.relative_fitness_,
                                     criteria.relative_rmse_, criteria.max_iteration_, mem,
                                     int(n_src_global if n_src_global is not None else self.n_src)))
        return self

    def enqueue_pass(self):
        c = self.ctx
        c.check(c._lib.o3d_icp_pass(c.handle))

    def enqueue_step(self):
        """pass + estimate for the unsharded loop (one fused finalize+solve launch)."""
        c = self.ctx
        c.check(c._lib.o3d_icp_step(c.handle))

    def enqueue_pass_to(self, dev_sums_ptr):
        c = self.ctx
        c.check(c._lib.o3d_icp_pass_to(c.handle, C.c_void_p(dev_sums_ptr)))

    def set_update(self, update):
        """Hand a host-computed update back to the loop (o3d_icp_set_update, Registration.cpp:158-159)."""
        c = self.ctx
        c.check(c._lib.o3d_icp_set_update(c.handle, _lib.colmajor(update).ctypes.data_as(_lib.c_dp)))

    def fetch(self, want_corr=True):
        """Correspondences, fitness and rmse of the last o3d_icp_pass (Registration.cpp:82-89)."""
        c = self.ctx
        corr = np.empty(max(self.n_src, 1), np.int32) if want_corr else None
        nc = C.c_int64()
        fit, rmse = C.c_double(), C.c_double()
        c.check(c._lib.o3d_icp_fetch(c.handle, _lib.MEM_HOST, _lib.ptr(corr), C.byref(nc), C.byref(fit), C.byref(rmse)))
        res = RegistrationResult()
        if want_corr:
            res.corr_ = corr[: self.n_src].copy()
            res.correspondence_set_ = _pairs(res.corr_)
        res.fitness_, res.inlier_rmse_, res.n_corr_ = fit.value, rmse.value, nc.value
        return res

    def profile(self, enable=True):
        c = self.ctx
        c.check(c._lib.o3d_icp_profile(c.handle, int(enable)))

    def profile_read(self):
        """(mean k_icp_pass duration in ms, number of passes recorded) from HIP events on the launch stream."""
        c = self.ctx
        ms, n = C.c_double(), C.c_int()
        c.check(c._lib.o3d_icp_profile_read(c.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_read2(self):
        """((search ms, accumulate ms), passes recorded): mean kernel durations from HIP events on the launch stream."""
        c = self.ctx
        ms = np.zeros(2)
        n = C.c_int()
        c.check(c._lib.o3d_icp_profile_read2(c.handle, ms.ctypes.data_as(_lib.c_dp), C.byref(n)))
        return (float(ms[0]), float(ms[1])), n.value

    def profile_read_all(self):
        """Kernel milliseconds of every recorded pass, in order (call before profile_read / profile_read2, which clear)."""
        c = self.ctx
        ms = np.zeros(340)
        n = C.c_int()
        c.check(c._lib.o3d_icp_profile_read_all(c.handle, ms.ctypes.data_as(_lib.c_dp), 340, C.byref(n)))
        return ms[: n.value].copy()

    def local_sums_ptr(self):
        return self.ctx._lib.o3d_icp_local_sums(self.ctx.handle)

    def enqueue_solve(self, all_sums_ptr=None, n_ranks=1):
        c = self.ctx
        p = all_sums_ptr if all_sums_ptr is not None else self.local_sums_ptr()
        c.check(c._lib.o3d_icp_solve(c.handle, C.c_void_p(p), int(n_ranks)))

    def poll(self):
        c = self.ctx
        done, passes = C.c_int(), C.c_int()
        c.check(c._lib.o3d_icp_poll(c.handle, C.byref(done), C.byref(passes)))
        return bool(done.value), passes.value

    def finish(self, want_corr=True, corr_out=None):
        c = self.ctx
        T = np.zeros(16)
        nc = C.c_int64()
        fit, rmse = C.c_double(), C.c_double()
        iters = C.c_int()
        corr = None
        mem = _lib.MEM_HOST
        if corr_out is not None:
            corr, mem = corr_out, (_lib.MEM_DEVICE if hasattr(corr_out, "data_ptr") else _lib.MEM_HOST)
        elif want_corr:
            corr = np.empty(max(self.n_src, 1), np.int32)
        c.check(c._lib.o3d_icp_finish(c.handle, mem, T.ctypes.data_as(_lib.c_dp), _lib.ptr(corr), C.byref(nc), C.byref(fit),
                                      C.byref(rmse), C.byref(iters)))
        res = RegistrationResult(_lib.from_colmajor(T))
        if corr is not None and not hasattr(corr, "data_ptr"):
            res.corr_ = corr[: self.n_src].copy()
            res.correspondence_set_ = _pairs(res.corr_)
        else:
            res.corr_ = corr
        res.fitness_ = fit.value
        res.inlier_rmse_ = rmse.value
        res.iterations_ = iters.value
        res.n_corr_ = nc.value
        return res

    def log(self):
        """Per-pass history rows {fitness, rmse, n_corr, sum_d2, T[16] col-major, sums[32]}."""
        c = self.ctx
        buf = np.zeros((self.max_iter + 1, _lib.LOG_STRIDE))
        rows = C.c_int()
        c.check(c._lib.o3d_icp_get_log(c.handle, buf.ctypes.data_as(_lib.c_dp), self.max_iter + 1, C.byref(rows)))
        return buf[: rows.value]

    def setup_ms(self):
        c = self.ctx
        ms = np.zeros(2)
        c.check(c._lib.o3d_icp_get_setup_ms(c.handle, ms.ctypes.data_as(_lib.c_dp)))
        return ms

    def run(self, check_every=6):
        """Single-GPU loop: Registration.cpp:154-166."""
        self.enqueue_step()
        it = 0
        while it < self.max_iter:
            upto = min(it + check_every, self.max_iter)
            while it < upto:
                self.enqueue_step()
                it += 1
            if it < self.max_iter and self.poll()[0]:
                break
        return self

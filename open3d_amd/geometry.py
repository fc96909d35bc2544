"""Host-side mirror of open3d::geometry for the ICP path (reference geometry/PointCloud.h:54-335,
geometry/KDTreeSearchParam.h:35-107).  All computation happens in libo3d_amd.so on the GPU."""
import ctypes as C

import numpy as np

from . import _lib


class KDTreeSearchParam:
    search_type = None


class KDTreeSearchParamKNN(KDTreeSearchParam):
    """geometry/KDTreeSearchParam.h:62-75 (default knn = 30)."""
    search_type = _lib.SEARCH_KNN

    def __init__(self, knn=30):
        self.knn_ = int(knn)
        self.radius_ = 0.0


class KDTreeSearchParamRadius(KDTreeSearchParam):
    """geometry/KDTreeSearchParam.h:77-90."""
    search_type = _lib.SEARCH_RADIUS

    def __init__(self, radius):
        self.radius_ = float(radius)
        self.knn_ = 0


class KDTreeSearchParamHybrid(KDTreeSearchParam):
    """geometry/KDTreeSearchParam.h:92-107."""
    search_type = _lib.SEARCH_HYBRID

    def __init__(self, radius, max_nn):
        self.radius_ = float(radius)
        self.knn_ = int(max_nn)

    @property
    def max_nn_(self):
        return self.knn_


def _as_points(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("expected an (N, 3) float64 array")
    return a


class PointCloud:
    """points_/normals_/colors_ as (N,3) float64 arrays (std::vector<Eigen::Vector3d> layout, PointCloud.h:330-334)."""

    def __init__(self, points=None, normals=None, colors=None):
        self.points_ = _as_points(points) if points is not None else np.zeros((0, 3))
        self.normals_ = _as_points(normals) if normals is not None else np.zeros((0, 3))
        self.colors_ = _as_points(colors) if colors is not None else np.zeros((0, 3))

    def HasPoints(self):
        return len(self.points_) > 0

    def HasNormals(self):
        return len(self.points_) > 0 and len(self.normals_) == len(self.points_)

    def HasColors(self):
        return len(self.points_) > 0 and len(self.colors_) == len(self.points_)

    def VoxelDownSample(self, voxel_size, ctx=None, return_voxel_index=False):
        """PointCloud::VoxelDownSample, geometry/PointCloud.cpp:273-306.  Output ordered by voxel index (z, y, x)."""
        ctx = ctx or _lib.default_context()
        n = len(self.points_)
        nrm = self.normals_ if self.HasNormals() else None
        col = self.colors_ if self.HasColors() else None
        op = np.empty((max(n, 1), 3))
        on = np.empty((max(n, 1), 3)) if nrm is not None else None
        oc = np.empty((max(n, 1), 3)) if col is not None else None
        ov = np.empty((max(n, 1), 3), np.int32) if return_voxel_index else None
        m = C.c_int64()
        ctx.check(ctx._lib.o3d_voxel_down_sample(ctx.handle, _lib.ptr(self.points_), _lib.ptr(nrm), _lib.ptr(col), n,
                                                 float(voxel_size), _lib.MEM_HOST, _lib.ptr(op), _lib.ptr(on), _lib.ptr(oc),
                                                 _lib.ptr(ov), C.byref(m)))
        m = m.value
        # (prefix views, no copies: the untouched tail of the np.empty buffers never becomes resident)
        out = PointCloud(op[:m], on[:m] if on is not None else None, oc[:m] if oc is not None else None)
        return (out, ov[:m]) if return_voxel_index else out

    def EstimateNormals(self, search_param=None, fast_normal_computation=True, ctx=None):
        """PointCloud::EstimateNormals, geometry/EstimateNormals.cpp:320-351 (fast_normal_computation=False: Eigen's
        SelfAdjointEigenSolver instead of FastEigen3x3, :245-251)."""
        ctx = ctx or _lib.default_context()
        ctx.check(ctx._lib.o3d_set_fast_normal_computation(ctx.handle, int(bool(fast_normal_computation))))
        sp = search_param or KDTreeSearchParamKNN()
        n = len(self.points_)
        has = self.HasNormals()
        nrm = np.ascontiguousarray(self.normals_) if has else np.zeros((n, 3))
        ctx.check(ctx._lib.o3d_estimate_normals(ctx.handle, _lib.ptr(self.points_), n, sp.search_type, sp.knn_, sp.radius_,
                                                _lib.MEM_HOST, _lib.ptr(nrm), int(has)))
        self.normals_ = nrm
        return self

    def SelectByIndex(self, indices):
        """PointCloud::SelectByIndex (points, normals, colours of the listed indices, in that order)."""
        idx = np.asarray(indices, dtype=np.int64)
        return PointCloud(self.points_[idx], self.normals_[idx] if self.HasNormals() else None,
                          self.colors_[idx] if self.HasColors() else None)

    def ComputePointCloudDistance(self, target, ctx=None):
        """PointCloud::ComputePointCloudDistance, geometry/PointCloud.cpp:112-132."""
        ctx = ctx or _lib.default_context()
        out = np.zeros(len(self.points_))
        ctx.check(ctx._lib.o3d_compute_point_cloud_distance(ctx.handle, _lib.ptr(self.points_), len(self.points_),
                                                            _lib.ptr(target.points_), len(target.points_), _lib.MEM_HOST,
                                                            _lib.ptr(out)))
        return out

    def ComputeNearestNeighborDistance(self, ctx=None):
        """PointCloud::ComputeNearestNeighborDistance, geometry/PointCloud.cpp:521-544."""
        ctx = ctx or _lib.default_context()
        out = np.zeros(len(self.points_))
        if len(self.points_) < 2:
            return out
        ctx.check(ctx._lib.o3d_compute_nearest_neighbor_distance(ctx.handle, _lib.ptr(self.points_), len(self.points_),
                                                                 _lib.MEM_HOST, _lib.ptr(out)))
        return out

    def RemoveRadiusOutliers(self, nb_points, search_radius, ctx=None):
        """PointCloud::RemoveRadiusOutliers, geometry/PointCloud.cpp:394-419 -> (cloud, indices)."""
        if nb_points < 1 or not (search_radius > 0):
            raise _lib.O3DError(5, "[Open3D ERROR] [RemoveRadiusOutliers] Illegal input parameters,"
                                   "number of points and radius must be positive")
        ctx = ctx or _lib.default_context()
        cnt = np.zeros(max(len(self.points_), 1), np.int32)
        ctx.check(ctx._lib.o3d_radius_neighbor_count(ctx.handle, _lib.ptr(self.points_), len(self.points_), float(search_radius),
                                                     _lib.MEM_HOST, _lib.ptr(cnt)))
        indices = np.nonzero(cnt[: len(self.points_)] > nb_points)[0]
        return self.SelectByIndex(indices), indices

    def RemoveStatisticalOutliers(self, nb_neighbors, std_ratio, ctx=None):
        """PointCloud::RemoveStatisticalOutliers, geometry/PointCloud.cpp:421-468 -> (cloud, indices).  The k-nearest mean
        distances come from the device; the cloud statistics are the reference's serial sums (:453-461)."""
        if nb_neighbors < 1 or not (std_ratio > 0):
            raise _lib.O3DError(5, "[Open3D ERROR] [RemoveStatisticalOutliers] Illegal input parameters, number "
                                   "of neighbors and standard deviation ratio must be positive")
        n = len(self.points_)
        if n == 0:
            return PointCloud(), np.zeros(0, np.int64)
        ctx = ctx or _lib.default_context()
        avg = np.zeros(n)
        ctx.check(ctx._lib.o3d_knn_mean_distance(ctx.handle, _lib.ptr(self.points_), n, int(nb_neighbors), _lib.MEM_HOST,
                                                 _lib.ptr(avg)))
        valid = int(np.count_nonzero(avg >= 0))  # mean = -1 marks "no neighbour" (:442-448)
        if valid == 0:
            return PointCloud(), np.zeros(0, np.int64)
        pos = avg > 0
        cloud_mean = float(np.cumsum(avg[pos])[-1]) / valid if pos.any() else 0.0  # sequential sum, as std::accumulate
        dev = avg[pos] - cloud_mean
        sq_sum = float(np.cumsum(dev * dev)[-1]) if pos.any() else 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            std_dev = float(np.sqrt(np.float64(sq_sum) / np.float64(valid - 1)))
        thr = cloud_mean + std_ratio * std_dev
        indices = np.nonzero((avg > 0) & (avg < thr))[0]
        return self.SelectByIndex(indices), indices



class DevicePointCloud:
    """A PointCloud whose points_ / normals_ / colors_ stay in HBM across calls (o3d_cloud, include/o3d_icp.h): what the
    reference gets for free by passing `const PointCloud &` from VoxelDownSample to EstimateNormals to RegistrationICP
    (PointCloud.h:130,192, Registration.h:149-152).  Uploaded once; every method below is the PointCloud method of the same
    name without a trip over PCIe.  Results are those of the host-buffer calls bit for bit."""

    def __init__(self, handle, ctx):
        self._h = handle
        self.ctx = ctx

    @classmethod
    def upload(cls, cloud, ctx=None):
        """From a PointCloud (or an (N, 3) array of points)."""
        ctx = ctx or _lib.default_context()
        if not isinstance(cloud, PointCloud):
            cloud = PointCloud(cloud)
        h = C.c_void_p()
        ctx.check(ctx._lib.o3d_cloud_upload(ctx.handle, _lib.ptr(cloud.points_), _lib.ptr(cloud.normals_ if cloud.HasNormals() else None),
                                            _lib.ptr(cloud.colors_ if cloud.HasColors() else None), len(cloud.points_), _lib.MEM_HOST,
                                            C.byref(h)))
        return cls(h, ctx)

    def __len__(self):
        return int(self.ctx._lib.o3d_cloud_size(self._h)) if self._h else 0

    def HasPoints(self):
        return len(self) > 0

    def HasNormals(self):
        return bool(self._h) and bool(self.ctx._lib.o3d_cloud_has_normals(self._h))

    def HasColors(self):
        return bool(self._h) and bool(self.ctx._lib.o3d_cloud_has_colors(self._h))

    def VoxelDownSample(self, voxel_size, ctx=None):
        """PointCloud::VoxelDownSample, geometry/PointCloud.cpp:273-306 -> a new DevicePointCloud."""
        ctx = ctx or self.ctx
        h = C.c_void_p()
        ctx.check(ctx._lib.o3d_cloud_voxel_down_sample(ctx.handle, self._h, float(voxel_size), C.byref(h)))
        return DevicePointCloud(h, ctx)

    def EstimateNormals(self, search_param=None, fast_normal_computation=True, ctx=None):
        """PointCloud::EstimateNormals, geometry/EstimateNormals.cpp:320-351, in place."""
        ctx = ctx or self.ctx
        ctx.check(ctx._lib.o3d_set_fast_normal_computation(ctx.handle, int(bool(fast_normal_computation))))
        sp = search_param or KDTreeSearchParamKNN()
        ctx.check(ctx._lib.o3d_cloud_estimate_normals(ctx.handle, self._h, sp.search_type, sp.knn_, sp.radius_))
        return self

    def download(self):
        """-> PointCloud (host copies of whatever the cloud holds)."""
        n = len(self)
        pts = np.empty((n, 3))
        nrm = np.empty((n, 3)) if self.HasNormals() else None
        col = np.empty((n, 3)) if self.HasColors() else None
        self.ctx.check(self.ctx._lib.o3d_cloud_download(self.ctx.handle, self._h, _lib.ptr(pts), _lib.ptr(nrm), _lib.ptr(col)))
        return PointCloud(pts, nrm, col)

    def free(self):
        if self._h:
            self.ctx._lib.o3d_cloud_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

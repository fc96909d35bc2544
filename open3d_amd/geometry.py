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
        ov = np.empty((max(n, 1), 3), np.int32)
        m = C.c_int64()
        ctx.check(ctx._lib.o3d_voxel_down_sample(ctx.handle, _lib.ptr(self.points_), _lib.ptr(nrm), _lib.ptr(col), n,
                                                 float(voxel_size), _lib.MEM_HOST, _lib.ptr(op), _lib.ptr(on), _lib.ptr(oc),
                                                 _lib.ptr(ov), C.byref(m)))
        m = m.value
        out = PointCloud(op[:m].copy(), on[:m].copy() if on is not None else None, oc[:m].copy() if oc is not None else None)
        return (out, ov[:m].copy()) if return_voxel_index else out

    def EstimateNormals(self, search_param=None, fast_normal_computation=True, ctx=None):
        """PointCloud::EstimateNormals, geometry/EstimateNormals.cpp:320-351."""
        if not fast_normal_computation:
            raise _lib.O3DError(_lib.load().o3d_status_string(9) and 9,
                                "fast_normal_computation=False (Eigen SelfAdjointEigenSolver) is not on the device path")
        ctx = ctx or _lib.default_context()
        sp = search_param or KDTreeSearchParamKNN()
        n = len(self.points_)
        has = self.HasNormals()
        nrm = np.ascontiguousarray(self.normals_) if has else np.zeros((n, 3))
        ctx.check(ctx._lib.o3d_estimate_normals(ctx.handle, _lib.ptr(self.points_), n, sp.search_type, sp.knn_, sp.radius_,
                                                _lib.MEM_HOST, _lib.ptr(nrm), int(has)))
        self.normals_ = nrm
        return self

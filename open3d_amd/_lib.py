"""ctypes binding of libo3d_amd.so (the C ABI declared in include/o3d_icp.h).

There is no CPU fallback anywhere in this package: if the shared library is
missing, or no HIP device is usable, every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libo3d_amd.so")

MEM_HOST, MEM_DEVICE = 0, 1
EST_POINT_TO_POINT, EST_POINT_TO_PLANE, EST_COLORED_ICP = 1, 2, 3
SEARCH_KNN, SEARCH_RADIUS, SEARCH_HYBRID = 0, 1, 2
LOG_STRIDE = 52
STATUS_OK = 0

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)

_SIGNATURES = {
    "o3d_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "o3d_ctx_create_multi": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "o3d_ctx_num_devices": (C.c_int, [C.c_void_p]),
    "o3d_ctx_destroy": (None, [C.c_void_p]),
    "o3d_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "o3d_ctx_use_own_stream": (C.c_int, [C.c_void_p]),
    "o3d_last_error": (C.c_char_p, [C.c_void_p]),
    "o3d_status_string": (C.c_char_p, [C.c_int]),
    "o3d_device_count": (C.c_int, []),
    "o3d_registration_icp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                       c_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, c_dp, C.c_void_p,
                                       C.POINTER(C.c_int64), c_dp, c_dp, C.POINTER(C.c_int)]),
    "o3d_evaluate_registration": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, c_dp,
                                            C.c_int, C.c_void_p, C.POINTER(C.c_int64), c_dp, c_dp]),
    "o3d_evaluate_hypotheses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "o3d_registration_colored_icp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int64, C.c_double, c_dp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                               c_dp, C.c_void_p, C.POINTER(C.c_int64), c_dp, c_dp, C.POINTER(C.c_int), C.c_void_p]),
    "o3d_registration_icp_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "o3d_compute_transformation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_int64, C.c_int, C.c_int, C.c_int, c_dp]),
    "o3d_compute_transformation_noalloc": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                                     C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "o3d_compute_rmse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                   C.c_int, C.c_int, c_dp]),
    "o3d_information_matrix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_double, c_dp, C.c_int,
                                         c_dp]),
    "o3d_icp_fetch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64), c_dp, c_dp]),
    "o3d_icp_set_update": (C.c_int, [C.c_void_p, c_dp]),
    "o3d_icp_setup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, c_dp,
                                C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int64]),
    "o3d_icp_pass": (C.c_int, [C.c_void_p]),
    "o3d_icp_local_sums": (C.c_void_p, [C.c_void_p]),
    "o3d_icp_step": (C.c_int, [C.c_void_p]),
    "o3d_icp_pass_to": (C.c_int, [C.c_void_p, C.c_void_p]),
    "o3d_debug_set": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "o3d_debug_tile_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    "o3d_debug_grid_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "o3d_debug_inject_bound_violation": (C.c_int, [C.c_void_p, C.c_uint]),
    "o3d_debug_get_source_perm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "o3d_debug_get_match_positions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "o3d_icp_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "o3d_icp_profile_read": (C.c_int, [C.c_void_p, c_dp, C.POINTER(C.c_int)]),
    "o3d_icp_profile_read2": (C.c_int, [C.c_void_p, c_dp, C.POINTER(C.c_int)]),
    "o3d_icp_profile_read_all": (C.c_int, [C.c_void_p, c_dp, C.c_int, C.POINTER(C.c_int)]),
    "o3d_icp_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "o3d_icp_poll": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "o3d_icp_finish": (C.c_int, [C.c_void_p, C.c_int, c_dp, C.c_void_p, C.POINTER(C.c_int64), c_dp, c_dp,
                                 C.POINTER(C.c_int)]),
    "o3d_icp_get_log": (C.c_int, [C.c_void_p, c_dp, C.c_int, C.POINTER(C.c_int)]),
    "o3d_icp_get_setup_ms": (C.c_int, [C.c_void_p, c_dp]),
    "o3d_compute_point_cloud_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "o3d_compute_nearest_neighbor_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "o3d_radius_neighbor_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_void_p]),
    "o3d_knn_mean_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "o3d_compute_fpfh_feature": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int,
                                           C.c_void_p]),
    "o3d_feature_nearest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                      C.c_int, C.c_void_p, C.c_void_p]),
    "o3d_ransac_hypotheses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "o3d_registration_ransac_feature_matching": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int,
                                                           C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int64,
                                                           C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), c_dp, c_dp,
                                                           C.c_void_p]),
    "o3d_registration_ransac_correspondence": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                                         C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                                         C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                                         C.POINTER(C.c_int64), c_dp, c_dp]),
    "o3d_estimate_normals_shard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                             C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]),
    "o3d_estimate_normals_assemble": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "o3d_synthetic_pair": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_double, C.c_double, c_dp, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "o3d_debug_copy_check": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_int64)]),
    "o3d_expand_correspondences": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p]),
    "o3d_debug_copy_bench": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, c_dp]),
    "o3d_voxel_down_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "o3d_set_fast_normal_computation": (C.c_int, [C.c_void_p, C.c_int]),
    "o3d_estimate_normals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                       C.c_int]),
    "o3d_cloud_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "o3d_cloud_free": (None, [C.c_void_p]),
    "o3d_cloud_size": (C.c_int64, [C.c_void_p]),
    "o3d_cloud_has_normals": (C.c_int, [C.c_void_p]),
    "o3d_cloud_has_colors": (C.c_int, [C.c_void_p]),
    "o3d_cloud_points": (C.c_void_p, [C.c_void_p]),
    "o3d_cloud_normals": (C.c_void_p, [C.c_void_p]),
    "o3d_cloud_colors": (C.c_void_p, [C.c_void_p]),
    "o3d_cloud_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "o3d_cloud_voxel_down_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_void_p)]),
    "o3d_cloud_estimate_normals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double]),
    "o3d_cloud_registration_icp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, c_dp, C.c_int, C.c_int, C.c_double,
                                             C.c_double, C.c_int, c_dp, C.c_void_p, C.POINTER(C.c_int64), c_dp, c_dp,
                                             C.POINTER(C.c_int)]),
    "o3d_cloud_evaluate_registration": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, c_dp, C.c_void_p,
                                                  C.POINTER(C.c_int64), c_dp, c_dp]),
}

_lib = None


def exported_symbols():
    """Names every build of the library must export (checked by the CPU test-suite)."""
    return sorted(_SIGNATURES)


def load():
    """dlopen libo3d_amd.so.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `make -C open3d_amd/csrc` (or __graft_entry__.build()); "
                               "open3d_amd has no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Loading torch
        # first makes libo3d_amd.so bind to that copy, so device pointers and streams can be shared with torch tensors;
        # the other order leaves torch unable to initialise its GPU context.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class O3DError(RuntimeError):
    """Mirror of the std::runtime_error thrown by utility::LogError (utility/Console.h:82-87)."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def colmajor(T):
    """4x4 numpy (math layout) -> 16 doubles in Eigen::Matrix4d storage order."""
    T = np.asarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("expected a 4x4 matrix")
    return np.ascontiguousarray(T.T).reshape(16)


def from_colmajor(flat):
    return np.array(flat, dtype=np.float64).reshape(4, 4).T.copy()


class Context:
    """One GPU + one stream + scratch memory (o3d_ctx)."""

    def __init__(self, device=0):
        """device: one device id, or a sequence of ids for a context over several GPUs (o3d_ctx_create_multi: RegistrationICP
        then shards the source over them; the same id may repeat to rehearse the sharded path on one GPU)."""
        self._lib = load()
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*[int(d) for d in device])
            st = self._lib.o3d_ctx_create_multi(len(device), ids, C.byref(h))
            self.devices = [int(d) for d in device]
            device = self.devices[0]
        else:
            st = self._lib.o3d_ctx_create(int(device), C.byref(h))
            self.devices = [int(device)]
        if st != STATUS_OK:
            raise O3DError(st, "o3d_ctx_create(%s) failed: %s" % (self.devices, self._lib.o3d_last_error(None).decode()))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.o3d_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st != STATUS_OK:
            msg = self._lib.o3d_last_error(self._h).decode() or self._lib.o3d_status_string(st).decode()
            raise O3DError(st, msg)

    def set_stream(self, stream_handle):
        """Run on the given hipStream_t handle (0 / None = the HIP null stream, torch's default stream)."""
        self.check(self._lib.o3d_ctx_set_stream(self._h, C.c_void_p(stream_handle or None)))

    def use_own_stream(self):
        self.check(self._lib.o3d_ctx_use_own_stream(self._h))

    def debug_set(self, use_tile_search=True, collect_tile_stats=False):
        """use_tile_search=False: run the per-lane fp64 search + accumulation kernels instead of the default cooperative
        fp32-prefilter search."""
        self.check(self._lib.o3d_debug_set(self._h, 0 if use_tile_search else 1, int(collect_tile_stats)))

    def source_perm(self, n_src):
        out = np.empty(n_src, np.uint32)
        self.check(self._lib.o3d_debug_get_source_perm(self._h, C.c_void_p(out.ctypes.data)))
        return out

    def match_positions(self, n_src):
        out = np.empty(n_src, np.uint32)
        self.check(self._lib.o3d_debug_get_match_positions(self._h, C.c_void_p(out.ctypes.data)))
        return out

    def tile_stats(self, reset=True):
        out = (C.c_uint64 * 8)()
        self.check(self._lib.o3d_debug_tile_stats(self._h, out, int(reset)))
        keys = ["wave_iters", "item_trips", "runs", "exact_lanes", "items", "valid_lanes", "certified", "_7"]
        return dict(zip(keys, [int(v) for v in out]))

    def grid_info(self):
        out = (C.c_double * 10)()
        self.check(self._lib.o3d_debug_grid_info(self._h, out))
        keys = ["cell_side", "nx", "ny", "nz", "row_length", "margin", "cooperative_search", "points", "brick_side", "bricks"]
        return dict(zip(keys, [float(v) for v in out]))

    @property
    def handle(self):
        return self._h


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def ptr(a):
    """Raw address of a numpy array (host) or of anything with data_ptr() (torch device tensor); None -> NULL."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)

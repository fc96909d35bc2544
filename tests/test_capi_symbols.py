"""CPU suite: libo3d_amd.so loads, exports every symbol include/o3d_icp.h declares, and refuses to run without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "o3d_icp.h")).read()
    return sorted(set(re.findall(r"O3D_API[^;(]*?\b(o3d_\w+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_symbols()
    for must in ("o3d_registration_icp", "o3d_evaluate_registration", "o3d_voxel_down_sample", "o3d_estimate_normals",
                 "o3d_icp_setup", "o3d_icp_pass_to", "o3d_icp_solve", "o3d_icp_finish", "o3d_ctx_create"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from open3d_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(_lib.exported_symbols()) == declared_symbols(), "python binding table and header disagree"


def test_no_cpu_fallback():
    """Without a HIP device every entry point fails loudly (no silent CPU path)."""
    from open3d_amd import _lib
    lib = _lib.load()
    if lib.o3d_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.O3DError, match="no CPU fallback"):
        _lib.Context(0)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "open3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f


def test_expand_correspondences_host_helper():
    """o3d_expand_correspondences is host-only: dense corr -> (source, target) pairs in ascending source index."""
    import ctypes as C
    import numpy as np
    from open3d_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    corr = rng.integers(-1, 50, 1000).astype(np.int32)
    corr[rng.random(1000) < 0.3] = -1
    out = np.full((1000, 2), -7, np.int32)
    m = lib.o3d_expand_correspondences(corr.ctypes.data, len(corr), out.ctypes.data)
    idx = np.nonzero(corr >= 0)[0]
    assert m == len(idx)
    assert np.array_equal(out[:m, 0], idx) and np.array_equal(out[:m, 1], corr[idx])
    assert (out[m:] == -7).all()
    assert lib.o3d_expand_correspondences(corr.ctypes.data, 0, out.ctypes.data) == 0
    none = np.full(8, -1, np.int32)
    assert lib.o3d_expand_correspondences(none.ctypes.data, 8, out.ctypes.data) == 0


def test_adapter_open3d_types_branch_compiles():
    """The branch of the C++ adapter a maintainer of the reference compiles (O3D_AMD_USE_OPEN3D_TYPES: open3d::geometry::PointCloud
    and Eigen types instead of the layout-compatible stand-ins) parses and type-checks -- against tests/cpp/compile_stub, a
    types-only stub (the image has neither Eigen nor the reference's headers installed)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stamp = os.path.join(root, "tests", "cpp", "compile_check")
    if os.path.exists(stamp):
        os.remove(stamp)
    p = subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp"), "compile_check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]

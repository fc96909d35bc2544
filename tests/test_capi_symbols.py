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

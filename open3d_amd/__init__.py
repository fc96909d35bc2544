"""open3d_amd: MI355X-native (gfx950) ICP registration hot path behind the PolyCam/open3d API.

    from open3d_amd import geometry, pipelines
    result = pipelines.registration.RegistrationICP(source, target, 0.02, init, estimation, criteria)

The package is a thin host-side mirror of the reference's C++ interface over the C ABI of libo3d_amd.so
(include/o3d_icp.h).  No CPU fallback exists: without the built library and a HIP device, calls raise.
"""
from . import _lib, geometry, pipelines, synthetic  # noqa: F401

__all__ = ["geometry", "pipelines", "synthetic"]

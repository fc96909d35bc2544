import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library uses certificates (the k_icp_cert half of a pass) from 200 k source points on (capi.hip, icp_setup_impl); the GPU
# tests are mostly smaller than that and are there to exercise them, so the suite turns them on at every size (O3D_NO_CERT,
# where a test sets it, still wins).  The library's own policy -- no certificates for small clouds outside a gang -- is what the
# core parity modules run under as well: they use the `cert_policy` fixture below, which runs every test of the module twice
# (test_gpu_tail.py::test_certificate_size_policy checks which kernels the default picks).
os.environ.setdefault("O3D_CERT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def ctx():
    from open3d_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def make_scene(n_src, n_tgt, cfg_id=0, L=6.0, theta_deg=0.5, tau=(1.0, -0.7, 0.5), noise=0.1):
    """Height-field pair with enough relief at small N (L fixed, spacing = L / sqrt(n_tgt))."""
    from open3d_amd import synthetic
    return synthetic.make_pair(n_src, n_tgt, cfg_id=cfg_id, theta_deg=theta_deg, tau_in_spacings=tau,
                               noise_in_spacings=noise, L=L)


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(params=["certificates forced on", "library default"])
def cert_policy(request):
    """The switches are read at every set-up (o3d_icp_setup), so a test can change the policy between registrations."""
    saved = os.environ.get("O3D_CERT")
    if request.param == "library default":
        os.environ.pop("O3D_CERT", None)
    else:
        os.environ["O3D_CERT"] = "1"
    yield request.param
    if saved is None:
        os.environ.pop("O3D_CERT", None)
    else:
        os.environ["O3D_CERT"] = saved

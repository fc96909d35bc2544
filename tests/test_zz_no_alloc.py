"""o3d_compute_transformation_noalloc -- the one entry point that allocates nothing (no reference counterpart; for callers
under HIP graph capture or in serving loops that must not hipMalloc) -- against the existing o3d_compute_transformation /
o3d_compute_rmse on the same device-resident buffers.  Same kernels, same launch grid, same fixed-order reductions: the
bar is bit-exact.  Two calls in a row share the library's static scratch array; the second is smaller, so stale partial
rows of the first must not leak into it.  Collected last (test_zz_*): a failure here must not hide the other tests."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _pairs(corr):
    idx = np.nonzero(corr >= 0)[0].astype(np.int32)
    return np.ascontiguousarray(np.stack([idx, corr[idx]], axis=1))


def _both(ctx, torch, lib, est, with_scaling, src, tgt, nrm, pairs):
    """-> (20 doubles of the no-alloc call, 19 of the existing entry points), all inputs as device tensors."""
    from open3d_amd import _lib
    dev = torch.device("cuda", 0)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    d_nrm = torch.from_numpy(nrm).to(dev) if nrm is not None else None
    d_pairs = torch.from_numpy(pairs).to(dev)
    d_out = torch.full((20,), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    st = lib.o3d_compute_transformation_noalloc(0, C.c_void_p(stream or None), _lib.ptr(d_src), len(src), _lib.ptr(d_tgt), _lib.ptr(d_nrm),
                                                len(tgt), _lib.ptr(d_pairs), len(pairs), est, with_scaling, _lib.ptr(d_out))
    assert st == 0, lib.o3d_last_error(None).decode()
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    T = np.zeros(16)
    ctx.check(lib.o3d_compute_transformation(ctx.handle, _lib.ptr(d_src), len(src), _lib.ptr(d_tgt), _lib.ptr(d_nrm), len(tgt),
                                             _lib.ptr(d_pairs), len(pairs), est, with_scaling, _lib.MEM_DEVICE,
                                             T.ctypes.data_as(_lib.c_dp)))
    rm = C.c_double()
    ctx.check(lib.o3d_compute_rmse(ctx.handle, _lib.ptr(d_src), len(src), _lib.ptr(d_tgt), _lib.ptr(d_nrm), len(tgt), _lib.ptr(d_pairs),
                                   len(pairs), est, _lib.MEM_DEVICE, C.byref(rm)))
    return got, T, rm.value


@pytest.mark.parametrize("est,with_scaling", [(2, 0), (1, 0), (1, 1)])
def test_noalloc_matches_compute_transformation_twice(ctx, oracle, est, with_scaling):
    import torch
    lib = ctx._lib
    for n in (700_000, 9_000):  # the second call is smaller: 35 workgroups after 2048
        sc = make_scene(n, n + 1000, cfg_id=77 + n)
        s = sc["spacing"]
        nrm = oracle.estimate_normals(sc["target"], oracle.HYBRID, 30, 4 * s) if est == 2 else None
        corr = oracle.correspondences(sc["source"], sc["target"], 3 * s)[0]
        pairs = _pairs(corr)
        assert len(pairs) > 0.5 * n
        got, T, rmse = _both(ctx, torch, lib, est, with_scaling, sc["source"], sc["target"], nrm, pairs)
        assert np.array_equal(got[:16], T), "n=%d" % n
        assert got[17 if est == 2 else 16] == rmse
        assert got[18] == len(pairs) and got[19] == 0.0
        assert np.abs(T.reshape(4, 4).T - np.eye(4)).max() > 1e-4  # a real estimate, not the identity


def test_noalloc_edge_cases(ctx):
    import torch
    from open3d_amd import _lib
    lib = ctx._lib
    dev = torch.device("cuda", 0)
    src = torch.rand((100, 3), dtype=torch.float64, device=dev)
    tgt = torch.rand((50, 3), dtype=torch.float64, device=dev)
    out = torch.full((20,), float("nan"), dtype=torch.float64, device=dev)
    # no correspondences: identity, like TransformationEstimation.cpp:51-52
    empty = torch.zeros((1, 2), dtype=torch.int32, device=dev)
    assert lib.o3d_compute_transformation_noalloc(0, None, _lib.ptr(src), 100, _lib.ptr(tgt), None, 50, _lib.ptr(empty), 0, 1, 0,
                                                  _lib.ptr(out)) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.array_equal(o[:16].reshape(4, 4), np.eye(4)) and o[18] == 0 and o[19] == 0
    # an index outside the clouds is counted, not dereferenced
    bad = torch.tensor([[0, 1], [5, 50], [100, 3], [7, 7]], dtype=torch.int32, device=dev)
    assert lib.o3d_compute_transformation_noalloc(0, None, _lib.ptr(src), 100, _lib.ptr(tgt), None, 50, _lib.ptr(bad), 4, 1, 0,
                                                  _lib.ptr(out)) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert o[19] == 2 and o[18] == 2
    # sizes beyond the stated maximum, bad arguments: refused with a message, nothing launched
    assert lib.o3d_compute_transformation_noalloc(0, None, _lib.ptr(src), 100, _lib.ptr(tgt), None, 50, _lib.ptr(bad), 2**31, 1, 0,
                                                  _lib.ptr(out)) == 9
    assert b"maximum" in lib.o3d_last_error(None)
    assert lib.o3d_compute_transformation_noalloc(0, None, _lib.ptr(src), 100, _lib.ptr(tgt), None, 50, _lib.ptr(bad), 4, 3, 0,
                                                  _lib.ptr(out)) == 9
    assert lib.o3d_compute_transformation_noalloc(0, None, _lib.ptr(src), 100, _lib.ptr(tgt), None, 50, _lib.ptr(bad), 4, 1, 0, None) == 5
    assert lib.o3d_compute_transformation_noalloc(99, None, _lib.ptr(src), 100, _lib.ptr(tgt), None, 50, _lib.ptr(bad), 4, 1, 0,
                                                  _lib.ptr(out)) == 6

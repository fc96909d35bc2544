"""Synthetic registration scenes.

The reference ships no data, so every benchmark and parity case runs on clouds
drawn from one analytic height field with a splitmix64 counter stream:
identical host buffers feed the CPU baseline and the GPU path.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, start, count):
    """count outputs of the splitmix64 stream with state seed + (start+1+i)*GOLDEN."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + 1 + count, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, start, count):
    """u = (z >> 11) * 2^-53 in [0, 1)."""
    return (splitmix64(seed, start, count) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def height(x, y):
    return 0.30 * np.sin(1.3 * x) * np.cos(0.9 * y) + 0.12 * np.sin(3.1 * x + 0.7) + 0.08 * np.cos(2.3 * y - 0.4)


def surface_normals(x, y):
    """Analytic unit normals of the height field (used only as a sanity reference)."""
    dzdx = 0.30 * 1.3 * np.cos(1.3 * x) * np.cos(0.9 * y) + 0.12 * 3.1 * np.cos(3.1 * x + 0.7)
    dzdy = -0.30 * 0.9 * np.sin(1.3 * x) * np.sin(0.9 * y) - 0.08 * 2.3 * np.sin(2.3 * y - 0.4)
    n = np.stack([-dzdx, -dzdy, np.ones_like(dzdx)], axis=1)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def rigid_transform(theta_deg, tau):
    """Rotation by theta about (1,2,3)/sqrt(14), then translation tau; 4x4."""
    a = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    th = np.deg2rad(theta_deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(tau, dtype=np.float64)
    return T


def extent_for(n_points, spacing=None):
    """Side L of the square domain; mean spacing s = L / sqrt(N).  10 M -> L = 20 m (s ~ 6.3 mm)."""
    if spacing is None:
        spacing = 20.0 / np.sqrt(1.0e7)
    return spacing * np.sqrt(float(n_points))


def make_pair(n_src, n_tgt, cfg_id=0, theta_deg=0.05, tau_in_spacings=(1.0, -0.7, 0.5), noise_in_spacings=0.1, L=None):
    """Return dict(source, target, T_true, spacing, L).

    target: n_tgt points on the surface.  source: an independent draw of n_src
    points from the same surface (not a subset) + N(0, (noise*s)^2) Box-Muller
    noise, moved by the inverse of T_true, so that RegistrationICP(source,
    target) should recover ~T_true.
    """
    seed = int(0x9E3779B97F4A7C15) ^ int(cfg_id)
    if L is None:
        L = extent_for(n_tgt)
    s = L / np.sqrt(float(n_tgt))
    u = uniform01(seed, 0, 2 * n_tgt)
    tx, ty = L * u[0::2], L * u[1::2]
    target = np.stack([tx, ty, height(tx, ty)], axis=1)
    off = 2 * n_tgt
    u = uniform01(seed, off, 2 * n_src)
    sx, sy = L * u[0::2], L * u[1::2]
    src = np.stack([sx, sy, height(sx, sy)], axis=1)
    off += 2 * n_src
    if noise_in_spacings > 0:
        m = 3 * n_src
        m2 = m + (m & 1)
        u = uniform01(seed, off, m2)
        u1 = 1.0 - u[0::2]  # (0, 1]
        u2 = u[1::2]
        rad = np.sqrt(-2.0 * np.log(u1))
        g = np.empty(m2)
        g[0::2] = rad * np.cos(2 * np.pi * u2)
        g[1::2] = rad * np.sin(2 * np.pi * u2)
        src = src + (noise_in_spacings * s) * g[:m].reshape(n_src, 3)
    T_true = rigid_transform(theta_deg, np.asarray(tau_in_spacings, dtype=np.float64) * s)
    Tinv = np.linalg.inv(T_true)
    source = src @ Tinv[:3, :3].T + Tinv[:3, 3]
    return {
        "source": np.ascontiguousarray(source),
        "target": np.ascontiguousarray(target),
        "T_true": T_true,
        "spacing": s,
        "L": L,
    }


def make_pair_device(ctx, n_src, n_tgt, cfg_id=0, theta_deg=0.05, tau_in_spacings=(1.0, -0.7, 0.5), noise_in_spacings=0.1, L=None,
                     want_normals=True, device=None):
    """make_pair generated on the GPU (o3d_synthetic_pair) into torch tensors: dict(source, target, normals, T_true, spacing,
    L).  The same clouds up to the last digits of the device's sin / cos / log."""
    import torch
    from . import _lib
    seed = (int(0x9E3779B97F4A7C15) ^ int(cfg_id)) & 0xFFFFFFFFFFFFFFFF
    if L is None:
        L = extent_for(n_tgt)
    s = L / np.sqrt(float(n_tgt))
    T_true = rigid_transform(theta_deg, np.asarray(tau_in_spacings, dtype=np.float64) * s)
    Tinv = np.linalg.inv(T_true)
    dev = device if device is not None else torch.device("cuda", ctx.device)
    src = torch.empty((n_src, 3), dtype=torch.float64, device=dev)
    tgt = torch.empty((n_tgt, 3), dtype=torch.float64, device=dev)
    nrm = torch.empty((n_tgt, 3), dtype=torch.float64, device=dev) if want_normals else None
    ctx.check(ctx._lib.o3d_synthetic_pair(ctx.handle, n_src, n_tgt, seed, float(L), float(noise_in_spacings * s),
                                          _lib.colmajor(Tinv).ctypes.data_as(_lib.c_dp), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(nrm)))
    return {"source": src, "target": tgt, "normals": nrm, "T_true": T_true, "spacing": s, "L": L}

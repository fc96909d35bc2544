#!/usr/bin/env python3
"""Check the GPU's source processing order against a numpy Hilbert order (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/tmp")
import numpy as np
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = synthetic.make_pair(n, n, cfg_id=3, L=20 * np.sqrt(n / 1e7))
s = sc["spacing"]
ctx = _lib.Context(0)
nrm = synthetic.surface_normals(sc["target"][:, 0], sc["target"][:, 1])
sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], nrm, 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0, 0, 8))
perm = ctx.source_perm(n)
assert np.array_equal(np.sort(perm), np.arange(n)), "not a permutation"
h = 2 * s * (1 + 1e-5)
gmin = sc["target"].min(0)
nc = np.floor((sc["target"].max(0) - gmin) / h).astype(np.int64) + 1
c = np.clip(np.floor((sc["source"] - gmin) * (1.0 / h)).astype(np.int64), 0, nc - 1)
cc = c[perm]
G = 8
g = cc[: n // G * G].reshape(-1, G, 3)
ext = g.max(1) - g.min(1) + 3
print("nc", nc, "ext mean", ext.mean(0), "frac ext<=10", (ext.max(1) <= 10).mean())
d = np.abs(np.diff(cc, axis=0)).max(1)
print("consecutive cell step: mean %.2f, frac<=1 %.3f, frac>4 %.4f" % (d.mean(), (d <= 1).mean(), (d > 4).mean()))
tc = np.clip(np.floor((sc["target"] - gmin) * (1.0 / h)).astype(np.int64), 0, nc - 1)
nxp = nc[0] + 2
key_t = (tc[:, 2] * nc[1] + tc[:, 1]) * nxp + tc[:, 0] + 1
cs = np.concatenate([[0], np.cumsum(np.bincount(key_t, minlength=nxp * nc[1] * nc[2] + 4))])
lo, hi = g.min(1), g.max(1)
by0 = np.maximum(lo[:, 1] - 1, 0); by1 = np.minimum(hi[:, 1] + 1, nc[1] - 1)
bz0 = np.maximum(lo[:, 2] - 1, 0); bz1 = np.minimum(hi[:, 2] + 1, nc[2] - 1)
nby = by1 - by0 + 1; nbz = bz1 - bz0 + 1
tot = np.zeros(len(g), np.int64)
for dz in range(12):
    for dy in range(12):
        m = (dy < nby) & (dz < nbz)
        row = ((bz0 + dz) * nc[1] + (by0 + dy)) * nxp
        row = np.where(m, row, 0)
        tot += np.where(m, cs[row + hi[:, 0] + 3] - cs[row + lo[:, 0]], 0)
ok = ((hi[:, 0] - lo[:, 0] + 3) <= 10) & (nby <= 10) & (nbz <= 10) & (nby * nbz <= 32)
print("ok %.3f  tot mean %.1f  frac tot<=96 %.3f  tiled %.3f" % (ok.mean(), tot[ok].mean(), (tot[ok] <= 96).mean(), (ok & (tot <= 96)).mean()))
ctx.debug_set(True, True)
ctx.tile_stats(reset=True)
sess.enqueue_step()
print(ctx.tile_stats())
for p in range(4):
    sess.enqueue_step()
    st = ctx.tile_stats()
    print("pass", p + 1, "tiled %.3f cand/tile %.1f exact %.3f scanlen %.1f" % (st["tiled"] / st["wave_iters"], st["candidates"] / max(st["tiled"], 1), st["exact_lanes"] / st["valid_lanes"], st["rows"] * 8 / st["wave_iters"]))

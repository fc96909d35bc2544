"""Size of the ICP update per iteration (max |T_k - T_(k-1)|) and change of the rmse, 2 M x 2 M points of the config-3 scene,
point-to-plane, 40 iterations: quadratic convergence to ~1e-16 within four iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg
ctx = _lib.Context(0)
n = 2_000_000
sc = synthetic.make_pair_device(ctx, n, n, cfg_id=3)
s = sc["spacing"]
sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], sc["normals"], 2 * s, np.eye(4), reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria(0, 0, 40))
for _ in range(41):
    sess.enqueue_step()
torch.cuda.synchronize()
lg = np.array(sess.log())
T = lg[:, 4:20]
print([float("%.2e" % np.abs(T[k] - T[k - 1]).max()) for k in range(1, len(T))])
print([float("%.3e" % abs(lg[k, 1] - lg[k - 1, 1])) for k in range(1, len(T))])

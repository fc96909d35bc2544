#!/usr/bin/env python3
"""A few whole RegistrationICP calls with the default criteria on device-resident 10 M x 10 M clouds, for a kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 scripts/call_trace.py [points] [radius in spacings]
then scripts/trace_timeline.py <kernel_trace.csv> prints the last call kernel by kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open3d_amd import _lib, synthetic
from open3d_amd.pipelines import registration as reg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
sc = synthetic.make_pair_device(ctx, n, n, cfg_id=3)
s = sc["spacing"]
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
est, crit = reg.TransformationEstimationPointToPlane(), reg.ICPConvergenceCriteria()
corr = torch.empty(n, dtype=torch.int32, device=dev)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sess = reg.IcpSession(ctx).setup(sc["source"], sc["target"], sc["normals"], rs * s, np.eye(4), est, crit)
    sess.run()
    res = sess.finish(corr_out=corr)
    torch.cuda.synchronize()
    print("call %d: %.3f ms, %d iterations, fitness %.4f" % (rep, 1e3 * (time.perf_counter() - t0), res.iterations_, res.fitness_), flush=True)
    time.sleep(0.05)

#!/bin/bash
# bench.py with one rank, then with two ranks sharing the box's one GPU (--dist-backend gloo: RCCL refuses two ranks on one
# device, so the 32 sums per rank go through the host): the N > 1 flow of bench.py / sharded.py on the real kernels.  The two
# lines must agree in n_corr, fitness, rmse and transformation to the last bits; timings under gloo mean nothing.
# usage: scripts/two_rank_rehearsal.sh <tag>
set -e
TAG=$1
A="--points 2000000 --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-hard-scene"
timeout -k 10 200 python3 bench.py --gpus 1 $A > gpurun_out/rehearsal_${TAG}_1.json 2> gpurun_out/rehearsal_${TAG}_1.err
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo $A > gpurun_out/rehearsal_${TAG}_2.json 2> gpurun_out/rehearsal_${TAG}_2.err
python3 - <<PY
import json
a = json.loads([l for l in open("gpurun_out/rehearsal_${TAG}_1.json") if l.startswith("{")][-1])
b = json.loads([l for l in open("gpurun_out/rehearsal_${TAG}_2.json") if l.startswith("{")][-1])
ca, cb = a["config"], b["config"]
print("n_corr", ca["n_corr"], cb["n_corr"], "fitness", ca["fitness"], cb["fitness"])
print("rmse diff", abs(ca["inlier_rmse"] - cb["inlier_rmse"]), "max |T| diff", max(abs(x - y) for x, y in zip(ca["transformation_rowmajor"], cb["transformation_rowmajor"])))
assert ca["n_corr"] == cb["n_corr"] and ca["fitness"] == cb["fitness"]
assert abs(ca["inlier_rmse"] - cb["inlier_rmse"]) < 1e-15 and max(abs(x - y) for x, y in zip(ca["transformation_rowmajor"], cb["transformation_rowmajor"])) < 1e-15
print("REHEARSAL_OK")
PY

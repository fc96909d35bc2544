#!/bin/bash
# A/B of prebuilt library variants (variants/lib*.so) on the headline bench; every step under its own time limit, stops at the first failure
set -e
for v in "$@"; do
  cp variants/lib$v.so open3d_amd/libo3d_amd.so
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_${v}_$rep.json 2> gpurun_out/ab_err.log
    python - "$v" "$rep" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/ab_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(d["value"], 1), round(d["roofline"]["search_kernel_ms"], 4), flush=True)
PY
  done
done

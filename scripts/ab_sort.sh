#!/bin/bash
# A/B of library variants on the set-up times reported by bench.py (grid build, source sort)
set -e
for v in "$@"; do
  cp variants/lib$v.so open3d_amd/libo3d_amd.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/abs_$v.json 2> gpurun_out/abs_err.log
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abs_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "grid %.3f ms  source %.3f ms" % (d["config"]["grid_build_ms"], d["config"]["source_sort_ms"]), flush=True)
PY
done

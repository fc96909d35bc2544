#!/bin/bash
# sweep of the search kernel's workgroup count on the headline bench (tuning aid)
set -e
for g in "$@"; do
  O3D_SEARCH_GRID=$g timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/gs_$g.json 2> gpurun_out/gs_err.log
  python - "$g" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/gs_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 1), round(d["roofline"]["search_kernel_ms"], 4), flush=True)
PY
done

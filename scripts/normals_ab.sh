#!/bin/bash
# A/B of prebuilt library variants (variants/lib<name>.so) with scripts/normals_time.py at 10 M points; every step under its
# own time limit.
set -e
for v in "$@"; do
  cp variants/lib$v.so open3d_amd/libo3d_amd.so
  echo "== $v" >> gpurun_out/normals_ab.jsonl
  timeout -k 10 120 python3 scripts/normals_time.py 10000000 >> gpurun_out/normals_ab.jsonl 2>> gpurun_out/normals_ab.err
  tail -1 gpurun_out/normals_ab.jsonl | cut -c1-120
done

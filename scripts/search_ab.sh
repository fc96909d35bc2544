#!/bin/bash
# A/B of prebuilt library variants (variants/lib<name>.so) with scripts/search_ab.py; every step under its own time limit.
# A variant written name@bits runs with O3D_SRC_KEY_BITS=bits.
set -e
for v in "$@"; do
  lib=${v%@*}
  unset O3D_SRC_KEY_BITS O3D_FORCE_BRICKS
  if [[ "$v" == *@bricks ]]; then export O3D_FORCE_BRICKS=1; elif [[ "$v" == *@* ]]; then export O3D_SRC_KEY_BITS=${v#*@}; fi
  cp variants/lib$lib.so open3d_amd/libo3d_amd.so
  timeout -k 10 120 python3 scripts/search_ab.py $v >> gpurun_out/search_ab.jsonl 2>> gpurun_out/search_ab.err
  tail -1 gpurun_out/search_ab.jsonl
done

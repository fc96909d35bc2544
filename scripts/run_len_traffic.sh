#!/bin/bash
# Fabric bytes of k_icp_pass per run length of wave_range (library variants built with -DO3D_RUN_BLOCKS=n as
# variants/librun<n>.so): FETCH_SIZE and WRITE_SIZE in a counter pass each over a short bench run.
# (The macro was an experiment of round 3 and is gone: the product keeps the x-th eighth of the window per XCD, which is what
# run length 512 was.  Kept as the record of how profiles/r03_l_run_length_traffic.txt was made.)
# usage: scripts/run_len_traffic.sh 8 64 512
set -e
export TMPDIR=/tmp
R=$PWD
for n in "$@"; do
  cp variants/librun$n.so open3d_amd/libo3d_amd.so
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_run${n}_$c -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-hard-scene > $R/gpurun_out/pmc_run${n}_$c.log 2>&1
  done
  cd $R
  for c in FETCH_SIZE WRITE_SIZE; do for k in k_icp_pass k_icp_search_coop; do echo -n "run$n $k "; python3 scripts/pmc_summary.py gpurun_out/pmc_run${n}_$c $k 4; done; done >> gpurun_out/run_len_traffic.txt
  find gpurun_out -name "*counter_collection.csv" -path "*pmc_run${n}_*" -delete
done
cat gpurun_out/run_len_traffic.txt

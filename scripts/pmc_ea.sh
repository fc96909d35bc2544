#!/bin/bash
# Exact fabric-side read / write bytes of k_icp_pass from the L2's request counters by size (FETCH_SIZE tallies a 128-byte request
# at 64): one counter pass each over a short bench run.  usage: scripts/pmc_ea.sh <tag>
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmc_${TAG}_$1 -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-hard-scene > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1
}
cd /tmp
run eard "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
run eawr "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
cd $R
for k in k_icp_pass k_icp_search_coop; do
  for d in eard eawr; do python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_$d $k 4; done > gpurun_out/pmc_${TAG}_${k}_ea_summary.txt 2>&1
done
find gpurun_out -name "*counter_collection.csv" -path "*pmc_${TAG}*" -delete
cat gpurun_out/pmc_${TAG}_k_icp_pass_ea_summary.txt

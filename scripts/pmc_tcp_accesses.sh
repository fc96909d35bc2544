#!/bin/bash
# L1 accesses and requests to L2 per launch of the pass kernels, one counter pass over a short bench run.  usage: <tag>
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_acc -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-hard-scene > $R/gpurun_out/pmc_${TAG}_acc.log 2>&1
cd $R
for k in k_icp_pass k_icp_search_coop; do echo $k; python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_acc $k 4; done > gpurun_out/pmc_${TAG}_acc_summary.txt 2>&1
find gpurun_out -name "*counter_collection.csv" -path "*pmc_${TAG}_acc*" -delete
cat gpurun_out/pmc_${TAG}_acc_summary.txt

#!/bin/bash
# Address translation in k_icp_pass / k_icp_search_coop: first-level translation cache requests, hits and misses per launch and how
# long the second level is busy, one counter pass over a short bench run.  usage: scripts/pmc_tlb.sh <tag>
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${TAG}_tlb -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-hard-scene > $R/gpurun_out/pmc_${TAG}_tlb.log 2>&1
cd $R
for k in k_icp_pass k_icp_search_coop; do echo $k; python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_tlb $k 4; done > gpurun_out/pmc_${TAG}_tlb_summary.txt 2>&1
find gpurun_out -name "*counter_collection.csv" -path "*pmc_${TAG}_tlb*" -delete
cat gpurun_out/pmc_${TAG}_tlb_summary.txt

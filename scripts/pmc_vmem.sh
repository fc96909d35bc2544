#!/bin/bash
# Is the vector memory path (address unit, L1, data return) the limit of k_icp_pass?  One counter pass each over a short bench
# run.  usage: scripts/pmc_vmem.sh <tag>
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmc_${TAG}_$1 -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-hard-scene > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1
}
cd /tmp
run ta "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"
run tcp "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
run ta2 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
run tcp2 "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"
run tcp3 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"
cd $R
for k in k_icp_pass k_icp_search_coop; do
  for d in ta tcp ta2 tcp2 tcp3; do python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_$d $k 4; done > gpurun_out/pmc_${TAG}_${k}_vmem_summary.txt 2>&1
done
find gpurun_out -name "*counter_collection.csv" -path "*pmc_${TAG}*" -delete
cat gpurun_out/pmc_${TAG}_k_icp_pass_vmem_summary.txt

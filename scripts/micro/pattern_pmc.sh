#!/bin/bash
# L1 / address-unit counters of the memory-pattern micro-benchmark with the real match positions, to set beside those of k_icp_pass
# (scripts/pmc_vmem.sh).  usage: scripts/micro/pattern_pmc.sh
set -e
export TMPDIR=/tmp
R=$PWD
timeout -k 10 200 python3 scripts/micro/real_positions.py /tmp/pos.bin > gpurun_out/pattern_pmc_positions.txt 2>&1
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/pmc_pattern_tcp -- $R/scripts/micro/cert_pattern 10000000 64 /tmp/pos.bin > $R/gpurun_out/pmc_pattern_tcp.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TOTAL_READ_sum --output-format csv -d $R/gpurun_out/pmc_pattern_ta -- $R/scripts/micro/cert_pattern 10000000 64 /tmp/pos.bin > $R/gpurun_out/pmc_pattern_ta.log 2>&1
cd $R
for m in 0 1 2 3 5 6; do echo "mode $m"; for d in tcp ta; do python3 scripts/pmc_summary.py gpurun_out/pmc_pattern_$d "k_pattern<$m>" 4; done; done > gpurun_out/pmc_pattern_summary.txt 2>&1
find gpurun_out -name "*counter_collection.csv" -path "*pmc_pattern*" -delete
cat gpurun_out/pmc_pattern_summary.txt

#!/bin/bash
# usage: scripts/normals_pmc_ta.sh <variant>  -- texture-addresser / L1 counters over k_estimate_normals_hist (is the vector memory
# pipeline the limit?)
set -e
export TMPDIR=/tmp NRM_ONLY=hybrid_4s_30
R=$PWD
V=$1
cp variants/lib$V.so open3d_amd/libo3d_amd.so
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmcn_${V}_$1 -- python3 scripts/normals_time.py 10000000 > $R/gpurun_out/pmcn_${V}_$1.log 2>&1
}
run ta "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"
run tcp "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
run ta2 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum"
for d in ta tcp ta2; do python3 scripts/pmc_summary.py gpurun_out/pmcn_${V}_$d k_estimate_normals_hist 3; done > gpurun_out/pmcn_${V}_ta_summary.txt 2>&1
find gpurun_out -name "*counter_collection.csv" -path "*pmcn_${V}*" -delete
echo PMC_DONE $V

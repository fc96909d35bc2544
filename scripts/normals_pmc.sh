#!/bin/bash
# usage: scripts/normals_pmc.sh <variant>  -- counter passes (one process each) over k_estimate_normals_hist, Hybrid(4 s, 30), 10 M
set -e
export TMPDIR=/tmp NRM_ONLY=hybrid_4s_30
R=$PWD
V=$1
cp variants/lib$V.so open3d_amd/libo3d_amd.so
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmcn_${V}_$1 -- python3 scripts/normals_time.py 10000000 > $R/gpurun_out/pmcn_${V}_$1.log 2>&1
}
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
run sq2 "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
run sq3 "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
for d in sq sq2 sq3; do python3 scripts/pmc_summary.py gpurun_out/pmcn_${V}_$d k_estimate_normals_hist 3; done > gpurun_out/pmcn_${V}_summary.txt 2>&1
find gpurun_out -name "*counter_collection.csv" -path "*pmcn_${V}*" -delete
echo PMC_DONE $V

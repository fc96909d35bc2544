#!/bin/bash
# usage: scripts/prof_normals_pmc.sh <tag>  -- PMC passes over k_estimate_normals<2> (Hybrid(4 s, 30), 10 M points): bench.py runs
# EstimateNormals three times before its loop; each pass is its own process
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmcn_${TAG}_$1 -- python3 bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > $R/gpurun_out/pmcn_${TAG}_$1.log 2>&1
}
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_HIT_sum TCC_MISS_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq2 "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
run sq3 "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
for d in sq tcp fetch write sq2 sq3; do python3 scripts/pmc_summary.py gpurun_out/pmcn_${TAG}_$d ${KERNEL:-k_estimate_normals} 3; done > gpurun_out/pmcn_${TAG}_summary.txt 2>&1
echo PMC_DONE

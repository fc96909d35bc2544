#!/bin/bash
# usage: scripts/prof_pmc.sh <tag> [kernel substring]  -- PMC passes over a short bench run (each pass is its own process)
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1
KERNEL=${2:-k_icp_search_coop}
run() { # name, counters
  EXTRA=${BENCH_ARGS:-}
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmc_${TAG}_$1 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline $EXTRA > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1
}
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_HIT_sum TCC_MISS_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq2 "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
run sq3 "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
for d in sq tcp fetch write sq2 sq3; do python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_$d "$KERNEL" 4; done > gpurun_out/pmc_${TAG}_summary.txt 2>&1
echo PMC_DONE

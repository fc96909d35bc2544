#!/bin/bash
# Round-3 profile of the headline bench: (1) rocprofv3 --kernel-trace --stats of the bench command, (2) PMC passes (one process
# each: counters are not split over passes) of a shorter run of the same command, summarised per kernel.  Every step under a
# time limit of its own.  usage: scripts/prof_r03.sh <tag>
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -o b -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}_bench.err
rm -f $R/gpurun_out/prof_${TAG}_trace/*kernel_trace.csv $R/gpurun_out/prof_${TAG}_trace/*/*kernel_trace.csv
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmc_${TAG}_$1 -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-hard-scene > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1
}
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
run sq2 "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
run sq3 "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
cd $R
for k in k_icp_pass k_icp_search_coop; do
  for d in fetch write sq sq2 sq3 tcp; do python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_$d $k 4; done > gpurun_out/pmc_${TAG}_${k}_summary.txt 2>&1
done
find gpurun_out -name "*counter_collection.csv" -path "*pmc_${TAG}*" -delete
echo PROF_DONE

#!/bin/bash
# PMC passes for k_estimate_normals over the bench's one EstimateNormals call (each pass is its own process)
set -e
export TMPDIR=/tmp
R=$PWD
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmcn_$1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcn_$1.log 2>&1
}
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
echo PMC_DONE

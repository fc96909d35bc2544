#!/bin/bash
# Kernel trace of the one-process-per-GPU exchange path (pass -> RCCL all-gather -> solve) with ONE rank, at shard sizes of an
# 8- and a 4-GPU run of the 10 M headline.  bench.py is started directly (no launcher): the rendezvous comes from the environment.
set -e
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
for n in 1250000 2500000; do
  rm -rf $R/gpurun_out/xtrace_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/xtrace_$n -- python3 bench.py --points $n --steps 40 --warmup 3 \
      --blocks 2 --force-exchange --no-cpu-baseline --no-hard-scene > $R/gpurun_out/xtrace_$n.json 2> $R/gpurun_out/xtrace_$n.err
  echo "== $n source points per rank" >> $R/gpurun_out/xtrace_summary.txt
  python3 scripts/trace_step_breakdown.py $(ls $R/gpurun_out/xtrace_$n/*/*kernel_trace.csv | head -1) 30 >> $R/gpurun_out/xtrace_summary.txt
  python3 - $R/gpurun_out/xtrace_$n.json >> $R/gpurun_out/xtrace_summary.txt <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("  bench (under the profiler): fresh-call value %.0f it/s, steady state %.1f us per step" % (j["value"], 1e3 * j["config"]["steady_state"]["ms_per_step"]))
PY
done
cat $R/gpurun_out/xtrace_summary.txt

#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py's timed region alone (--timed-only: the rehearsal and the 5 timed registrations):
# the average k_icp_pass duration of the stats is then comparable with roofline.kernel_ms of the same run.  usage: <tag>
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_timed -o b -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --timed-only > $R/gpurun_out/prof_${TAG}_timed_bench.json 2> $R/gpurun_out/prof_${TAG}_timed_bench.err
rm -f $R/gpurun_out/prof_${TAG}_timed/*kernel_trace.csv $R/gpurun_out/prof_${TAG}_timed/*/*kernel_trace.csv
cd $R
python3 scripts/kstats.py gpurun_out/prof_${TAG}_timed k_icp_pass k_icp_search_coop
python3 -c "import json;d=json.load(open('gpurun_out/prof_${TAG}_timed_bench.json'));print('bench: value',d['value'],'roofline.kernel_ms',d['roofline']['kernel_ms'],'launches',d['roofline']['launches_timed'],'search pass ms',d['roofline']['search_pass']['kernel_ms'])"

# usage: c5sweep.sh  -- config 5 over hardware queues x per-pair search grid x certificates (tuning experiment)
for q in 8 16; do for g in 64 128; do for c in 0 1; do
  if [ $c = 1 ]; then export O3D_CERT=1; else unset O3D_CERT; fi
  GPU_MAX_HW_QUEUES=$q O3D_SEARCH_GRID=$g timeout -k 10 200 python scripts/config5_full.py 4096 8 16 2>>gpurun_out/c5.err | python -c "import sys,json; d=json.load(sys.stdin); print('queues $q grid $g cert $c:', round(d['pairs_per_s_workers_8']), round(d['pairs_per_s_workers_16']), d['sample_single_calls_identical'])" || exit 1
done; done; done

# usage: c5sweep.sh  -- config 5 over gang size x gang threads (tuning experiment)
for gsz in 8 16 32; do
  O3D_BATCH_GANG=$gsz timeout -k 10 200 python scripts/config5_full.py 4096 1 2 4 8 2>>gpurun_out/c5.err | python -c "import sys,json; d=json.load(sys.stdin); print('gang $gsz:', [round(d['pairs_per_s_workers_%d' % w]) for w in (1,2,4,8)], d['sample_single_calls_identical'], d['max_abs_T_minus_truth'], d['status_nonzero'])" || exit 1
done

set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/prof_pmc.sh r02y k_icp_pass > gpurun_out/prof_r02y.log 2>&1
for d in sq tcp fetch write sq2 sq3; do python3 scripts/pmc_summary.py gpurun_out/pmc_r02y_$d k_icp_search_coop 4; done > gpurun_out/pmc_r02y_search_summary.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02y_trace -o b -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/prof_r02y_bench.json 2>gpurun_out/prof_r02y_bench.err
rm -f gpurun_out/prof_r02y_trace/b_kernel_trace.csv
find gpurun_out -name "*counter_collection.csv" -path "*pmc_r02y*" -delete
echo DONE

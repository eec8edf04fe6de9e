# rocprofv3 kernel trace of the reference's training loop body with the occupancy estimator (run ON the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tocc -- python3 $R/bench.py --workload train-occ --steps 40 --warmup 8 --no-cpu-baseline > $R/gpurun_out/prof_tocc.log 2>&1
grep -h ms_per_step $R/gpurun_out/prof_tocc.log | cut -c1-400
cp $R/gpurun_out/prof_tocc/*/*kernel_stats.csv $R/gpurun_out/prof_tocc_kernel_stats.csv
head -25 $R/gpurun_out/prof_tocc_kernel_stats.csv | cut -c1-200

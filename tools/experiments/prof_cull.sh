# rocprofv3 kernel traces of the opt-in bf16 cull: training loop body and evaluation frame (run ON the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_cull_t $R/gpurun_out/prof_cull_f
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cull_t -- python3 $R/bench.py --workload train-occ --steps 40 --warmup 8 --no-cpu-baseline --cull-precision bf16 > $R/gpurun_out/prof_cull_t.log 2>&1
cp $R/gpurun_out/prof_cull_t/*/*kernel_stats.csv $R/gpurun_out/prof_cull_train_kernel_stats.csv
head -8 $R/gpurun_out/prof_cull_train_kernel_stats.csv | cut -c1-150
python3 $R/tools/experiments/step_trace.py $R/gpurun_out/prof_cull_t/*/*kernel_trace.csv "k_render_occ<8, 1>" > $R/gpurun_out/prof_cull_step.txt
tail -1 $R/gpurun_out/prof_cull_step.txt
if [ "$1" != "train" ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cull_f -- python3 $R/tools/exp_cull.py > $R/gpurun_out/prof_cull_f.log 2>&1
cp $R/gpurun_out/prof_cull_f/*/*kernel_stats.csv $R/gpurun_out/prof_cull_frame_kernel_stats.csv
head -12 $R/gpurun_out/prof_cull_frame_kernel_stats.csv | cut -c1-150
fi

#!/bin/bash
# Kernel times of the training step for several library builds (run ON the GPU box): rocprofv3 kernel trace of
# tools/bench_train.py --no-opt per library.  usage: bash tools/ab_wgrad.sh <lib.so>...
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for L in "$@"; do
  N=$(basename $L .so)
  export FSN_LIB_PATH=$R/$L
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abw_$N -- python3 $R/tools/bench_train.py --no-opt --steps 10 --warmup 2 > $R/gpurun_out/abw_$N.log 2>&1 || exit $?
  echo "== $N: $(grep -h ms_per_step $R/gpurun_out/abw_$N.log | cut -c1-120)"
  grep -h "k_wgrad<\|k_train_\|k_heads_wgrad" $R/gpurun_out/abw_$N/*/*kernel_stats.csv | awk -F'","' '{printf "   %-60s calls %s avg %.1f us\n", $1, $2, $4/1000}'
done

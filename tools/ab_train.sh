#!/bin/bash
# Interleaved A/B of the training step (tools/bench_train.py, BASELINE config 4 shape) across library builds, run ON
# the GPU box.  usage: bash tools/ab_train.sh [rounds] <lib.so>...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for r in $(seq $N); do
  for L in "$@"; do
    FSN_LIB_PATH=$R/$L timeout -k 10 200 python3 $R/tools/bench_train.py --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'], 3), 'ms/step')" || exit $?
  done
done

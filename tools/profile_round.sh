#!/bin/bash
# Round profile of the headline bench (run ON the GPU box): rocprofv3 kernel-trace stats + the PMC passes.
# usage: bash tools/profile_round.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/, gpurun_out/pmc_<tag>/
TAG=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1 || exit $?
echo "kernel trace done"
bash $R/tools/run_pmc.sh $TAG "$@"

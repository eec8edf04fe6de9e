#!/bin/bash
# Collect rocprofv3 PMC passes for the bench workload (run ON the GPU box via gpurun).
# usage: bash tools/run_pmc.sh <tag> [bench args...]
set -e
TAG=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1
  echo "pass $i done"
done

#!/bin/bash
# Run GPU steps one after another on the gpurun box; a step that times out (or is killed) ends the call - no
# further GPU step is started after a hang.  usage: bash tools/gpu_steps.sh "<cmd1>" "<cmd2>" ...
# each step: timeout -k 10 ${STEP_TIMEOUT:-420}; output appended to gpurun_out/steps.log
mkdir -p gpurun_out
i=0
for c in "$@"; do
  i=$((i+1))
  echo "=== step $i: $c" | tee -a gpurun_out/steps.log
  timeout -k 10 ${STEP_TIMEOUT:-420} bash -c "$c" >> gpurun_out/steps.log 2>&1
  rc=$?
  echo "=== step $i rc=$rc" | tee -a gpurun_out/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $i timed out: stopping" | tee -a gpurun_out/steps.log; exit $rc; fi
done
exit 0

#!/bin/bash
# Round-3 evidence, run ON the GPU box (gpurun): kernel traces + PMC passes of the headline render workload, the
# occupancy workload and the training step.  Outputs under gpurun_out/ (summaries are copied into profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
set -x
bash $R/tools/profile_round.sh r03 || exit $?
mkdir -p $R/gpurun_out/prof_r03occ $R/gpurun_out/prof_r03train
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03occ -- python3 $R/bench.py --workload occgrid --steps 3 --warmup 1 > $R/gpurun_out/prof_r03occ.log 2>&1 || exit $?
bash $R/tools/run_pmc.sh r03occ --workload occgrid || exit $?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03train -- python3 $R/bench.py --workload train --steps 20 --warmup 3 > $R/gpurun_out/prof_r03train.log 2>&1 || exit $?
echo profiles done
bash $R/tools/run_pmc.sh r03train --workload train --steps 3 --warmup 1 || exit $?
echo train pmc done

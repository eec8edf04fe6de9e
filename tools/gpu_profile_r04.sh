#!/bin/bash
# Round-4 evidence, run ON the GPU box (gpurun): kernel traces + PMC passes of the headline render workload (fp16x3 on the
# scaled network), its bf16 form, the occupancy workload and the training step.  Outputs under gpurun_out/ (summaries are
# copied into profiles/ afterwards by tools/summarize_pmc.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
bash $R/tools/profile_round.sh r04 || exit $?
echo "== headline done"
bash $R/tools/run_pmc.sh r04bf16 --precision bf16 || exit $?
echo "== bf16 pmc done"
mkdir -p $R/gpurun_out/prof_r04occ $R/gpurun_out/prof_r04train
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04occ -- python3 $R/bench.py --workload occgrid --steps 3 --warmup 1 > $R/gpurun_out/prof_r04occ.log 2>&1 || exit $?
bash $R/tools/run_pmc.sh r04occ --workload occgrid || exit $?
echo "== occ done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04train -- python3 $R/bench.py --workload train --steps 20 --warmup 3 > $R/gpurun_out/prof_r04train.log 2>&1 || exit $?
bash $R/tools/run_pmc.sh r04train --workload train --steps 3 --warmup 1 || exit $?
echo "== train done"

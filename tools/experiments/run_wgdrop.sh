set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/exp_wgrad_drop.py > gpurun_out/wgdrop_base.log 2>&1
for v in big1 big2 big3 all3; do
  FSN_LIB_PATH=$GRAFT_REPO_ROOT/ab/wg_$v.so timeout -k 10 400 python3 tools/exp_wgrad_drop.py > gpurun_out/wgdrop_$v.log 2>&1
done
grep -h "worst" gpurun_out/wgdrop_*.log

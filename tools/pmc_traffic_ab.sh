#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_render_fused for several library builds (run ON the GPU box; one PMC pass each).
# usage: bash tools/pmc_traffic_ab.sh <lib.so>...
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for L in "$@"; do
  N=$(basename $L .so)
  for CNT in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    FSN_LIB_PATH=$R/$L timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmcab_$N/$(echo $CNT | cut -d' ' -f1) -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 || exit $?
  done
  python3 - <<PY
import csv, glob
tot = {}
for p in glob.glob("$R/gpurun_out/pmcab_$N/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "k_render_fused" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("$N", "fetch GB %.2f" % (2 * tot.get("FETCH_SIZE", 0) * 1024 / 1e9), "write GB %.2f" % (tot.get("WRITE_SIZE", 0) * 1024 / 1e9))
PY
done

#!/bin/bash
# one rocprofv3 PMC pass: instruction-cache and instruction-fetch counters of the fused frame
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_icache
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/p1.log 2>&1
python3 - <<PY
import csv, glob
tot={}
for p in glob.glob("$OUT/p1/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "k_render_fused" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]=tot.get(r["Counter_Name"],0)+float(r["Counter_Value"])
for k,v in sorted(tot.items()): print(f"{k:32s} {v:.4e}")
PY

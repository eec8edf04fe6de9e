#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes collected by tools/run_pmc.sh into profiles/<tag>_pmc_summary.json
(+ a copy of the per-kernel rows).  usage: python tools/summarize_pmc.py <tag> [kernel-substring] [precision]
The summary is stamped with the precision mode and a hash of the kernel sources (bench.csrc_sha): bench.py quotes
`traffic` from it only while both still match the code being benchmarked."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kern = sys.argv[2] if len(sys.argv) > 2 else "k_render_fused"
prec = sys.argv[3] if len(sys.argv) > 3 else "fp16x3"
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
tot, disp, dur = {}, {}, []
for p in sorted(glob.glob(os.path.join(src, "p*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(p)):
        if kern not in r["Kernel_Name"]:
            continue
        c = r["Counter_Name"]
        tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"])
        disp.setdefault(c, set()).add((p, r["Dispatch_Id"]))
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
# launches of the kernel per pass (bench.py --steps 1 --warmup 0: one for the render workload, two for occgrid, whose
# rank 0 renders the frame once more for the sample counts): counters are averaged per launch
n = max([len(v) for v in disp.values()] + [1])
tot = {c: v * n / len(disp[c]) for c, v in tot.items()}
import bench  # noqa: E402  (csrc_sha only; no GPU use)
out = {"kernel": kern, "precision": prec, "csrc_sha": bench.csrc_sha(), "launches_per_pass": n, "counters_per_launch": {k: v / n for k, v in sorted(tot.items())}}
c = out["counters_per_launch"]
if dur:
    out["kernel_seconds_profiled"] = sum(dur) / len(dur)
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE/WRITE_SIZE are KiB of L2<->fabric requests; on gfx950 FETCH_SIZE reads
    # exactly half of a wide (16 B/lane) coalesced stream -> doubled; WRITE_SIZE is exact.  Infinity-Cache hits are
    # included (the weight stream is served from L2 / Infinity Cache, so this is an upper bound on HBM bytes).
    out["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    out["fetch_bytes_corrected"] = 2.0 * c["FETCH_SIZE"] * 1024.0
    out["write_bytes"] = c["WRITE_SIZE"] * 1024.0
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    out["kernel_cycles"] = cyc
    out["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
    if dur:
        out["clock_ghz"] = cyc / out["kernel_seconds_profiled"] / 1e9
if "TCC_HIT_sum" in c:
    out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if "SQ_LDS_IDX_ACTIVE" in c and "kernel_cycles" in out:
    out["lds_array_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (256.0 * out["kernel_cycles"])  # one LDS per CU
if "SQ_INSTS_MFMA" in c and "SQ_INSTS_VALU" in c:
    out["valu_non_mfma_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
    out["lds_insts_per_mfma"] = c.get("SQ_INSTS_LDS", 0.0) / c["SQ_INSTS_MFMA"]
    out["salu_insts_per_mfma"] = c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_MFMA"]
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
    if k in c and "SQ_WAVE_CYCLES" in c:
        out[k.lower() + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))

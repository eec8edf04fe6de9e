#!/usr/bin/env python3
"""Kernel sequence of one steady-state training step out of a rocprofv3 kernel trace (gap before each launch, duration):
usage: step_trace.py <kernel_trace.csv> <marker kernel substring> [step index from the end, default 3]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a, b = idx[-k], idx[-k + 1]
prev, busy = None, 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    busy += e - s
    print(f"gap {gap:7.1f} us  dur {(e - s) / 1e3:7.1f} us  {re.sub('at::native::', '', r['Kernel_Name'])[:100]}")
t = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"step {t / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, {b - a} launches")

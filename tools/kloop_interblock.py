#!/usr/bin/env python3
"""Instruction mix BETWEEN the hand-scheduled GEMM-pair blocks of a kernel, from a hipcc -S listing: what a wave
executes per output pair beside the block's own stream (epilogue, bookkeeping).
usage: hipcc ... -S render.hip -o render.s ; python tools/kloop_interblock.py render.s [mangled-kernel-substring] [mfmas-per-block]"""
import collections
import sys

path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_render_fusedILi8ELi2ELb1"
want = int(sys.argv[3]) if len(sys.argv) > 3 else 48
s = open(path).read()
i = s.index("\n_ZN3fsn14" + kern)
j = s.index(".Lfunc_end", i)
segs, cur, inblk, buf, prev = [], None, False, [], 0
for l in s[i:j].split("\n"):
    t = l.strip()
    if t.startswith(";;#ASMSTART"):
        inblk, buf = True, []
    elif t.startswith(";;#ASMEND"):
        inblk = False
        nm = sum(1 for x in buf if x.startswith("v_mfma"))
        if nm >= 12:  # a k-loop block
            if cur is not None:
                segs.append((prev, cur))
            prev, cur = nm, []
        elif cur is not None:
            cur.extend(buf)
    elif inblk:
        if t and not t.startswith(";") and not t.endswith(":"):
            buf.append(t.split()[0])
    elif cur is not None and t and not t.startswith(";") and not t.startswith(".") and not t.endswith(":"):
        cur.append(t.split()[0])
c, n, hist = collections.Counter(), 0, collections.Counter()
for pm, seg in segs:
    if pm == want and len(seg) < 400:
        n += 1
        hist[len(seg)] += 1
        c.update(seg)
print(f"{kern}: {n} gaps after {want}-MFMA blocks; lengths {sorted(hist.items())}")
tot = sum(c.values())
print(f"average {tot / max(n, 1):.1f} instructions per gap")
cls = collections.Counter()
for op, k in c.items():
    cls["VALU" if op.startswith("v_") else "SALU/branch" if op.startswith("s_") else "LDS" if op.startswith("ds_") else
        "VMEM" if op.startswith(("global_", "scratch_", "buffer_")) else "other"] += k
print({k: round(v / max(n, 1), 1) for k, v in cls.items()})
for op, k in c.most_common(24):
    print(f"  {op:28s} {k / n:6.2f}")

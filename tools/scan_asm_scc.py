#!/usr/bin/env python3
"""Static check of the device code for one inline-asm trap: a generated k-loop block contains `s_cmp_*` (its phase
openings), so the asm statement must declare the `scc` clobber - otherwise hipcc may keep a condition of its own alive
across the block (found in round 2: the standalone MLP kernel selected its weight-stream pointers by the block's
comparison and faulted).  Compiles a translation unit to assembly and reports every `s_cselect / s_cbranch_scc / s_addc /
s_subb` that consumes SCC after such a block without an SCC definition in between.
usage: python tools/scan_asm_scc.py [render.hip mlp.hip ...]   (default: render.hip)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fs-nerf_amd", "csrc")
SCC_DEF = re.compile(r"^(s_cmp|s_cmpk|s_add_|s_addc|s_sub_|s_subb|s_and_|s_or_|s_xor_|s_lshl|s_lshr|s_ashr|s_bfe|s_andn2|"
                     r"s_orn2|s_not|s_abs|s_min|s_max|s_addk|s_mulk|s_bitcmp|s_absdiff|s_wqm|s_quadmask|s_nand|s_nor|s_xnor|"
                     r"s_bcnt|s_ff|s_flbit)")
SCC_USE = re.compile(r"^(s_cselect|s_cbranch_scc|s_addc|s_subb|s_cmov)")


def scan(path):
    kern, in_asm, has_cmp, pending = None, False, False, False
    res = {}
    for n, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, pending = m.group(1), False
            res[kern] = [0, []]
            continue
        if kern is None:
            continue
        t = line.strip()
        if "ASMSTART" in t:
            in_asm, has_cmp = True, False
            continue
        if "ASMEND" in t:
            in_asm = False
            if has_cmp:
                res[kern][0] += 1
                pending = True
            continue
        if in_asm:
            has_cmp |= t.startswith("s_cmp")
            continue
        if not line.startswith("\t") or t.startswith((".", ";")):
            continue
        if pending:
            if SCC_USE.match(t):
                res[kern][1].append((n, t))
                pending = False
            elif SCC_DEF.match(t):
                pending = False
    return res


def main():
    bad = 0
    for src in (sys.argv[1:] or ["render.hip"]):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "a.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S",
                            os.path.join(CSRC, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
            for k, (nblocks, uses) in scan(out).items():
                if nblocks:
                    print(f"{src}: {k}: {nblocks} asm blocks with s_cmp, {len(uses)} SCC reads behind them")
                    for n, t in uses[:3]:
                        print(f"    line {n}: {t}")
                    bad += len(uses)
    print("OK" if bad == 0 else f"{bad} suspicious SCC reads")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

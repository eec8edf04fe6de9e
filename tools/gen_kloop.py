#!/usr/bin/env python3
"""Generates fs-nerf_amd/csrc/kloop_gen.hpp: the hand-scheduled gfx950 instruction streams of the hidden-layer
k-loop of the NeRF MLP kernels (mlp_dev.hpp), as inline-asm string macros.

Why generated text and not C++: hipcc places every `ds_read_b128` of an A operand (weights in the LDS ring) directly
in front of the MFMA that consumes it - the read's whole LDS latency is then exposed to the wave and only the partner
wave of the SIMD covers it (profiles/r01: 60 % matrix-pipe busy, 49 % of wave cycles parked in s_waitcnt).  The
streams below keep D units (D = 2: two 1-KiB high-part + two 1-KiB low-part reads) in flight ahead of the MFMAs in a
three-set register rotation, wait with counted lgkmcnt, open the next weight phase (counted vmcnt + s_barrier + the
two LDS-DMA loads of this wave) at its fixed place inside the stream, and spend exactly
    per unit (x3):  2 ds_read_b128 + 1 s_waitcnt + 3 v_mfma_f32_16x16x32
instructions.  One macro = one output pair (32 features x 16 samples of a wave) of one GEMM.

A block FSN_KLOOP_<MODE>_<NU>_<OFF>(MFMA) covers NU consecutive units of the weight stream starting OFF units into
a phase (UPP units per phase).  Operands (named):
  b<k>h / b<k>l   B operand (activations, high / low 16-bit parts) of k-step k, k < NU/2          [v, 128 bit, in]
  z0, z1          initial accumulators = bias rows of the two 16-feature tiles                     [v, 128 bit, in]
  c0, c1          accumulators of the two tiles                                                     [v, 128 bit, out&]
  s0h..s2l        three A-operand register sets; on entry set 0 / 1 hold units 0 / 1 (landed),
                  on exit set (NU % 3) / ((NU+1) % 3) hold units NU / NU+1 (landed)               [v, 128 bit, in/out]
  a0..a<P-1>      LDS byte address (ring slot base + 16*lane) of the phases the block touches      [v, 32 bit, in]
  mv<e>, gb<e>    M0 value and 64-bit global base of the e-th phase opening inside the block       [s, in]
  voff            this lane's byte offset inside its wave's share of a phase (LDS-DMA vaddr)       [v, in]
  keep            scratch SGPR for M0                                                               [s, out&]

Hazards handled in the text (hipcc pads nothing inside asm): the last MFMA of each accumulator is followed by
>= 11 wait states before the block ends (XDL 8-pass result -> VALU read), all ds_reads have landed at the end
(s_waitcnt lgkmcnt(0)), an A set is only overwritten by a ds_read issued after the last MFMA reading it.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UPP = {"X3": 8, "X2": 8}      # units per 16-KiB phase (a unit = 1 KiB high + 1 KiB low parts)
UB = 2048                      # bytes per unit in the x3 stream layout
ABL = os.environ.get("FSN_KLOOP_ABL", "")
LEAD = 2                       # a phase is opened LEAD units before the previous one ends (= prefetch distance D)
D = 2


def block(mode, nu, off):
    """instruction list of one block"""
    upp = upp_of(mode)
    rpu = 2 if mode == "X3" else 1  # ds_reads per unit (x2: the weights' low parts are not read)
    ins = []
    ev = 0
    first = {0: True, 1: True}

    def read(u):
        gu = off + u
        ph, o = gu // upp, (gu % upp) * UB
        s = u % 3
        ins.append(f"ds_read_b128 %[s{s}h], %[a{ph}] offset:{o}")
        if mode == "X3":
            ins.append(f"ds_read_b128 %[s{s}l], %[a{ph}] offset:{o + 1024}")

    for u in range(nu):
        if (off + u + LEAD) % upp == 0:
            # open the next phase: its loads (all but the 2 youngest of this wave) have landed, every wave is past
            # the phase whose slot is restaged; then issue this wave's two 1-KiB LDS-DMA loads of the phase after
            # (FSN_KLOOP_ABL: timing experiments that drop parts of this - results are then garbage)
            if "nobarrier" not in ABL:
                ins += ["s_waitcnt vmcnt(2)", "s_barrier"]
            if "nodma" not in ABL:
                ins += ["s_mov_b32 %[keep], m0", f"s_mov_b32 m0, %[mv{ev}]", "s_nop 0",
                        f"global_load_lds_dwordx4 %[voff], %[gb{ev}]",
                        f"global_load_lds_dwordx4 %[voff], %[gb{ev}] offset:1024", "s_mov_b32 m0, %[keep]"]
            ev += 1
        read(u + D)
        ins.append(f"s_waitcnt lgkmcnt({D * rpu})")
        k, t, s = u // 2, u % 2, u % 3
        src = f"%[z{t}]" if first[t] else f"%[c{t}]"
        first[t] = False
        ins.append(f"MFMA %[c{t}], %[s{s}h], %[b{k}h], {src}")
        if mode == "X3":
            ins.append(f"MFMA %[c{t}], %[s{s}l], %[b{k}h], %[c{t}]")
        ins.append(f"MFMA %[c{t}], %[s{s}h], %[b{k}l], %[c{t}]")
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 3"]
    return ins, ev


def upp_of(mode):
    return UPP[mode]


def n_phases(mode, nu, off):
    return (off + nu + D - 1) // upp_of(mode) + 1


def emit(mode, nu, off):
    ins, ev = block(mode, nu, off)
    name = f"FSN_KLOOP_{mode}_{nu}_{off}"
    lines = [f"// {name}: {nu} units from unit {off} of a phase; {ev} phase opening(s); "
             f"{n_phases(mode, nu, off)} phase address(es); {len(ins)} instructions",
             f"#define {name}_EVENTS {ev}", f"#define {name}_PHASES {n_phases(mode, nu, off)}",
             f"#define {name}(MFMA) \\"]
    for i in ins:
        if i.startswith("MFMA "):
            lines.append(f'  MFMA "{i[4:]}\\n\\t" \\')
        else:
            lines.append(f'  "{i}\\n\\t" \\')
    lines[-1] = lines[-1][:-2]
    return "\n".join(lines) + "\n"


def main():
    out = ["// kloop_gen.hpp - GENERATED by tools/gen_kloop.py (python tools/gen_kloop.py); do not edit.",
           "// Hand-scheduled hidden-layer k-loop instruction streams for gfx950; see the generator's docstring.",
           "#pragma once", ""]
    for mode in ("X3", "X2"):
        out.append(emit(mode, 16, 0))   # 256 -> 256 layers: 8 k-steps, pair = 2 whole phases
        for off in (0, 4):              # the skip layer [h, x_in]: 10 k-steps, pairs start 0 or 4 units into a phase
            out.append(emit(mode, 20, off))
        for off in (0, 2, 4, 6):        # branch [feat, dir_enc]: 9 k-steps
            out.append(emit(mode, 18, off))
        for off in (0, 4):              # first layer: 2 k-steps
            out.append(emit(mode, 4, off))
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fs-nerf_amd", "csrc", "kloop_gen.hpp")
    open(dst, "w").write("\n".join(out))
    print("wrote", dst)


if __name__ == "__main__":
    main()

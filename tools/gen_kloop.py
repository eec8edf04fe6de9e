#!/usr/bin/env python3
"""Generates fs-nerf_amd/csrc/kloop_gen.hpp: the hand-scheduled gfx950 instruction streams of the GEMM pairs of the
NeRF MLP kernels (mlp_dev.hpp), as inline-asm string macros.

Why generated text and not C++: (1) hipcc places every `ds_read_b128` of an A operand (weights in the LDS ring)
directly in front of the MFMA that consumes it, so the read's whole LDS latency is exposed to the wave; (2) it runs
each pair's epilogue (ReLU, fp16 high/low split, range guard) and the bias loads as a separate block after the
pair's MFMAs, and because the weight stream's workgroup barrier keeps the two waves of a SIMD in step, both leave
the matrix pipe idle at the same time (stamps, tools/stamp_report.py: 200-360 cycles of epilogue + ~300 of
bookkeeping per 1536-cycle pair).  The streams below
  * keep D = 2 units of A operands in flight ahead of the MFMAs in a three-set register rotation, with counted
    lgkmcnt waits;
  * carry the weight-phase openings (counted vmcnt + s_barrier + this wave's two LDS-DMA loads) at their fixed places;
  * load the bias of the NEXT pair into the other of two pinned accumulator sets behind the MFMAs;
  * keep the two CORRECTION products of the split (al.wh, ah.wl) in their own accumulator set C = v[232:239]: the low
    parts of the fp16 modes are stored scaled by 2^11 (mlp_layout.hpp, kLoScaleF16) so that they stay normal fp16
    numbers for activations down to ~1e-4; the pair's epilogue folds 2^-11 * C into the main sums.  (The epilogue of
    a pair runs in C++ after its block; running it inside the next block's stream was measured slower)
One macro = one output pair (32 features x 16 samples of a wave) of one GEMM.

Naming: FSN_KLOOP_<MODE>_<NU>_<OFF>_N<PAR>(MFMA)
  MODE  X3 (a.w = ah.wh + [al.wh + ah.wl]), X2 (weights' low parts dropped: ah.wh + [ah.wl] in operand terms) or
        X3S (round 4, FSN_PREC_FP16X3U: the three products of X3 into ONE accumulator tile - the low parts are stored
        UNSCALED and the layer's activations are kept at 2^4..2^10 by per-layer power-of-two scales folded into the
        packed weights, mlp_layout.hpp; no correction set, no merge in the epilogue: C0 / C1 are not operands)
  NU    units in the pair (2 x k-steps), OFF = units between the start of the current phase and the pair's first unit
  PAR   0: the pair accumulates in set E = v[240:247], the next pair's bias goes to O = v[248:255];
        1: the other way round.  Tile 0 (features 0-15 of the pair) is the low half of a set, tile 1 the high half.
Operands (named):
  b<k>h / b<k>l   B operand (activations, high / low 16-bit parts) of k-step k                     [v, 128 bit, in]
  E0 E1 O0 O1     the four pinned accumulator tiles ("+{v[240:243]}" ...): on entry the current set holds the
                  pair's bias; on exit it holds this pair's main sums and the other set the next pair's bias
                                                                                                    [pinned, in/out]
  C0 C1           the pinned correction tiles ("=&{v[232:235]}", "=&{v[236:239]}"): written from zero (first MFMA of
                  each takes the inline constant 0 as its C operand), hold the correction sums on exit [pinned, out]
  s0h..s2l        three A-operand register sets; on entry set 0 / 1 hold units 0 / 1 (landed), on exit set
                  (NU % 3) / ((NU+1) % 3) hold units NU / NU+1 (landed)                             [v, 128 bit, in/out]
  a0..a3          LDS byte address (ring slot base + 16*lane) of the phases the block touches      [v, 32 bit, in]
  abn             LDS byte address of the next pair's bias row (this lane's 4 floats of tile 0; tile 1 at +64)
  mv<e>, gb<e>    M0 value and 64-bit global base of the e-th phase opening inside the block       [s, in]
  voff            this lane's byte offset inside its wave's share of a phase (LDS-DMA vaddr)       [v, in]
  keep            scratch SGPR for M0                                                               [s, out&]

Hazards handled in the text (hipcc pads nothing inside asm): >= 11 wait states after the last MFMA before the block
ends (XDL 8-pass result -> VALU read), all ds_reads landed at the end (s_waitcnt lgkmcnt(0)), an A set / the other
accumulator set is only overwritten by a ds_read issued after the last instruction reading it.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.environ.get("FSN_KLOOP_ABL", "")
WAIT2 = os.environ.get("FSN_KLOOP_WAIT2", "1") == "1"
UPP = 8        # units per 16-KiB phase (a unit = 1 KiB high + 1 KiB low parts)
UB = 2048      # bytes per unit in the x3 stream layout
D = int(os.environ.get("FSN_KLOOP_D", "2"))   # units of A operands in flight ahead of the MFMAs (D + 1 register sets)
LEAD = D       # a phase is opened LEAD units before the previous one ends
LOADS = 4     # LDS-DMA loads (1 KiB each) of a loader wave per phase: four loader waves x 4 KiB
LOOK = int(os.environ.get("FSN_KLOOP_LOOK", "2"))  # phases staged ahead of the one being opened (mlp_dev.hpp kLook =
# kNSlot - 2): a phase opening may leave (LOOK - 1) x LOADS loads of this wave in flight.  2 = the four-slot ring of the
# product; 3 = a five-slot ring (experiment, with -DFSN_NSLOT=5 -DFSN_RING_EXPERIMENT)
NSETS = D + 1  # A register sets
SETS = {0: (240, 248), 1: (248, 240)}  # parity -> (first register of the current set, of the other set)
CSET = 232     # first register of the correction set


def tile(base, t):
    return f"v[{base + 4 * t}:{base + 4 * t + 3}]"


def block(mode, nu, off, par):
    cur, other = SETS[par]
    ins = []
    ev = 0
    # next pair's bias into the other set (its previous contents were consumed by the C++ epilogue before this block)
    fill = [f"ds_read_b128 {tile(other, 0)}, %[abn]", f"ds_read_b128 {tile(other, 1)}, %[abn] offset:64"]
    state = {"n_mfma": 0, "n_lds": 0}
    # the two bias reads follow the MFMAs of unit FILL_AT (the middle of the interval between the block's two phase
    # openings), one behind each; the 12-MFMA pairs of the first layer take them from their third MFMA on
    fill_at = min(10, nu - 2) if nu > 4 else 0
    unit_last_read = {i: 0 for i in range(D)}    # number of ds_reads issued when the unit's last read was issued (0: landed on entry)
    c_started = [False, False]                   # correction tile t has been written in this block

    def emit(text, lds=False):
        ins.append(text)
        if lds:
            state["n_lds"] += 1

    def read(u):
        gu = off + u
        ph, o = gu // UPP, (gu % UPP) * UB
        s = u % NSETS
        emit(f"ds_read_b128 %[s{s}h], %[a{ph}] offset:{o}", True)
        if mode in ("X3", "X3S"):
            emit(f"ds_read_b128 %[s{s}l], %[a{ph}] offset:{o + 1024}", True)
        unit_last_read[u] = state["n_lds"]

    def mfma(dst, src_c, s, bop, may_fill):
        emit(f"MFMA {dst}, %[s{s}], %[b{bop}], {src_c}")
        state["n_mfma"] += 1
        if may_fill and fill and state["n_mfma"] >= 3:
            emit(fill.pop(0), True)

    def corr(t, s, bop, may_fill):
        if mode == "X3S" or "samechain" in ABL:   # the corrections into the main tile (one accumulator per tile)
            mfma(tile(cur, t), tile(cur, t), s, bop, may_fill)
            return
        c = tile(CSET, t)
        mfma(c, c if c_started[t] else "0", s, bop, may_fill)
        c_started[t] = True

    for u in range(nu):
        if (off + u + LEAD) % UPP == 0:
            # open the next phase: its loads (all but the youngest LOADS of this wave) have landed, every wave is past
            # the phase whose slot is restaged; then the loader waves issue their four 1-KiB LDS-DMA loads of the phase
            # after (FSN_KLOOP_ABL: timing experiments that drop parts of this - results are then garbage)
            if "nobarrier" not in ABL:
                ins += [f"s_waitcnt vmcnt({(LOOK - 1) * LOADS})", "s_barrier"]
            if "nodma" not in ABL:
                # only the loader waves (voff < 4 x 4096) issue LDS-DMA: the stall of issuing them then falls beside
                # the MFMAs of their SIMD partners instead of on both waves at once (the predicate is taken from
                # voff = 4096*wave + 16*lane: an "s" input operand may silently arrive in a VGPR, an "=&s" output cannot)
                ins += ["v_readfirstlane_b32 %[keep], %[voff]", f"s_cmp_ge_u32 %[keep], {LOADS * 1024 * 4}",
                        f"s_cbranch_scc1 .Lnoload{ev}_%=",
                        "s_mov_b32 %[keep], m0", f"s_mov_b32 m0, %[mv{ev}]", "s_nop 0"]
                ins += [f"global_load_lds_dwordx4 %[voff], %[gb{ev}]" + (f" offset:{1024 * i}" if i else "")
                        for i in range(LOADS)]
                ins += ["s_mov_b32 m0, %[keep]", f".Lnoload{ev}_%=:"]
            ev += 1
        read(u + D)
        # Every read of the units about to be used must have landed; LDS returns in order, so allow the reads issued
        # after their last one.  WAIT2 (default): one wait per k-step, in front of its first unit, covering both
        # of its units (one instruction less per k-step; the second unit's reads were issued a k-step ago).
        if WAIT2 and D == 2 and mode in ("X3", "X3S"):  # (x2: measured 1.5 % slower with the merged wait)
            if u % 2 == 0:
                last = max(unit_last_read[u], unit_last_read.get(u + 1, 0))
                ins.append(f"s_waitcnt lgkmcnt({state['n_lds'] - last})")
        else:
            ins.append(f"s_waitcnt lgkmcnt({state['n_lds'] - unit_last_read[u]})")
        k, t, s = u // 2, u % 2, u % NSETS
        may = u >= fill_at
        c = tile(cur, t)
        mfma(c, c, f"{s}h", f"{k}h", may)          # main sum: high x high
        if mode in ("X3", "X3S"):
            corr(t, f"{s}l", f"{k}h", may)         # corrections (X3: both scaled by 2^11 in the fp16 modes)
        corr(t, f"{s}h", f"{k}l", may)
    for text in fill:   # (never: every shape has room for the two bias reads)
        emit(text, True)
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 3"]
    return ins, ev


def n_phases(nu, off):
    return (off + nu + D - 1) // UPP + 1


def emit_macro(mode, nu, off, par):
    ins, ev = block(mode, nu, off, par)
    name = f"FSN_KLOOP_{mode}_{nu}_{off}_N{par}"
    lines = [f"// {name}: {nu} units from unit {off} of a phase; {ev} phase opening(s); {n_phases(nu, off)} phase "
             f"address(es); {len(ins)} instructions", f"#define {name}(MFMA) \\"]
    for i in ins:
        if i.startswith("MFMA "):
            lines.append(f'  MFMA "{i[4:]}\\n\\t" \\')
        else:
            lines.append(f'  "{i}\\n\\t" \\')
    lines[-1] = lines[-1][:-2]
    return "\n".join(lines) + "\n"


SHAPES = [(16, 0), (20, 0), (20, 4), (18, 0), (18, 2), (18, 4), (18, 6), (4, 0), (4, 4)]


def main():
    out = ["// kloop_gen.hpp - GENERATED by tools/gen_kloop.py (python tools/gen_kloop.py); do not edit.",
           "// Hand-scheduled GEMM-pair instruction streams for gfx950; see the generator's docstring.",
           "#pragma once", ""]
    for nu, off in SHAPES:
        _, ev = block("X3", nu, off, 0)
        out.append(f"#define FSN_KLOOP_{nu}_{off}_EVENTS {ev}")
        out.append(f"#define FSN_KLOOP_{nu}_{off}_PHASES {n_phases(nu, off)}")
    out.append(f"#define FSN_KLOOP_D {D}")
    out.append(f"#define FSN_KLOOP_NSETS {NSETS}")
    out.append("")
    for mode in ("X3", "X2", "X3S"):
        for nu, off in SHAPES:
            for par in (0, 1):
                out.append(emit_macro(mode, nu, off, par))
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fs-nerf_amd", "csrc", "kloop_gen.hpp")
    open(dst, "w").write("\n".join(out))
    print("wrote", dst)


if __name__ == "__main__":
    main()

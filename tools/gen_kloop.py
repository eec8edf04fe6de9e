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
  * interleave, one instruction per MFMA, the epilogue of the PREVIOUS pair (whose accumulators sit in the other of
    two pinned register sets) and the bias load of the NEXT pair (into that set once the epilogue has consumed it).
One macro = one output pair (32 features x 16 samples of a wave) of one GEMM.

Naming: FSN_KLOOP_<MODE>_<NU>_<OFF>_<EPI><PAR>(MFMA)
  MODE  X3 (a.w = ah.wh + al.wh + ah.wl) or X2 (weights' low parts dropped)
  NU    units in the pair (2 x k-steps), OFF = units between the start of the current phase and the pair's first unit
  EPI   N: no epilogue in the stream; R: ReLU + fp16 split of the previous pair; C: fp16 split only (signed values)
  PAR   0: the pair accumulates in set E = v[240:247], the previous pair's results / next bias use O = v[248:255];
        1: the other way round.  Tile 0 (features 0-15 of the pair) is the low half of a set, tile 1 the high half.
Operands (named):
  b<k>h / b<k>l   B operand (activations, high / low 16-bit parts) of k-step k                     [v, 128 bit, in]
  E0 E1 O0 O1     the four pinned accumulator tiles ("+{v[240:243]}" ...): on entry the current set holds the
                  pair's bias, the other set the previous pair's finished accumulators (EPI R / C); on exit the
                  current set holds this pair's results and the other set the next pair's bias      [pinned, in/out]
  s0h..s2l        three A-operand register sets; on entry set 0 / 1 hold units 0 / 1 (landed), on exit set
                  (NU % 3) / ((NU+1) % 3) hold units NU / NU+1 (landed)                             [v, 128 bit, in/out]
  a0..a3          LDS byte address (ring slot base + 16*lane) of the phases the block touches      [v, 32 bit, in]
  abn             LDS byte address of the next pair's bias row (this lane's 4 floats of tile 0; tile 1 at +64)
  mv<e>, gb<e>    M0 value and 64-bit global base of the e-th phase opening inside the block       [s, in]
  voff            this lane's byte offset inside its wave's share of a phase (LDS-DMA vaddr)       [v, in]
  keep            scratch SGPR for M0                                                               [s, out&]
  oh0..oh3, ol0..ol3   the previous pair's activations as fp16 high / low dwords (EPI R / C)       [v, 32 bit, out&]
  fmax            running packed max of |high part| bit patterns (range guard)                      [v, in/out]
  tmp             scratch VGPR (EPI C)                                                              [v, out&]

Hazards handled in the text (hipcc pads nothing inside asm): >= 11 wait states after the last MFMA before the block
ends (XDL 8-pass result -> VALU read), all ds_reads landed at the end (s_waitcnt lgkmcnt(0)), an A set / the other
accumulator set is only overwritten by a ds_read issued after the last instruction reading it, the epilogue starts
after the block's third MFMA (the previous block's last MFMA result is long complete).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.environ.get("FSN_KLOOP_ABL", "")
WAIT2 = os.environ.get("FSN_KLOOP_WAIT2", "1") == "1"
ILV = os.environ.get("FSN_KLOOP_ILV", "0") == "1"
SPREAD = os.environ.get("FSN_KLOOP_SPREAD", "0") == "1"
UPP = 8        # units per 16-KiB phase (a unit = 1 KiB high + 1 KiB low parts)
UB = 2048      # bytes per unit in the x3 stream layout
D = int(os.environ.get("FSN_KLOOP_D", "2"))   # units of A operands in flight ahead of the MFMAs (D + 1 register sets)
LEAD = D       # a phase is opened LEAD units before the previous one ends
LOADS = 4     # LDS-DMA loads (1 KiB each) of a loader wave per phase: four loader waves x 4 KiB
NSETS = 4 if (os.environ.get("FSN_KLOOP_ILV", "0") == "1") else int(os.environ.get("FSN_KLOOP_D", "2")) + 1  # A register sets
SETS = {0: (240, 248), 1: (248, 240)}  # parity -> (first register of the current set, of the other set)


def tile(base, t):
    return f"v[{base + 4 * t}:{base + 4 * t + 3}]"


def epilogue(kind, other):
    """instructions of the previous pair's epilogue on the register set starting at `other`"""
    if kind == "N":
        return []
    out = []
    regs = [f"v{other + j}" for j in range(8)]
    if kind == "R":
        out += [f"v_max_i32 {r}, 0, {r}" for r in regs]
    for i in range(4):
        a, b = regs[2 * i], regs[2 * i + 1]
        out.append(f"v_cvt_pk_f16_f32 %[oh{i}], {a}, {b}")
        out.append(f"v_fma_mixlo_f16 %[ol{i}], %[oh{i}], -1.0, {a} op_sel_hi:[1,0,0]")
        out.append(f"v_fma_mixhi_f16 %[ol{i}], %[oh{i}], -1.0, {b} op_sel:[1,0,0] op_sel_hi:[1,0,0]")
        if kind == "C":
            out.append(f"v_and_b32 %[tmp], 0x7fff7fff, %[oh{i}]")
            out.append("v_pk_max_u16 %[fmax], %[fmax], %[tmp]")
        else:
            out.append(f"v_pk_max_u16 %[fmax], %[fmax], %[oh{i}]")
    return out


def block(mode, nu, off, kind, par):
    cur, other = SETS[par]
    ins = []
    ev = 0
    fill = [(t, False) for t in epilogue(kind, other)]
    # next pair's bias into the other set, after the epilogue has consumed it
    fill += [(f"ds_read_b128 {tile(other, 0)}, %[abn]", True), (f"ds_read_b128 {tile(other, 1)}, %[abn] offset:64", True)]
    state = {"n_mfma": 0, "n_lds": 0, "dma_every": 3, "dma_wait": 0}
    pend = []   # (FSN_KLOOP_SPREAD) instruction groups of the current stage still to be issued
    # fillers per unit (three MFMAs): issue-slot budget of a SIMD shared by its two waves is ~24 slots per unit
    # pair, of which the MFMAs, A reads and waits of both waves take 18 (measured: three fillers per unit cost
    # ~20 cycles each); FSN_KLOOP_FILL overrides for experiments
    per_unit = int(os.environ.get("FSN_KLOOP_FILL", "0"))
    # FSN_KLOOP_FILL = 0 (default): the previous pair's epilogue runs as ONE burst in front of unit BURST_AT, i.e. in
    # the middle of the interval between the block's two phase openings.  Only waves 4..7 use the R / C variants:
    # their SIMD partners (waves 0..3) run their epilogue at the end of the pair, so the two bursts of a SIMD are
    # half a pair apart and each falls beside the partner's MFMAs.
    burst_at = min(10, nu - 2) if per_unit == 0 else -1
    if nu <= 4:
        per_unit, burst_at = 8, -1   # the 12-MFMA pairs of the first layer: interleaved
    unit_last_read = {i: 0 for i in range(D)}    # number of ds_reads issued when the unit's last read was issued (0: landed on entry)

    def emit(text, lds=False):
        ins.append(text)
        if lds:
            state["n_lds"] += 1

    def read(u):
        gu = off + u
        ph, o = gu // UPP, (gu % UPP) * UB
        s = u % NSETS
        emit(f"ds_read_b128 %[s{s}h], %[a{ph}] offset:{o}", True)
        if mode == "X3":
            emit(f"ds_read_b128 %[s{s}l], %[a{ph}] offset:{o + 1024}", True)
        unit_last_read[u] = state["n_lds"]

    def mfma(t, s, bop, n_fill):
        n_fill = n_fill if per_unit else 0
        c = tile(cur, t)
        emit(f"MFMA {c}, %[s{s}], %[b{bop}], {c}")
        state["n_mfma"] += 1
        if pend:
            state["dma_wait"] -= 1
            if state["dma_wait"] <= 0:
                for t in pend.pop(0):
                    ins.append(t)
                state["dma_wait"] = state["dma_every"]
        if state["n_mfma"] >= 3:
            for _ in range(n_fill):
                if fill:
                    text, is_lds = fill.pop(0)
                    emit(text, is_lds)

    npu = 3 if mode == "X3" else 2   # MFMAs per unit
    share = [per_unit // npu + (1 if i < per_unit % npu else 0) for i in range(npu)]  # fillers behind each MFMA

    for u in range(nu):
        if (off + u + LEAD) % UPP == 0:
            # open the next phase: its loads (all but the 2 youngest of this wave) have landed, every wave is past
            # the phase whose slot is restaged; then issue this wave's two 1-KiB LDS-DMA loads of the phase after
            # (FSN_KLOOP_ABL: timing experiments that drop parts of this - results are then garbage)
            if "nobarrier" not in ABL:
                ins += [f"s_waitcnt vmcnt({LOADS})", "s_barrier"]
            if "nodma" not in ABL and SPREAD:
                # FSN_KLOOP_SPREAD: the four loads of a loader wave are issued one at a time behind the following
                # MFMAs instead of back to back behind the barrier (tools/ubench/gen_issue.py: a burst of four costs
                # a w8 stream 11 points of matrix-pipe time, spread loads 3).  SCC holds "not a loader wave" and M0
                # the stage's LDS base until the last of them (nothing else in a block writes either).
                ins += ["v_readfirstlane_b32 %[keep], %[voff]", f"s_cmp_ge_u32 %[keep], {LOADS * 1024 * 4}",
                        "s_mov_b32 %[keep], m0", f"s_mov_b32 m0, %[mv{ev}]"]
                left = (nu - u) * (3 if mode == "X3" else 2)   # MFMAs from here to the end of the block
                state["dma_every"] = max(1, min(3 if mode == "X3" else 2, left // (LOADS + 1)))
                state["dma_wait"] = state["dma_every"]
                for i in range(LOADS):
                    pend.append([f"s_cbranch_scc1 .Lnl{ev}_{i}_%=",
                                 f"global_load_lds_dwordx4 %[voff], %[gb{ev}]" + (f" offset:{1024 * i}" if i else ""),
                                 f".Lnl{ev}_{i}_%=:"])
                pend.append(["s_mov_b32 m0, %[keep]"])
            elif "nodma" not in ABL:
                # only the loader waves (0..3, %[ldr] != 0) issue LDS-DMA: the stall of issuing them then falls
                # beside the MFMAs of their SIMD partners (waves 4..7) instead of on both waves at once
                # (the predicate is taken from voff = 4096*wave + 16*lane: an "s" input operand may silently arrive in
                # a VGPR, an "=&s" output cannot)
                ins += ["v_readfirstlane_b32 %[keep], %[voff]", f"s_cmp_ge_u32 %[keep], {LOADS * 1024 * 4}",
                        f"s_cbranch_scc1 .Lnoload{ev}_%=",
                        "s_mov_b32 %[keep], m0", f"s_mov_b32 m0, %[mv{ev}]", "s_nop 0"]
                ins += [f"global_load_lds_dwordx4 %[voff], %[gb{ev}]" + (f" offset:{1024 * i}" if i else "")
                        for i in range(LOADS)]
                ins += ["s_mov_b32 m0, %[keep]", f".Lnoload{ev}_%=:"]
            ev += 1
        if u == burst_at:
            bias = [f for f in fill if f[1]]
            for text, is_lds in [f for f in fill if not f[1]]:
                emit(text, False)
            fill[:] = bias           # (the two bias reads follow behind the next MFMAs)
            per_unit = 3
            share = [1, 1, 1] if mode == "X3" else [2, 1]
        if ILV and D == 2:
            if u % 2 == 0:   # both units of the next k-step, into the two sets the previous k-step has released
                read(u + 2)
                read(u + 3)
        else:
            read(u + D)
        # Every read of the units about to be used must have landed; LDS returns in order, so allow the reads issued
        # after their last one.  WAIT2 (default): one wait per k-step, in front of its first unit, covering both
        # of its units (one instruction less per k-step; the second unit's reads were issued a k-step ago).
        if ILV and D == 2:
            if u % 2 == 0:
                ins.append(f"s_waitcnt lgkmcnt({state['n_lds'] - max(unit_last_read[u], unit_last_read.get(u + 1, 0))})")
        elif WAIT2 and D == 2 and mode == "X3":  # (x2: measured 1.5 % slower with the merged wait)
            if u % 2 == 0:
                last = max(unit_last_read[u], unit_last_read.get(u + 1, 0))
                ins.append(f"s_waitcnt lgkmcnt({state['n_lds'] - last})")
        else:
            ins.append(f"s_waitcnt lgkmcnt({state['n_lds'] - unit_last_read[u]})")
        k, t, s = u // 2, u % 2, u % NSETS
        if ILV and D == 2:
            # the two units of a k-step interleaved MFMA by MFMA: consecutive MFMAs never share an accumulator
            if t == 0:
                continue
            s0 = (u - 1) % NSETS
            mfma(0, f"{s0}h", f"{k}h", share[0]); mfma(1, f"{s}h", f"{k}h", 0)
            if mode == "X3":
                mfma(0, f"{s0}l", f"{k}h", share[1]); mfma(1, f"{s}l", f"{k}h", 0)
            mfma(0, f"{s0}h", f"{k}l", share[-1]); mfma(1, f"{s}h", f"{k}l", 0)
            continue
        mfma(t, f"{s}h", f"{k}h", share[0])
        if mode == "X3":
            mfma(t, f"{s}l", f"{k}h", share[1])
        mfma(t, f"{s}h", f"{k}l", share[-1])
    for grp in pend:            # (a stage opened close to the end of the block)
        ins.extend(grp)
    pend.clear()
    for text, is_lds in fill:   # (short pairs: what did not fit beside the MFMAs)
        emit(text, is_lds)
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 3"]
    return ins, ev


def n_phases(nu, off):
    return (off + nu + D - 1) // UPP + 1


def emit_macro(mode, nu, off, kind, par):
    ins, ev = block(mode, nu, off, kind, par)
    name = f"FSN_KLOOP_{mode}_{nu}_{off}_{kind}{par}"
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
        _, ev = block("X3", nu, off, "N", 0)
        out.append(f"#define FSN_KLOOP_{nu}_{off}_EVENTS {ev}")
        out.append(f"#define FSN_KLOOP_{nu}_{off}_PHASES {n_phases(nu, off)}")
    out.append(f"#define FSN_KLOOP_D {D}")
    out.append(f"#define FSN_KLOOP_NSETS {NSETS}")
    out.append("")
    for mode in ("X3", "X2"):
        for nu, off in SHAPES:
            for kind in "NRC":
                for par in (0, 1):
                    out.append(emit_macro(mode, nu, off, kind, par))
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fs-nerf_amd", "csrc", "kloop_gen.hpp")
    open(dst, "w").write("\n".join(out))
    print("wrote", dst)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Micro-benchmark generator (design input for the MLP kernels' GEMM-pair streams; results: profiles/r02_ubench_issue.txt).

Question: what does the matrix pipe of a SIMD sustain when the wave(s) on it carry the whole stream themselves -
A operands (weights) from LDS two units ahead, VALU epilogue fillers between the MFMAs, one workgroup barrier and this
wave's four 1-KiB LDS-DMA loads per 16-KiB weight phase - for three shapes of the same arithmetic:
  w8  : 8 waves (two per SIMD), 16 samples per wave,  v_mfma_f32_16x16x32_f16, 3 MFMAs per unit  (the round-1/2 kernel)
  w4  : 4 waves (one per SIMD), 32 samples per wave as two groups of 16, 16x16x32, 6 MFMAs per unit sharing the A reads
  w4x : 4 waves (one per SIMD), 32 samples per wave, v_mfma_f32_32x32x16_f16, 3 MFMAs per unit
One unit = 2 ds_read_b128 (A high / low parts, 1 KiB each); one phase = 8 units = 768 matrix-pipe cycles per SIMD in
every shape.  Operands are random fp16 data (the clock the chip holds depends on it)."""
import sys

FILL = ["v_max_i32 v150, 0, v150", "v_cvt_pk_f16_f32 v151, v152, v153",
        "v_fma_mixlo_f16 v154, v151, -1.0, v152 op_sel_hi:[1,0,0]", "v_pk_max_u16 v155, v155, v151"]


def body(shape, fillers, barrier, dma, order="e3"):
    """fillers: VALU instructions per MFMA as a fraction n/d (n fillers every d MFMAs).
    order (w8 only; round 3: the fp16 modes keep the two correction products in their own accumulator C):
      e3    all three products of a unit into the main tile E_t (rounds 1-2; dependent chains of 3)
      e1c2  E_t <- hh; C_t <- lh, hl            (chains 1, 2: the round-3 stream)
      g24   units in same-tile pairs: E(u) E(u+1) C(u)x2 C(u+1)x2   (chains 2, 4)
      g222  same pairs: C(u)x2 E(u) E(u+1) C(u+1)x2                 (chains 2, 2, 2; A set of u free after 3 MFMAs)"""
    if shape == "w8" and order != "e3":
        return body_w8_corr(fillers, barrier, dma, order)
    fn, fd = fillers
    ins = []
    nm = 0
    fi = 0
    pending = []
    for u in range(8):
        if u == 6 and (barrier or dma):
            if dma:
                ins.append("s_waitcnt vmcnt(4)")
            if barrier:
                ins.append("s_barrier")
            if dma:
                loads = [f"global_load_lds_dwordx4 %[voff], %[gb] offset:{1024*i}" for i in range(4)]
                if dma == "burst":
                    ins += loads
                else:
                    pending = loads
        s, s2 = u % 3, (u + 2) % 3
        ins.append(f"ds_read_b128 v[{160+8*s2}:{163+8*s2}], %[la] offset:{2048*u}")
        ins.append(f"ds_read_b128 v[{164+8*s2}:{167+8*s2}], %[la] offset:{2048*u+1024}")
        if u % 2 == 0:
            ins.append("s_waitcnt lgkmcnt(2)")
        t = u % 2
        ah, al = f"v[{160+8*s}:{163+8*s}]", f"v[{164+8*s}:{167+8*s}]"
        k = u // 2
        if shape == "w4x":
            acc = "v[200:215]"
            # B: 32 samples x 16 k, high parts v[8k'..], low parts +4 (k' = unit index mod 16 -> 16 k-steps of 16)
            bh, bl = f"v[{8*u}:{8*u+3}]", f"v[{8*u+4}:{8*u+7}]"
            seq = [("v_mfma_f32_32x32x16_f16", acc, ah, bh), ("v_mfma_f32_32x32x16_f16", acc, al, bh),
                   ("v_mfma_f32_32x32x16_f16", acc, ah, bl)]
        else:
            seq = []
            groups = (0, 1) if shape == "w4" else (0,)
            for (a, part) in ((ah, 0), (al, 0), (ah, 4)):
                for g in groups:
                    acc = f"v[{200+4*(t+2*g)}:{203+4*(t+2*g)}]"
                    b0 = 8 * k + 64 * g + part
                    seq.append(("v_mfma_f32_16x16x32_f16", acc, a, f"v[{b0}:{b0+3}]"))
        for (op, acc, a, b) in seq:
            ins.append(f"{op} {acc}, {a}, {b}, {acc}")
            nm += 1
            if fn and nm % fd == 0:
                for _ in range(fn):
                    ins.append(FILL[fi % 4]); fi += 1
            if pending and nm % 3 == 0:
                ins.append(pending.pop(0))
    return ins, nm


def body_w8_corr(fillers, barrier, dma, order):
    fn, fd = fillers
    ins, nm, fi, pending = [], 0, 0, []
    state = {"nm": 0, "fi": 0}

    def mf(acc, a, b):
        ins.append(f"v_mfma_f32_16x16x32_f16 {acc}, {a}, {b}, {acc}")
        state["nm"] += 1
        if fn and state["nm"] % fd == 0:
            for _ in range(fn):
                ins.append(FILL[state["fi"] % 4]); state["fi"] += 1
        if pending and state["nm"] % 3 == 0:
            ins.append(pending.pop(0))

    E = lambda t: f"v[{200+4*t}:{203+4*t}]"
    C = lambda t: f"v[{208+4*t}:{211+4*t}]"
    A = lambda u: (f"v[{160+8*(u%3)}:{163+8*(u%3)}]", f"v[{164+8*(u%3)}:{167+8*(u%3)}]")
    B = lambda k: (f"v[{8*k}:{8*k+3}]", f"v[{8*k+4}:{8*k+7}]")

    def opening(u):
        if u == 6 and (barrier or dma):
            if dma:
                ins.append("s_waitcnt vmcnt(4)")
            if barrier:
                ins.append("s_barrier")
            if dma:
                loads = [f"global_load_lds_dwordx4 %[voff], %[gb] offset:{1024*i}" for i in range(4)]
                if dma == "burst":
                    ins.extend(loads)
                else:
                    pending.extend(loads)

    def read(u):  # unit u + 2 into the set unit u - 1 used
        s2 = (u + 2) % 3
        ins.append(f"ds_read_b128 v[{160+8*s2}:{163+8*s2}], %[la] offset:{2048*(u%8)}")
        ins.append(f"ds_read_b128 v[{164+8*s2}:{167+8*s2}], %[la] offset:{2048*(u%8)+1024}")

    if order == "f8":
        # the two correction products on the 2x-rate fp8 path: per 4 k-steps (8 units) 8 f16 MFMAs
        # (high x high) + 2 tiles x 2 v_mfma_f32_16x16x128_f8f6f4 (K = 128 = the same 4 k-steps).  The table's TF/s
        # counts the stream as the 24 f16-MFMA equivalents it replaces.
        for u in range(8):
            opening(u)
            s2 = (u + 2) % 3
            ins.append(f"ds_read_b128 v[{160+4*s2}:{163+4*s2}], %[la] offset:{2048*u}")
            if u % 2 == 0:
                ins.append("s_waitcnt lgkmcnt(1)")
            t, k = u % 2, u // 2
            mf(E(t), f"v[{160+4*(u%3)}:{163+4*(u%3)}]", B(k)[0])
            if u in (3, 7):   # fp8 A operands (2 KiB each) of tile t' = (u == 7), both products, read ahead of their use
                tt = 0 if u == 3 else 1
                for pr in range(2):
                    base = 172 + 8 * pr
                    ins.append(f"ds_read_b128 v[{base}:{base+3}], %[la] offset:{16384 + 4096*tt + 2048*pr}")
                    ins.append(f"ds_read_b128 v[{base+4}:{base+7}], %[la] offset:{16384 + 4096*tt + 2048*pr + 1024}")
            if u in (5, 7):
                tt = 0 if u == 5 else 1
                ins.append("s_waitcnt lgkmcnt(0)" if u == 7 else "s_waitcnt lgkmcnt(2)")
                for pr in range(2):
                    base = 172 + 8 * pr
                    ins.append(f"v_mfma_f32_16x16x128_f8f6f4 {C(tt)}, v[{base}:{base+7}], v[{16*pr}:{16*pr+7}], {C(tt)}")
                    state["nm"] += 0
                    for _ in range(4 * fn // fd if fd else 0):
                        ins.append(FILL[state["fi"] % 4]); state["fi"] += 1
        return ins, 24
    if order == "p32":
        # 32-KiB phases: one barrier and eight LDS-DMA loads per 16 units (the loop body covers two 8-unit rounds)
        for u in range(16):
            if u == 14:
                ins.append("s_waitcnt vmcnt(0)")
                ins.append("s_barrier")
                pending.extend(f"global_load_lds_dwordx4 %[voff], %[gb] offset:{1024*i}" for i in range(8))
            read(u)
            if u % 2 == 0:
                ins.append("s_waitcnt lgkmcnt(2)")
            t, k = u % 2, (u // 2) % 4
            (ah, al), (bh, bl) = A(u), B(k)
            mf(E(t), ah, bh); mf(C(t), al, bh); mf(C(t), ah, bl)
        return ins, state["nm"]          # (48 MFMAs per loop body: "cyc/phase" is per 16 units here)
    if order == "e1c2":
        for u in range(8):
            opening(u)
            read(u)
            if u % 2 == 0:
                ins.append("s_waitcnt lgkmcnt(2)")
            t, k = u % 2, u // 2
            (ah, al), (bh, bl) = A(u), B(k)
            mf(E(t), ah, bh); mf(C(t), al, bh); mf(C(t), ah, bl)
    else:
        # units in same-tile pairs (u, u+1): tile (u // 2) % 2, k-steps 2 (u // 4) + (u % 2)
        for u in range(0, 8, 2):
            t = (u // 2) % 2
            k0, k1 = 2 * (u // 4), 2 * (u // 4) + 1
            (ah0, al0), (ah1, al1) = A(u), A(u + 1)
            (bh0, bl0), (bh1, bl1) = B(k0), B(k1)
            opening(u)
            read(u)           # unit u + 2 -> the third set (free)
            ins.append("s_waitcnt lgkmcnt(2)")
            if order == "g24":
                mf(E(t), ah0, bh0); mf(E(t), ah1, bh1); mf(C(t), al0, bh0); mf(C(t), ah0, bl0)
                opening(u + 1)
                read(u + 1)   # unit u + 3 -> unit u's set
                mf(C(t), al1, bh1); mf(C(t), ah1, bl1)
            else:  # g222
                mf(C(t), al0, bh0); mf(C(t), ah0, bl0); mf(E(t), ah0, bh0)
                opening(u + 1)
                read(u + 1)
                mf(E(t), ah1, bh1); mf(C(t), al1, bh1); mf(C(t), ah1, bl1)
    return ins, state["nm"]


VARIANTS = []
for order in ("e1c2", "p32", "e1c2", "p32", "e1c2", "p32"):
    for fillers in ((3, 4),):
        VARIANTS.append(("w8", fillers, 1, "spread", order))
for shape in ():
    for fillers in ((0, 1), (1, 2), (1, 1), (2, 1), (3, 1)):
        if shape != "w4x" and fillers == (3, 1):
            continue
        for barrier, dma in ((0, None), (1, None), (1, "burst"), (1, "spread")):
            VARIANTS.append((shape, fillers, barrier, dma))


def main():
    out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <cstdlib>']
    names = []
    for i, (shape, fillers, barrier, dma, order) in enumerate(VARIANTS):
        ins, nm = body(shape, fillers, barrier, dma, order)
        waves = 8 if shape == "w8" else 4
        name = f"k{i}"
        flop = 32768 if shape == "w4x" else 16384
        shape_ = shape
        names.append((name, shape + ":" + order, waves, fillers, barrier, dma, nm, flop))
        clob = ", ".join(f'"v{r}"' for r in list(range(0, 184)) + list(range(200, 216)))
        init = [f"ds_read_b128 v[{r}:{r+3}], %[la] offset:{(r * 64) % 32768}" for r in range(0, 184, 4)]
        init += [f"ds_read_b128 v[{r}:{r+3}], %[la] offset:{(r * 64) % 32768}" for r in range(200, 216, 4)]
        init += ["s_waitcnt lgkmcnt(0)", "s_mov_b32 m0, %[m0v]", "s_memtime %[t0]", "s_memrealtime %[r0]", "s_waitcnt lgkmcnt(0)", ".Lloop_%=:"]
        tail = ["s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 .Lloop_%=", "s_waitcnt lgkmcnt(0)",
                "s_memtime %[t1]", "s_memrealtime %[r1]", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
        text = "\n".join(f'      "{t}\\n\\t"' for t in init + ins + tail)
        out.append(f'''
__global__ __launch_bounds__({64*waves}) void {name}(int iters, const char* w, const unsigned* rnd, unsigned long long* cyc) {{
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  for (int i = threadIdx.x; i < 24576; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = rnd[i];
  __syncthreads();
  const unsigned la = (threadIdx.x & 63) * 16;
  const unsigned voff = (threadIdx.x & 63) * 16;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned m0v = __builtin_amdgcn_readfirstlane(65536u + (wave & 3) * 4096u);
  unsigned long long gbv = (unsigned long long)(w + (size_t)blockIdx.x % 8 * 65536 + (wave & 3) * 4096);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)gbv), hi = __builtin_amdgcn_readfirstlane((unsigned)(gbv >> 32));
  const unsigned long long gb = ((unsigned long long)hi << 32) | lo;
  unsigned n = __builtin_amdgcn_readfirstlane(iters);
  unsigned long long t0, t1, r0, r1;
  asm volatile(
{text}
      : [n] "+s"(n), [t0] "=&s"(t0), [t1] "=&s"(t1), [r0] "=&s"(r0), [r1] "=&s"(r1)
      : [la] "v"(la), [voff] "v"(voff), [gb] "s"(gb), [m0v] "s"(m0v) : "memory", {clob});
  if ((threadIdx.x & 63) == 0) {{ cyc[(blockIdx.x * {waves} + wave) * 2] = t1 - t0; cyc[(blockIdx.x * {waves} + wave) * 2 + 1] = r1 - r0; }}
}}''')
    out.append('''
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  char* w; unsigned* rnd; unsigned long long* cyc;
  hipMalloc(&w, 1 << 20); hipMalloc(&rnd, 24576 * 4); hipMalloc(&cyc, 256 * 8 * 16);
  {
    std::vector<unsigned short> h(1 << 19);  // random fp16 values in (-2, 2)
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (unsigned short)(((s >> 16) & 0x8000u) | (0x3000u + ((s >> 8) & 0x0fffu))); }
    hipMemcpy(w, h.data(), 1 << 20, hipMemcpyHostToDevice); hipMemcpy(rnd, h.data(), 24576 * 4, hipMemcpyHostToDevice);
  }
  std::vector<unsigned long long> h(256 * 8 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-4s %-8s %5s %4s %7s | %9s %8s %7s %8s %8s\\n", "k", "shp", "fill", "bar", "dma", "cyc/phase", "pipe%", "GHz", "ms", "TF/s");
''')
    for (name, shape, waves, fillers, barrier, dma, nm, flop) in names:
        shape = shape
        out.append(f'''  {{
    hipFuncSetAttribute((const void*){name}, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    {name}<<<256, {64*waves}, 98304>>>(2000, w, rnd, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); {name}<<<256, {64*waves}, 98304>>>(iters, w, rnd, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), cyc, 256 * {waves} * 16, hipMemcpyDeviceToHost);
    double c = 0, r = 0; for (int i = 0; i < 256 * {waves}; ++i) {{ c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }}
    c /= 256 * {waves}; r /= 256 * {waves};
    const double per_simd_mfma = (double)iters * {nm} * {waves // 4};
    const double tf = 256.0 * 4 * per_simd_mfma * {flop}.0 / (ms * 1e-3) / 1e12;
    printf("%-4s %-8s %2d/%-2d %4d %7s | %9.1f %8.1f %7.3f %8.2f %8.1f\\n", "{name}", "{shape}", {fillers[0]}, {fillers[1]}, {barrier}, "{dma or '-'}",
           c / iters, 76800.0 / (c / iters), c / r * 0.1, ms, tf);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) {{ printf("error %s\\n", hipGetErrorString(e)); return 1; }}
  }}''')
    out.append("  return 0;\n}")
    open(sys.argv[1], "w").write("\n".join(out))


main()

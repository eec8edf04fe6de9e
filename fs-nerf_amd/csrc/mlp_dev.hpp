// mlp_dev.hpp — the NeRF MLP (src/core/models.py:111-143) for one tile of 128 samples per
// 256-thread workgroup (4 wavefronts, one per SIMD, 32 samples each), on the matrix cores.
//
//  * activations stay in registers from the positional encoding to sigma/rgb: each fp32
//    accumulator tile (32 features x 32 samples) is ReLU'd, split into bf16 high/low parts and
//    used directly as the B operand of the next layer's v_mfma_f32_32x32x16_bf16;
//  * weights (A operands, pre-packed by mlp_layout.hpp) are streamed L2 -> LDS by
//    global_load_lds_dwordx4 into a ring of 16-KiB phases shared by the 4 waves, LOOK phases
//    ahead, with counted vmcnt + raw s_barrier (no full drain inside the stream);
//  * x3 modes (FSN_PREC_FP16X3 default, FSN_PREC_BF16X3): a.w = ah.wh + al.wh + ah.wl, three MFMA
//    passes on 16-bit high/low parts with fp32 accumulation (fp16x3 ~ fp32 accuracy, bf16x3
//    ~1e-5 per product but no range limit); FSN_PREC_BF16 / FSN_PREC_FP16: one pass.
//  * sigma (256 -> 1) and rgb (128 -> 3) heads are fp32 VALU dot products on the accumulators.
#pragma once
#include "common.hpp"
#include "mlp_layout.hpp"

namespace fsn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;  // 8 raw 16-bit elements (bf16 or fp16 bits)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kNSlot = 4;            // LDS ring slots (phases)
#ifndef FSN_LEAD
#define FSN_LEAD 2
#endif
// A phase is "opened" (its loads waited for, workgroup barrier, next stage issued) kLead units
// before the previous phase's last unit, so that the first LDS reads of the new phase are issued
// underneath the tail MFMAs of the old one instead of behind the barrier.
constexpr int kLead = FSN_LEAD;
// phases staged ahead of the one being opened; with kLead > 0 the phase before the opened one is
// still being read, so one more slot must stay untouched
constexpr int kLook = kLead > 0 ? kNSlot - 2 : kNSlot - 1;
constexpr int kGldsPerWave = kPhaseBytes / 1024 / 4;  // 1-KiB glds instructions per wave per phase
constexpr int kRingBytes = kNSlot * kPhaseBytes;
constexpr int kAuxCapFloats = 3456;  // LDS reserved per network for biases / heads (8x256 needs 3392)
constexpr int kPeStashBytes = 2 * kKsPos * 256 * 16;  // 32 KiB: every lane's positional-encoding operands

struct Frag {  // one k-step (16 features x 32 samples) of activations as MFMA B operand
  s16x8 hi, lo;
};

// ---------------------------------------------------------------- weight stream
// Pass schedule: pass A (ptrA, nphA phases) repA times, then pass B repB times, repeating.
// All members are wave-uniform.
struct WStream {
  char* ring;
  uint32_t ring_lds;  // LDS byte address of the ring (M0 base of the LDS-DMA)
  const char *ptrA, *ptrB;
  uint32_t nphA, nphB, repA, repB;
  // stager state
  const char* s_ptr;
  uint32_t s_left, s_which, s_rep, s_slot;
  // consumer state
  uint32_t c_slot;
  const char* c_base;  // LDS address of the phase being computed (+ lane*16)
  const char* n_base;  // ... of the phase opened last

  __device__ __forceinline__ void begin_pass_(uint32_t which) {
    s_which = which;
    s_ptr = which ? ptrB : ptrA;
    s_left = which ? nphB : nphA;
  }
  __device__ __forceinline__ void advance_pass_() {
    const uint32_t my_rep = s_which ? repB : repA;
    if (++s_rep >= my_rep) {
      s_rep = 0;
      const uint32_t other = s_which ^ 1u;
      const uint32_t other_rep = other ? repB : repA;
      begin_pass_(other_rep ? other : s_which);
    } else {
      begin_pass_(s_which);
    }
  }
  // Issue the loads of the next phase to stage (this wave's quarter of it: 4 x 1 KiB).
  // Inline asm on purpose: (1) hipcc then does not see LDS-DMA writes and so does not put a
  // full `s_waitcnt vmcnt(0)` in front of every ds_read of the ring (cdna_hip_programming.md
  // section 5, "Three .s-level traps"); completion is tracked by hand in boundary(); (2) the saddr
  // form + immediate offsets need one M0 write and no per-load address VALU.  The immediate
  // offset of global_load_lds advances BOTH the global and the LDS address.
  __device__ __forceinline__ void stage() {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t m0v = ring_lds + s_slot * kPhaseBytes + wave * (kGldsPerWave * 1024);
    const uint32_t voff = (threadIdx.x >> 6) * (kGldsPerWave * 1024) + (threadIdx.x & 63) * 16;
    const uint64_t sp = (uint64_t)s_ptr;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32));
    const uint64_t sbase = ((uint64_t)hi << 32) | lo;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(m0v), "s"(sbase)
        : "memory");
    static_assert(kGldsPerWave == 4, "stage() issues exactly 4 loads per wave");
    s_ptr += kPhaseBytes;
    s_slot = (s_slot + 1) & (kNSlot - 1);
    if (--s_left == 0) advance_pass_();
  }
  __device__ __forceinline__ void init(char* ring_, const char* pA, uint32_t nA, uint32_t rA, const char* pB,
                                       uint32_t nB, uint32_t rB) {
    ring = ring_;
    ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring_;
    ptrA = pA; nphA = nA; repA = nA ? rA : 0;
    ptrB = pB; nphB = nB; repB = nB ? rB : 0;
    s_rep = 0; s_slot = 0; c_slot = 0;
    begin_pass_(repA ? 0u : 1u);
    c_base = n_base = ring + (threadIdx.x & 63) * 16;
#pragma unroll
    for (int i = 0; i < kLook; ++i) stage();
    if (kLead > 0) open_next();  // every pass finds its first phase already opened
  }
  // Open the next phase: its loads have landed for every wave (vmcnt counts in issue order, so
  // allowing the (kLook-1)*kGldsPerWave youngest loads to stay in flight retires exactly the oldest
  // staged phase), every wave is past the phase whose slot is restaged next.
  __device__ __forceinline__ void open_next() {
#ifdef FSN_ABL_NOSTREAM  // timing experiment: no waits, barriers or staging
    n_base = ring + c_slot * kPhaseBytes + (threadIdx.x & 63) * 16;
    c_slot = (c_slot + 1) & (kNSlot - 1);
    return;
#endif
    if (kLead > 0)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((kLook - 1) * kGldsPerWave) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((kLook - 1) * kGldsPerWave) : "memory");
    stage();
    n_base = ring + c_slot * kPhaseBytes + (threadIdx.x & 63) * 16;
    c_slot = (c_slot + 1) & (kNSlot - 1);
  }
  __device__ __forceinline__ void enter_phase() { c_base = n_base; }
  // before the workgroup exits (or touches the ring for anything else)
  __device__ __forceinline__ void drain() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
};

// LDS-only workgroup barrier that does not drain the weight stream
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------- helpers
// float -> 16-bit element of the mode (round to nearest even) and back
template <bool F16>
__device__ __forceinline__ short to_h(float v) {
  if (F16) { const _Float16 h = (_Float16)v; return __builtin_bit_cast(short, h); }
  const __bf16 h = (__bf16)v;
  return __builtin_bit_cast(short, h);
}
template <bool F16>
__device__ __forceinline__ float from_h(short b) {
  if (F16) return (float)__builtin_bit_cast(_Float16, b);
  return (float)__builtin_bit_cast(__bf16, b);
}
template <bool F16, bool X3>
__device__ __forceinline__ void split_store(const float v[8], Frag& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const short h = to_h<F16>(v[j]);
    f.hi[j] = h;
    f.lo[j] = X3 ? to_h<F16>(v[j] - from_h<F16>(h)) : (short)0;
  }
}

// Network description as the kernel needs it (wave-uniform).  Aux offsets follow build_geom():
// bias of GEMM l at l*D; w_sigma at (L+2)*D; w_rgb at (L+3)*D; misc at (L+5)*D.
struct NetParams {  // host -> kernel argument
  const char* blob;
  int32_t aux_off, aux_floats, stream_off;
  int32_t nph_density, nph_full;
  int32_t n_layers;
  uint32_t skip_mask;
  int32_t n_freqs_pos, n_freqs_dir;
};

struct NetDev {
  const float* aux;       // LDS copy of the blob's aux region
  const float* pos_mask;  // LDS, 64 floats (ones when no mask)
  const float* dir_mask;  // LDS, 32 floats
  char* pe_stash;         // LDS, this lane's slot of the [2*kKsPos][256 lanes] x 16 B operand stash
  int32_t n_layers;
  uint32_t skip_mask;
  int32_t n_freqs_pos, n_freqs_dir;
};

// sin and cos of a float32 argument, branch-free: Cody-Waite reduction by pi/2 in three fma
// steps (absolute error <= ~1e-7 for |a| <~ 1e4, i.e. far beyond 2^9 * scene extent), minimax
// polynomials on [-pi/4, pi/4], quadrant fix-up.  The reference evaluates torch.sin / torch.cos
// of the same float32 argument (models.py:37-39).
__device__ __forceinline__ void sincos_f32(float a, float& s_out, float& c_out) {
  const float n = rintf(a * 0.63661977236758134f);
  float r = __builtin_fmaf(-n, 1.57079625129699707031e+00f, a);
  r = __builtin_fmaf(-n, 7.54978941586159635335e-08f, r);
  r = __builtin_fmaf(-n, 5.39030252995776476554e-15f, r);
  const float r2 = r * r;
  float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(r2, ps, -1.6666654611e-1f);
  const float sn = __builtin_fmaf(r * r2, ps, r);
  float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(r2, pc, 4.166664568298827e-2f);
  const float cs = __builtin_fmaf(r2 * r2, pc, __builtin_fmaf(r2, -0.5f, 1.0f));
  const int q = (int)n & 3;
  const float s1 = (q & 1) ? cs : sn;
  const float c1 = (q & 1) ? sn : cs;
  s_out = (q & 2) ? -s1 : s1;
  c_out = ((q + 1) & 2) ? -c1 : c1;
}

// Positional / direction encoding of one sample straight into B-operand fragments.
// Slot layout = enc_slot_feature() in mlp_layout.hpp; value = reference feature (models.py:37-39)
// times the frequency mask (LDS, all ones when absent).  NKS k-steps (NKS*8 slots per lane half).
template <int NKS, bool F16, bool X3>
__device__ __forceinline__ void encode(float x0, float x1, float x2, int n_freqs, const float* __restrict__ freqs,
                                       const float* __restrict__ mask, int h, Frag (&out)[NKS]) {
  constexpr int SLOTS = 8 * NKS;
  constexpr int NPAIR = (SLOTS - 2) / 2;
  float v[SLOTS];
  const int P = 3 * n_freqs;
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) {
    const int p = 2 * i + h;
    // p = 3*band + coord, evaluated for both halves with compile-time indices, selected by h
    const int b0 = (2 * i) / 3, c0 = (2 * i) % 3, b1 = (2 * i + 1) / 3, c1 = (2 * i + 1) % 3;
    const float xa = (c0 == 0) ? x0 : (c0 == 1 ? x1 : x2);
    const float xb = (c1 == 0) ? x0 : (c1 == 1 ? x1 : x2);
    const float xc = h ? xb : xa;
    const int band = h ? b1 : b0, coord = h ? c1 : c0;
    const bool ok = p < P;
    const int bsafe = ok ? band : 0;
    float s, c;
    sincos_f32(xc * freqs[bsafe], s, c);
    v[2 * i] = ok ? s * mask[3 + bsafe * 6 + coord] : 0.f;
    v[2 * i + 1] = ok ? c * mask[3 + bsafe * 6 + 3 + coord] : 0.f;
  }
  const float ia = (h ? x2 : x0) * mask[h ? 2 : 0];
  const float ib = h ? 0.f : x1 * mask[1];
  v[SLOTS - 2] = ia;
  v[SLOTS - 1] = ib;
#pragma unroll
  for (int k = 0; k < NKS; ++k) split_store<F16, X3>(&v[8 * k], out[k]);
}

// ---------------------------------------------------------------- one GEMM layer
enum : int { EPI_RELU_CVT = 0, EPI_LAST_FULL = 1, EPI_LAST_DENS = 4, EPI_CVT = 2, EPI_RGB = 3 };

struct Heads {
  float sigma;  // partial dot (this lane half's features)
  float rgb[3];
};

// One unit: this wave's 32-sample slice of  acc[32 out x 32 samples] += W_unit[32 x 16] . act[16 x 32]
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const s16x8& a, const s16x8& b, const f32x16& c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int PREC>
__device__ __forceinline__ void unit_mfma(const char* ubase, const Frag& b, f32x16& acc) {
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
#ifdef FSN_ABL_NOLDS  // timing experiment: operands from registers instead of the LDS ring
  acc = mfma16<F16>(b.lo, b.hi, acc);
  if (X3) {
    acc = mfma16<F16>(b.hi, b.hi, acc);
    acc = mfma16<F16>(b.lo, b.lo, acc);
  }
  return;
#endif
  const s16x8 ah = *reinterpret_cast<const s16x8*>(ubase);
  acc = mfma16<F16>(ah, b.hi, acc);
  if (X3) {
    const s16x8 al = *reinterpret_cast<const s16x8*>(ubase + 1024);
    acc = mfma16<F16>(al, b.hi, acc);
    acc = mfma16<F16>(ah, b.lo, acc);
  }
}

// NT_OUT output tiles; KS_ACT k-steps from `act`, KS_ENC from `enc`; units are consumed in
// (tile, k-step) order starting phase-aligned.  Epilogue per finished tile:
//   EPI_RELU_CVT : out[2t..2t+1] = split(relu(acc))
//   EPI_LAST_*   : heads.sigma += w_sigma . relu(acc); _FULL also converts  (last hidden layer)
//   EPI_CVT      : out = split(acc)                                             (connection)
//   EPI_RGB      : heads.rgb[c] += w_rgb[c] . relu(acc)                        (branch)
// ENC_LDS: the encoding operands are read back from this lane's LDS stash (written once per tile by
// mlp_tile) instead of occupying 32 registers through the widest layer of the network.
template <int PREC, int NT_OUT, int KS_ACT, int KS_ENC, int EPI, bool ENC_LDS, int NACT, int NENC, int NOUT>
__device__ __forceinline__ void gemm_layer(WStream& st, const NetDev& net, int aux_bias, const Frag (&act)[NACT],
                                           const Frag (&enc)[NENC], Frag (&out)[NOUT], Heads& heads, int h) {
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
  constexpr int UPP = X3 ? 8 : 16;
  constexpr int UB = X3 ? 2048 : 1024;
  constexpr int KS = KS_ACT + KS_ENC;
  static_assert(KS_ACT <= NACT && (ENC_LDS || KS_ENC <= NENC), "operand arrays too small");
  const float* bias = net.aux + aux_bias;
#pragma unroll
  for (int t = 0; t < NT_OUT; ++t) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 32 * t + 8 * q + 4 * h);
      acc[4 * q + 0] = bv[0]; acc[4 * q + 1] = bv[1]; acc[4 * q + 2] = bv[2]; acc[4 * q + 3] = bv[3];
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      constexpr int dummy = 0; (void)dummy;
      const int u = t * KS + ks;  // compile-time after unrolling
      constexpr int TOTAL = NT_OUT * KS;
      // open the phase that starts kLead units from here (the one after this layer's last unit too)
      if (((u + kLead) % UPP == 0 && u + kLead <= TOTAL && (kLead > 0 || u < TOTAL)) ||
          (kLead > 0 && u + kLead == TOTAL && TOTAL % UPP != 0))
        st.open_next();
      if (u % UPP == 0) st.enter_phase();
      const char* ub = st.c_base + (u % UPP) * UB;
      if (ks < KS_ACT) {
        unit_mfma<PREC>(ub, act[ks < KS_ACT ? ks : 0], acc);
      } else if (ENC_LDS) {
        Frag e;
        const int k = ks - KS_ACT;
        e.hi = *reinterpret_cast<const s16x8*>(net.pe_stash + (2 * k) * 4096);
        if (X3) e.lo = *reinterpret_cast<const s16x8*>(net.pe_stash + (2 * k + 1) * 4096);
        unit_mfma<PREC>(ub, e, acc);
      } else {
        unit_mfma<PREC>(ub, enc[ks >= KS_ACT ? ks - KS_ACT : 0], acc);
      }
    }
    // ---- epilogue of tile t
    if (EPI == EPI_RELU_CVT || EPI == EPI_LAST_FULL || EPI == EPI_LAST_DENS || EPI == EPI_RGB) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {  // relu on the sign bit: one v_max_i32, no canonicalising pre-max as fmaxf has
        const int b = __builtin_bit_cast(int, (float)acc[i]);
        acc[i] = __builtin_bit_cast(float, b < 0 ? 0 : b);
      }
    }
    if (EPI == EPI_LAST_FULL || EPI == EPI_LAST_DENS) {
      const float* ws = net.aux + (net.n_layers + 2) * (NT_OUT * 32);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + 32 * t + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) heads.sigma = __builtin_fmaf(wv[i], acc[4 * q + i], heads.sigma);
      }
    }
    if (EPI == EPI_RGB) {
      const float* wr = net.aux + (net.n_layers + 3) * (NT_OUT * 64);  // D = 2*NT_OUT*32 for the branch
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c * (NT_OUT * 32) + 32 * t + 8 * q + 4 * h);
#pragma unroll
          for (int i = 0; i < 4; ++i) heads.rgb[c] = __builtin_fmaf(wv[i], acc[4 * q + i], heads.rgb[c]);
        }
      }
    }
#ifdef FSN_ABL_NOCVT  // timing experiment: skip the fp32 -> hi/lo split of the layer output
    if (EPI == EPI_RELU_CVT || EPI == EPI_CVT || EPI == EPI_LAST_FULL) {
      asm volatile("" ::"v"(acc));
      out[(2 * t) < NOUT ? 2 * t : 0] = act[0];
      out[(2 * t + 1) < NOUT ? 2 * t + 1 : 0] = act[0];
    }
    if (false) {
#else
    if (EPI == EPI_RELU_CVT || EPI == EPI_CVT || EPI == EPI_LAST_FULL) {
#endif
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = acc[i];
      split_store<F16, X3>(&v[0], out[(2 * t) < NOUT ? 2 * t : 0]);
      split_store<F16, X3>(&v[8], out[(2 * t + 1) < NOUT ? 2 * t + 1 : 0]);
    }
  }
}

// ---------------------------------------------------------------- whole network, one tile
// `src` supplies this lane's sample on demand: src.pos(x,y,z) and src.dir(x,y,z) (lanes l and l+32
// hold the same sample).  The encoded position is parked in LDS for the wide (skip) layers instead
// of keeping its 32 operand registers alive across the hidden layers, and the direction is read
// only in front of the branch layer: the register file (512 per lane) is the scarce resource of
// this kernel.  Outputs (valid in all lanes): sigma, and rgb when FULL.
template <int NT, int PREC, bool FULL, class Src>
__device__ __forceinline__ void mlp_tile(WStream& st, const NetDev& net, const Src& src, float& sigma,
                                         float (&rgb)[3]) {
  constexpr int NA = 2 * NT;
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
  const int h = (threadIdx.x >> 5) & 1;
  constexpr int D = 32 * NT;
  const int L = net.n_layers;
  const float* misc = net.aux + (L + 5) * D;
  Frag A[NA], B[NA];
  Frag none[1];
  Heads heads{0.f, {0.f, 0.f, 0.f}};
  {
    Frag pe[kKsPos];
    float px, py, pz;
    src.pos(px, py, pz);
    encode<kKsPos, F16, X3>(px, py, pz, net.n_freqs_pos, misc + 4, net.pos_mask, h, pe);
    if (net.skip_mask) {  // park the operands for the wide (skip) layers in this lane's LDS slot
#pragma unroll
      for (int k = 0; k < kKsPos; ++k) {
        *reinterpret_cast<s16x8*>(net.pe_stash + (2 * k) * 4096) = pe[k].hi;
        if (X3) *reinterpret_cast<s16x8*>(net.pe_stash + (2 * k + 1) * 4096) = pe[k].lo;
      }
    }
    gemm_layer<PREC, NT, 0, kKsPos, EPI_RELU_CVT, false>(st, net, 0, none, pe, A, heads, h);
  }
#define FSN_HIDDEN(EPI, IN, OUT, LIDX)                                                              \
  do {                                                                                              \
    if ((net.skip_mask >> ((LIDX)-1)) & 1u)                                                         \
      gemm_layer<PREC, NT, NA, kKsPos, EPI, true>(st, net, (LIDX)*D, IN, none, OUT, heads, h);      \
    else                                                                                            \
      gemm_layer<PREC, NT, NA, 0, EPI, false>(st, net, (LIDX)*D, IN, none, OUT, heads, h);          \
  } while (0)
  for (int l = 1; l <= L - 2; l += 2) {
    FSN_HIDDEN(EPI_RELU_CVT, A, B, l);
    if (l + 1 <= L - 2) {
      FSN_HIDDEN(EPI_RELU_CVT, B, A, l + 1);
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) A[i] = B[i];
    }
  }
  // last hidden layer (index L-1): sigma head on its fp32 output (models.py:127,141)
  FSN_HIDDEN((FULL ? EPI_LAST_FULL : EPI_LAST_DENS), A, B, L - 1);
#undef FSN_HIDDEN
  sigma = heads.sigma + __shfl_xor(heads.sigma, 32, 64) + misc[0];
  if (FULL) {
    // connection (no activation, models.py:130), then branch on [feat, dir_enc] (models.py:131-133)
    gemm_layer<PREC, NT, NA, 0, EPI_CVT, false>(st, net, L * D, B, none, A, heads, h);
    Frag de[kKsDir];
    float dx, dy, dz;
    src.dir(dx, dy, dz);
    encode<kKsDir, F16, X3>(dx, dy, dz, net.n_freqs_dir, misc + 20, net.dir_mask, h, de);
    gemm_layer<PREC, NT / 2, NA, kKsDir, EPI_RGB, false>(st, net, (L + 1) * D, A, de, B, heads, h);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float z = heads.rgb[c] + __shfl_xor(heads.rgb[c], 32, 64) + misc[1 + c];
      rgb[c] = 1.0f / (1.0f + expf(-z));  // sigmoid (models.py:135)
    }
  }
}

// Copy a blob's aux region and the two frequency masks into LDS (all threads of the workgroup;
// the caller synchronises afterwards) and describe the net.  lds: aux_floats + 96 floats;
// pe_stash: kPeStashBytes of LDS shared by both networks of a kernel.
__device__ __forceinline__ void load_net(const NetParams& p, const float* __restrict__ pos_mask_g,
                                         const float* __restrict__ dir_mask_g, float* lds, char* pe_stash,
                                         NetDev& net) {
  const f32x4* src = reinterpret_cast<const f32x4*>(p.blob + p.aux_off);
  f32x4* dst = reinterpret_cast<f32x4*>(lds);
  for (int i = threadIdx.x; i < p.aux_floats / 4; i += blockDim.x) dst[i] = src[i];
  float* pm = lds + p.aux_floats;
  float* dm = pm + 64;
  const int npe = 3 * (1 + 2 * p.n_freqs_pos), nde = 3 * (1 + 2 * p.n_freqs_dir);
  for (int i = threadIdx.x; i < 64; i += blockDim.x) pm[i] = (pos_mask_g && i < npe) ? pos_mask_g[i] : 1.0f;
  for (int i = threadIdx.x; i < 32; i += blockDim.x) dm[i] = (dir_mask_g && i < nde) ? dir_mask_g[i] : 1.0f;
  net.aux = lds;
  net.pos_mask = pm;
  net.dir_mask = dm;
  net.pe_stash = pe_stash + threadIdx.x * 16;
  net.n_layers = p.n_layers;
  net.skip_mask = p.skip_mask;
  net.n_freqs_pos = p.n_freqs_pos;
  net.n_freqs_dir = p.n_freqs_dir;
}

}  // namespace fsn

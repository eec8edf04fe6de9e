// mlp_dev.hpp — the NeRF MLP (src/core/models.py:111-143) for one tile of 128 samples per
// 512-thread workgroup: 8 wavefronts, two per SIMD, 16 samples each, on the matrix cores.
//
//  * activations stay in registers from the positional encoding to sigma/rgb: a pair of fp32
//    accumulator tiles (2 x 16 features x 16 samples) is ReLU'd, split into 16-bit high/low parts
//    and is, lane for lane, the B operand of one k-step of the next layer's
//    v_mfma_f32_16x16x32_{f16,bf16};
//  * two waves per SIMD (<= 256 registers each): while one wave converts, waits for LDS or sits at
//    the phase barrier, its partner keeps the matrix pipe busy;
//  * weights (A operands, pre-packed by mlp_layout.hpp) are streamed L2 -> LDS by
//    global_load_lds_dwordx4 into a ring of 16-KiB phases shared by the 8 waves, two phases
//    ahead, with counted vmcnt + raw s_barrier (no full drain inside the stream);
//  * x3 modes (FSN_PREC_FP16X3 default, FSN_PREC_BF16X3): a.w = ah.wh + (al.wh + ah.wl), three MFMA
//    passes on 16-bit high/low parts with fp32 accumulation; the two correction products have their
//    own accumulator, and in the fp16 modes the low parts are kept scaled by 2^11 (mlp_layout.hpp,
//    kLoScaleF16) so that they are normal fp16 numbers for |v| down to ~6e-5: fp16x3 ~ fp32 accuracy
//    for layer scales from 2^-14 to 65504 (both ends are detected, never silent); bf16x3
//    ~1e-5 per product, float32's range; FSN_PREC_BF16 / FSN_PREC_FP16: one pass.
//  * FSN_PREC_FP16X3U (PREC 4, round 4; inference kernels): the same three products with UNSCALED low parts, all into one
//    accumulator tile, no merge and a three-instruction split in the epilogue (rounds 1-2's arithmetic, 6 % faster).  It
//    is float32-grade because the packed network is SCALED: a power of two per layer, folded into weights / biases /
//    heads by the packer (mlp_pack.hpp) and calibrated from the layers' measured maxima, keeps every layer's activations
//    at 2^4 .. 2^10 whatever the network's own scale.  Both ends of the calibration are detected (range_layer_end).
//  * sigma (256 -> 1) and rgb (128 -> 3) heads are fp32 VALU dot products on the accumulators.
#pragma once
#include "common.hpp"
#include "mlp_layout.hpp"
#ifndef FSN_KLOOP_HEADER
#define FSN_KLOOP_HEADER "kloop_gen.hpp"
#endif
#include FSN_KLOOP_HEADER

#include <type_traits>
#include <utility>

namespace fsn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;  // 8 raw 16-bit elements (bf16 or fp16 bits)
typedef __attribute__((ext_vector_type(4))) float f32x4;

// threadIdx.x behind an empty asm (FSN_LAUNDER_ALL, a measured experiment, off): values derived from it are recomputed
// where they are used instead of being hoisted out of the tile loops and spilled.  It takes the fused kernel to ZERO
// spilled VGPRs in bf16 and 11 in fp16x3 - and makes the frames 8 % and 3.5 % SLOWER (141.2 against 130.5 ms, 427.2
// against 412.8 ms): the recomputation lands inside the unit loop, and the volatile asm statements order against the
// hand-scheduled blocks.  The per-group laundering in render.hip (57 -> 20 spills) is the part that pays.
#ifdef FSN_LAUNDER_ALL
__device__ __forceinline__ unsigned fsn_tidx() {
  unsigned t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
#define FSN_TIDX (fsn_tidx())
#else
#define FSN_TIDX threadIdx.x
#endif

#ifndef FSN_NSLOT
#define FSN_NSLOT 4
#endif
#ifndef FSN_LAG
#define FSN_LAG 0
#endif
constexpr int kNSlot = FSN_NSLOT;     // LDS ring slots (phases)
// FSN_LAG = 1: waves 4..7 of the workgroup run the weight stream ONE PHASE behind waves 0..3 (their SIMD partners):
// they execute the same instruction stream, delayed by one barrier event (WStream::pass_begin / pass_end), so that
// one wave's pair epilogue / bookkeeping falls into the middle of its partner's MFMA k-loop instead of both waves
// leaving the matrix pipe idle at the same time.  Needs two more ring slots (see kLook).
constexpr int kLag = FSN_LAG;
#ifndef FSN_LEAD
#ifdef FSN_KLOOP_ASM_D3
#define FSN_LEAD 3
#else
#define FSN_LEAD 2
#endif
#endif
// A phase is "opened" (its loads waited for, workgroup barrier, next stage issued) kLead units
// before the previous phase's last unit, so that the first LDS reads of the new phase are issued
// underneath the tail MFMAs of the old one instead of behind the barrier.
constexpr int kLead = FSN_LEAD;
// phases staged ahead of the one being opened; with kLead > 0 the phase before the opened one is
// still being read, so one more slot must stay untouched
// Ring safety (E_b = the b-th barrier event; a phase q is read by the leading waves between E_q and E_q+1 - plus
// the two units after E_q+1 on the compiler-scheduled path - and by lagging waves one event later; the stage issued
// at E_b overwrites the slot of phase b + kLook - kNSlot): kNSlot >= kLook + 2 + kLag.
constexpr int kLook = kLead > 0 ? kNSlot - 2 - kLag : kNSlot - 1;
// loads of this wave that may stay in flight when a phase is opened: the lagging waves' share of a phase must have
// landed one event before they read it themselves (the leading waves read it then), hence "- kLag"
// LDS-DMA is issued by kLoaders loader waves only (one per SIMD; which half of the workgroup: FSN_LOADER_XOR): issuing a 1-KiB load costs the wave
// 60-185 cycles, and with the workgroup barrier in front of it every wave of the CU used to pay that at the same
// moment; now the partner wave of each loader (w + 4) runs its MFMAs meanwhile.
constexpr int kLoaders = 4;
#ifndef FSN_LOADER_XOR  // 4: the loaders are waves 4..7, the later-dispatched wave of every SIMD (0: waves 0..3;
#define FSN_LOADER_XOR 4  // measured 396.6 against 399.9 ms per fp16x3 frame, no difference in bf16)
#endif
constexpr int kOpenVmcnt = (kLook - 1 - kLag) * (kPhaseBytes / 1024 / kLoaders);
#ifndef FSN_RING_EXPERIMENT  // (timing experiments with other ring depths: compiler-scheduled paths only)
static_assert(kOpenVmcnt == 4, "the generated k-loop blocks wait with vmcnt(4)");
#endif
__device__ __forceinline__ uint32_t slot_add(uint32_t s, uint32_t k) {  // (s + k) mod kNSlot, s < kNSlot, k <= kNSlot
  if ((kNSlot & (kNSlot - 1)) == 0) return (s + k) & (kNSlot - 1);
  const uint32_t t = s + k;
  return t >= (uint32_t)kNSlot ? t - kNSlot : t;
}
constexpr int kWaves = 8;             // wavefronts per workgroup (two per SIMD)
constexpr int kThreads = 64 * kWaves;
constexpr int kGldsPerWave = kPhaseBytes / 1024 / kLoaders;  // 1-KiB LDS-DMA instructions per loader wave per phase
constexpr int kRingBytes = kNSlot * kPhaseBytes;
constexpr int kAuxCapFloats = 3456;  // LDS reserved per network for biases / heads (8x256 needs 3392)
constexpr int kTileCols = 128;       // samples per workgroup tile = columns of a T-layout tile (train_fused.hip)
constexpr int kTRow = 16;            // T-layout: samples of a chunk (consecutive pair-rows lie kTRow x NPL dwords apart)
// T-layout (train_fused.hip) of a saved matrix with 2 P rows, per 128-sample tile: [chunk of 16 samples][P pair-rows]
// [16 samples][NPL parts] dwords, NPL = 2 in the x3 modes (high part, low part side by side), 1 in the single-pass
// modes.  Offset of the first part of (pair-row pr, sample s of the tile) inside the tile's block of 2 P x 128 dwords:
__host__ __device__ constexpr int64_t t_layout_off(int NPL, int P, int pr, int s) {
  return (((int64_t)(s >> 4) * P + pr) * kTRow + (s & 15)) * NPL;
}

// properties of the arithmetic mode PREC (FSN_PREC_*)
constexpr bool prec_f16(int P) { return P >= 2; }
constexpr bool prec_x3(int P) { return (P & 1) == 0; }
constexpr bool prec_lo_scaled_k(int P) { return P == 2 || P == 6; }  // fp16 low parts stored as fp16((v - high) * 2^11)
#ifndef FSN_BF16X3_ONEACC  // bf16x3 (low parts unscaled anyway) on the one-accumulator form too in the INFERENCE kernels:
#define FSN_BF16X3_ONEACC 0  // 383.3 against 399.3 ms per headline frame (render.hip / render_occ.hip / mlp.hip set it)
#endif
constexpr bool prec_one_acc(int P) { return P == 4 || (FSN_BF16X3_ONEACC && P == 0); }  // one accumulator per tile, no merge

struct Frag {  // one k-step (32 features x 16 samples) of activations as MFMA B operand
  s16x8 hi, lo;
};

// ---------------------------------------------------------------- weight stream
// Pass schedule: pass A (ptrA, nphA phases) repA times, then pass B repB times, repeating.
// All members are wave-uniform.
struct WStream {
  char* ring;
  uint32_t ring_lds;  // LDS byte address of the ring (M0 base of the LDS-DMA)
  uint32_t m0_base;   // ... + this wave's share of a phase (loader waves; SGPR)
  uint32_t is_loader;  // this wave issues LDS-DMA (one per SIMD)
  const char *ptrA, *ptrB;
  uint32_t nphA, nphB, repA, repB;
  // stager state
  const char* s_ptr;
  uint32_t s_left, s_which, s_rep, s_slot;
#ifdef FSN_STAMP  // diagnostic build: per-wave cycle totals (s_memtime) of the k-loop blocks / epilogues
  uint64_t t_k = 0, t_e = 0, t_n = 0, t_ko = 0, t_eo = 0;
#endif
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
  // Issue the loads of the next phase to stage (this wave's eighth of it: 2 x 1 KiB).
  // Inline asm on purpose: (1) hipcc then does not see LDS-DMA writes and so does not put a
  // full `s_waitcnt vmcnt(0)` in front of every ds_read of the ring (cdna_hip_programming.md
  // section 5, "Three .s-level traps"); completion is tracked by hand in open_next(); (2) the saddr
  // form + immediate offsets need one M0 write and no per-load address VALU.  The immediate
  // offset of global_load_lds advances BOTH the global and the LDS address.
  __device__ __forceinline__ void stage() {
    // wave-uniform SALU arithmetic only (as next_stage below): the M0 base and the loader predicate are formed once
    // in init(), the end of a pass is a real branch
    const uint32_t m0v = m0_base + s_slot * kPhaseBytes;
    const uint32_t voff = ((FSN_TIDX >> 6) ^ FSN_LOADER_XOR) * (kGldsPerWave * 1024) + (FSN_TIDX & 63) * 16;
    const uint64_t sbase = (uint64_t)s_ptr;
    uint32_t keep;
    if (is_loader) {
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
    }
    static_assert(kGldsPerWave == 4, "stage() issues exactly 4 loads per loader wave");
    s_ptr += kPhaseBytes;
    s_slot = slot_add(s_slot, 1);
    if (__builtin_expect(--s_left == 0, 0)) {
      asm volatile("" ::: "memory");  // keep it a branch (no if-conversion into s_cselect chains)
      advance_pass_();
    }
  }
  __device__ __forceinline__ void init(char* ring_, const char* pA, uint32_t nA, uint32_t rA, const char* pB,
                                       uint32_t nB, uint32_t rB) {
    ring = ring_;
    ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring_;
    m0_base = __builtin_amdgcn_readfirstlane(ring_lds + ((FSN_TIDX >> 6) ^ FSN_LOADER_XOR) * (kGldsPerWave * 1024));
    is_loader = __builtin_amdgcn_readfirstlane((((FSN_TIDX >> 6) ^ FSN_LOADER_XOR) < (uint32_t)kLoaders) ? 1u : 0u);
    ptrA = pA; nphA = nA; repA = nA ? rA : 0;
    ptrB = pB; nphB = nB; repB = nB ? rB : 0;
    s_rep = 0; s_slot = 0; c_slot = 0;
    begin_pass_(repA ? 0u : 1u);
    c_base = n_base = ring + (FSN_TIDX & 63) * 16;
#pragma unroll
    for (int i = 0; i < kLook; ++i) stage();
    if (kLead > 0) open_next();  // every pass finds its first phase already opened
  }
  // Data-dependent schedules (render_occ.hip: the number of density-only and full tiles of a batch is only known
  // while it runs).  Both kinds of tile stream the SAME blob from its first phase and differ in where they stop, so the
  // stager can always wrap to phase 0; a tile says how long it is when it starts.  At that moment exactly kLook + 1
  // phases of it are staged (the invariant of init() / the phase openings).  init() with nphB = kDynamicPhases.
  static constexpr uint32_t kDynamicPhases = 1u << 30;
  __device__ __forceinline__ void begin_tile(uint32_t nph) { s_left = nph - (uint32_t)(kLook + 1); }
  // Open the next phase: its loads have landed for every wave (vmcnt counts in issue order, so
  // allowing the (kLook-1)*kGldsPerWave youngest loads to stay in flight retires exactly the oldest
  // staged phase), every wave is past the phase whose slot is restaged next.
  __device__ __forceinline__ void open_next() {
#ifdef FSN_ABL_NOSTREAM  // timing experiment: no waits, barriers or staging
    n_base = ring + c_slot * kPhaseBytes + (FSN_TIDX & 63) * 16;
    c_slot = slot_add(c_slot, 1);
    return;
#endif
#ifdef FSN_NLW_SKIP_VMWAIT
    // Only the loader waves have LDS-DMA loads in flight; a non-loader wave's vmcnt counts nothing but its OWN stores
    // (the training savers), which nobody needs to wait for here: it goes straight to the barrier, behind which every
    // loader has seen its share of the phase land.
    if (kLead > 0) {
      if (is_loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kOpenVmcnt) : "memory");
      asm volatile("s_barrier" ::: "memory");
    } else
#else
    if (kLead > 0)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kOpenVmcnt) : "memory");
    else
#endif
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((kLook - 1) * kGldsPerWave) : "memory");
    stage();
    n_base = ring + c_slot * kPhaseBytes + (FSN_TIDX & 63) * 16;
    c_slot = slot_add(c_slot, 1);
  }
  __device__ __forceinline__ void enter_phase() { c_base = n_base; }
  // ---- interface of the hand-scheduled k-loop blocks (kloop_gen.hpp), which carry the phase openings inside
  // their instruction stream.  next_stage(): M0 value and global base of the next phase to stage, advancing the
  // stager exactly as stage() does.  phase_lds(k): LDS byte address (+ 16*lane) of the k-th phase counted from the
  // one opened last (k = 0).  opened(n): the block executed n openings.
  __device__ __forceinline__ void next_stage(uint32_t& m0v, uint64_t& gbase) {
    // (wave-uniform SALU arithmetic only: the M0 base of this wave's share of slot 0 is formed once in init(); the
    // end of a pass - once in 60..72 stages - is a real branch, not a chain of selects on every stage: the gaps
    // between the blocks carried ~60 scalar instructions per output pair before, tools/kloop_interblock.py)
    m0v = m0_base + s_slot * kPhaseBytes;
    gbase = (uint64_t)s_ptr;
    s_ptr += kPhaseBytes;
    s_slot = slot_add(s_slot, 1);
    if (__builtin_expect(--s_left == 0, 0)) {
      asm volatile("" ::: "memory");  // keep it a branch (no if-conversion into s_cselect chains)
      advance_pass_();
    }
  }
  __device__ __forceinline__ uint32_t phase_lds(uint32_t k) const {
    return ring_lds + slot_add(slot_add(c_slot, kNSlot - 1), k) * kPhaseBytes + (FSN_TIDX & 63) * 16;
  }
  __device__ __forceinline__ void opened(uint32_t n) {
    c_slot = slot_add(c_slot, n);
    n_base = ring + slot_add(c_slot, kNSlot - 1) * kPhaseBytes + (FSN_TIDX & 63) * 16;
    c_base = n_base;
  }
  // ---- the one-phase lag between the two waves of a SIMD (kLag).  A "pass" is a run of MLP tiles between two
  // workgroup barriers of the caller.  Waves 4..7 join one barrier event before their first unit (they then read
  // phase p while waves 0..3 read phase p+1); waves 0..3 join one after their last so that every wave has taken
  // part in the same number of events when the pass ends.  No staging, no waits: events only.
  static __device__ __forceinline__ bool lagging() { return kLag && (__builtin_amdgcn_readfirstlane(FSN_TIDX) >= 256); }
  __device__ __forceinline__ void pass_begin() const {
#ifdef FSN_ABL_LAGSLEEP  // timing experiment (with the barrier-free ablation of the generator): waves 4..7 start every
    // pass FSN_ABL_LAGSLEEP x 64 cycles late, so that the two waves of a SIMD run out of step
    if (__builtin_amdgcn_readfirstlane(FSN_TIDX) >= 256) __builtin_amdgcn_s_sleep(FSN_ABL_LAGSLEEP);
#endif
    if (kLag && lagging()) asm volatile("s_barrier" ::: "memory");
  }
  __device__ __forceinline__ void pass_end() const {
    if (kLag && !lagging()) asm volatile("s_barrier" ::: "memory");
  }
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
// plain C++ form of the high / low split (every mode); fp16 x3: low part scaled by 2^11 (mlp_layout.hpp)
template <bool F16, bool X3, bool LS = (F16 && X3)>
__device__ __forceinline__ void split_store_cpp(const float v[8], Frag& f) {
  constexpr float K = LS ? kLoScaleF16 : 1.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const short h = to_h<F16>(v[j]);
    f.hi[j] = h;
    f.lo[j] = X3 ? to_h<F16>((v[j] - from_h<F16>(h)) * K) : (short)0;
  }
}

// The split the epilogues use.  NOTE (fp16 x3, asm form): the inputs must be results of VALU instructions the compiler
// knows (the epilogue's merge / ReLU), never MFMA accumulators read directly - hipcc pads the XDL-write -> VALU-read
// wait states only for instructions it sees.
template <bool F16, bool X3, bool LS = (F16 && X3)>
__device__ __forceinline__ void split_store(const float v[8], Frag& f) {
#ifndef FSN_SPLIT_CPP
  if constexpr (F16 && X3 && !LS) {
    // UNSCALED low parts (FSN_PREC_FP16X3U): packed convert + one mixed-precision fma per value, low = fp16(v -
    // float(high)) with a single rounding - bit-identical to split_store_cpp (checked on 65,536 values incl. subnormals
    // in round 2).  Three instructions per two values.
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t hh, ll;
      asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
          "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
          "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
          : "=&v"(hh), "=&v"(ll)
          : "v"(v[2 * i]), "v"(v[2 * i + 1]));
      h[i] = hh;
      l[i] = ll;
    }
    f.hi = __builtin_bit_cast(s16x8, h);
    f.lo = __builtin_bit_cast(s16x8, l);
    return;
  }
  if constexpr (F16 && X3 && LS) {
    // fp16 high part and SCALED low part of a pair of values in five instructions: packed round-to-nearest convert;
    // r = v - float(high) as one mixed-precision fma each (fp32 result: exact, |r| <= 2^-11 |v|); low = fp16(2^11 r),
    // again as mixed fma so that the scale costs nothing.  Bit-identical to split_store_cpp (one rounding, in the
    // last step).  Written as asm: hipcc folds fma(h, -1, v) back into a subtraction and then forms v_pk_add_f32.
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 h, l;
    const float kk = kLoScaleF16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t hh, ll;
      float ra, rb;
      asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
          "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
          "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
          "v_fma_mixlo_f16 %1, %2, %6, 0 op_sel_hi:[0,0,0]\n\t"
          "v_fma_mixhi_f16 %1, %3, %6, 0 op_sel_hi:[0,0,0]"
          : "=&v"(hh), "=&v"(ll), "=&v"(ra), "=&v"(rb)
          : "v"(v[2 * i]), "v"(v[2 * i + 1]), "s"(kk));
      h[i] = hh;
      l[i] = ll;
    }
    f.hi = __builtin_bit_cast(s16x8, h);
    f.lo = __builtin_bit_cast(s16x8, l);
    return;
  }
#endif
  split_store_cpp<F16, X3, LS>(v, f);
}

// Range guard of the fp16 modes.  range_track: running packed maximum of the |high part| bit patterns a lane has
// produced in the current layer (4 instructions per output pair).  range_layer_end: folds it into the tile's overall
// maximum (0x7c00 = infinity: a value left the fp16 range, FSN_STATUS_FP16_RANGE) and, in the split modes' forward
// passes (SMALL), looks at the layer's scale: when the largest |high part| of the wave's 16 samples x all features of
// this layer is non-zero but an fp16 SUBNORMAL (< 2^-14) the split no longer carries 22 bits relative to the layer's
// scale (high parts on the fixed 2^-24 grid, scaled low parts 2^-36: FSN_STATUS_FP16_SMALL).  Both ends are reported, the host re-runs / continues in bf16x3 (core/models.py).
template <bool SIGNED>
__device__ __forceinline__ void range_track(uint32_t& fmax, const s16x8& hi) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  const u32x4 w = __builtin_bit_cast(u32x4, hi);
  if constexpr (!SIGNED) {
    // non-negative fp16 values order like their bit patterns; infinity (0x7c00) and NaN propagate through the IEEE
    // maximum: two 3-input packed maxima per pair instead of four 2-input ones
    asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(fmax) : "v"(w[0]), "v"(w[1]));
    asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(fmax) : "v"(w[2]), "v"(w[3]));
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t x = w[i] & 0x7fff7fffu;
    asm("v_pk_max_u16 %0, %0, %1" : "+v"(fmax) : "v"(x));
  }
}

constexpr uint32_t kSmallBits = 0x0400u;  // fp16 bit pattern of 2^-14, the smallest normal number

struct RangeState {
  uint32_t fmax;   // this layer, packed pair of u16
  uint32_t fall;   // all layers of the tile so far
  uint32_t small;  // wave-uniform: some layer's scale was below kSmallBits
};

// FSN_PREC_FP16X3U: the low parts are unscaled, i.e. fp16 subnormals (fixed 2^-24 resolution) for |v| < 2^-3; the scaled
// network keeps a layer's largest activation at 2^4 .. 2^10 (calibration target 2^10), so a wavefront whose layer
// maximum is below 2^-4 says that the layer's scale no longer fits the data: reported like the other modes' low end.
constexpr uint32_t kSmallBitsU = 0x2c00u;  // fp16 bit pattern of 2^-4

template <bool SMALL, uint32_t BITS = kSmallBits>
__device__ __forceinline__ void range_layer_end(RangeState& r) {
  asm("v_pk_max_u16 %0, %0, %1" : "+v"(r.fall) : "v"(r.fmax));
  if constexpr (SMALL) {
    const uint32_t lo = r.fmax & 0xffffu, hi = r.fmax >> 16;
    const uint32_t m = lo > hi ? lo : hi;
    const int any_big = __builtin_amdgcn_readfirstlane((int)__any(m >= BITS));
    const int any_nz = __builtin_amdgcn_readfirstlane((int)__any(m != 0u));
    if (!any_big && any_nz) r.small = 1u;
  }
  r.fmax = 0u;
}

// tell the host (never silent): bit 0 = a value reached fp16 infinity, bit 1 = a layer ran below the split's scale
__device__ __forceinline__ void range_report(uint32_t* status, const RangeState& r, uint32_t range_bit = 1u) {
  const uint32_t f = r.fall;
  const bool bad = (f & 0xffffu) >= 0x7c00u || (f >> 16) >= 0x7c00u;
  uint32_t bits = __builtin_amdgcn_readfirstlane((int)__any(bad)) ? range_bit : 0u;
  if (__builtin_amdgcn_readfirstlane((int)r.small)) bits |= 2u;
  if (bits && status) { if ((FSN_TIDX & 63) == 0) __hip_atomic_fetch_or(status, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
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
  uint32_t* status;  // device word or null: bit 0 is set when a hidden activation leaves the fp16 range
};

struct NetDev {
  const float* aux;       // LDS copy of the blob's aux region
  const float* pos_mask;  // LDS, 64 floats (ones when no mask)
  const float* dir_mask;  // LDS, 32 floats
  int32_t n_layers;
  uint32_t skip_mask;
  int32_t n_freqs_pos, n_freqs_dir;
  uint32_t* status;
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
// times the frequency mask (LDS, all ones when absent).  NKS k-steps (NKS*8 slots per lane; the
// four lanes g = lane>>4 of a sample share the NKS*32 slots).
// `save` (training forward only, else null): this lane's column of the encoding's packed T-layout buffer; slot
// (k-step ks, element j) is row 32 ks + 8 g + j, i.e. B-operand order (train_fused.hip).
template <int NKS, bool F16, bool X3, bool SAVE = false, bool LS = (F16 && X3)>
__device__ __forceinline__ void encode(float x0, float x1, float x2, int n_freqs, const float* __restrict__ freqs,
                                       const float* __restrict__ mask, int g, Frag (&out)[NKS], uint32_t* save = nullptr) {
  constexpr int SLOTS = 8 * NKS;
  float v[SLOTS];
  const int P = 3 * n_freqs;
#pragma unroll
  for (int i = 0; i < SLOTS / 2; ++i) {
    const int p = 4 * i + g;
    const bool ok = p < P;
    const int ps = ok ? p : 0;
    const int band = (ps * 11) >> 5;  // ps / 3 for ps < 32
    const int coord = ps - 3 * band;
    const float xc = coord == 0 ? x0 : (coord == 1 ? x1 : x2);
    float s, c;
    if constexpr (X3) {
      sincos_f32(xc * freqs[band], s, c);
    } else {
      // single-pass modes: the features are rounded to 16 bits (relative 2^-9 / 2^-12) right below, so the hardware
      // sine / cosine of the reduced argument is exact enough: argument in revolutions, reduced by v_fract
      // (absolute error <= |arg| 6e-8 / 2pi revolutions, ~1e-4 rad at the highest frequency), 5 instructions
      // instead of ~30 per pair.  The x3 (parity) modes never take this branch.
      const float rev = (xc * freqs[band]) * 0.15915494309189535f;
      const float fr = __builtin_amdgcn_fractf(rev);
      s = __builtin_amdgcn_sinf(fr);
      c = __builtin_amdgcn_cosf(fr);
    }
    v[2 * i] = ok ? s * mask[3 + band * 6 + coord] : 0.f;
    v[2 * i + 1] = ok ? c * mask[3 + band * 6 + 3 + coord] : 0.f;
  }
  if (g == 2) {
    v[SLOTS - 2] = x0 * mask[0];
    v[SLOTS - 1] = x1 * mask[1];
  }
  if (g == 3) v[SLOTS - 2] = x2 * mask[2];
#pragma unroll
  for (int k = 0; k < NKS; ++k) split_store<F16, X3, LS>(&v[8 * k], out[k]);
#ifdef FSN_ABL_SAVE_NOLOADER
  if (SAVE && (((FSN_TIDX >> 6) ^ FSN_LOADER_XOR) < (uint32_t)kLoaders)) return;
#endif
  if constexpr (SAVE) {
    // packed T-layout (train_fused.hip): slots (k, 2i), (k, 2i+1) are rows 32k + 8g + 2i, +1 = pair-row 16k + 4g + i;
    // `save` = this lane's sample at pair-row 4g (t_layout_off); x3: the two parts of a pair as one 8-byte store
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
#pragma unroll
    for (int k = 0; k < NKS; ++k) {
      const u32x4 h = __builtin_bit_cast(u32x4, out[k].hi), l = __builtin_bit_cast(u32x4, out[k].lo);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (X3) __builtin_nontemporal_store((u32x2){h[i], l[i]}, reinterpret_cast<u32x2*>(save + (16 * k + i) * kTRow * 2));
        else __builtin_nontemporal_store(h[i], save + (16 * k + i) * kTRow);
      }
    }
  }
}

// ---------------------------------------------------------------- one GEMM layer
enum : int { EPI_RELU_CVT = 0, EPI_LAST_FULL = 1, EPI_LAST_DENS = 4, EPI_CVT = 2, EPI_RGB = 3, EPI_NONE = 5 };

struct Heads {
  float sigma;  // partial dot (this lane group's features)
  float rgb[3];
  RangeState rs;  // fp16 modes: range guard
};

// One unit: this wave's 16-sample slice of  acc[16 out x 16 samples] += W_unit[16 x 32] . act[32 x 16]
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const s16x8& a, const s16x8& b, const f32x4& c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
struct AFrag {  // A operand (weights) of one unit
  s16x8 hi, lo;
};
// A operands are read from the LDS ring one k-step (two units) ahead of their MFMAs; the pair in flight
// carries over from GEMM to GEMM, tile to tile and pass to pass: on entry to a GEMM `cur` holds its
// units 0 and 1.
#ifndef FSN_KLOOP_D
#define FSN_KLOOP_D 2
#endif
#ifndef FSN_KLOOP_NSETS
#define FSN_KLOOP_NSETS (FSN_KLOOP_D + 1)
#endif
constexpr int kKD = FSN_KLOOP_D;  // A-operand units landed ahead of a pair on entry to a hand-scheduled block
constexpr int kNS = FSN_KLOOP_NSETS;  // A register sets in rotation (unit u lives in set u mod kNS)
struct ARing {
  AFrag cur[kNS > 3 ? kNS : 3];  // hand-scheduled path: kNS sets in rotation; compiler-scheduled prefetch modes: cur[0..1]
};
template <int PREC>
__device__ __forceinline__ void load_afrag(const char* p, AFrag& f) {
  f.hi = *reinterpret_cast<const s16x8*>(p);
  if ((PREC & 1) == 0 && PREC != 6) f.lo = *reinterpret_cast<const s16x8*>(p + 1024);
}
// acc: main sum (high x high); cor: the correction products (x3 modes; scaled by 2^11 in the fp16 modes)
template <int PREC>
__device__ __forceinline__ void unit_mfma_r(const AFrag& a, const Frag& b, f32x4& acc, f32x4& cor) {
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
  acc = mfma16<F16>(a.hi, b.hi, acc);
  if (X3) {
    f32x4& c = prec_one_acc(PREC) ? acc : cor;  // (FSN_PREC_FP16X3U: everything into the main tile)
    if (PREC != 6) c = mfma16<F16>(a.lo, b.hi, c);  // (fp16x2: the weights' low parts are dropped)
    c = mfma16<F16>(a.hi, b.lo, c);
  }
}

template <int PREC>
__device__ __forceinline__ void unit_mfma(const char* ubase, const Frag& b, f32x4& acc, f32x4& cor) {
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
#ifdef FSN_ABL_NOLDS  // timing experiment: operands from registers instead of the LDS ring
  acc = mfma16<F16>(b.lo, b.hi, acc);
  if (X3) {
    acc = mfma16<F16>(b.hi, b.hi, acc);
    acc = mfma16<F16>(b.lo, b.lo, acc);
  }
  return;
#endif
#ifdef FSN_ABL_LDSDUMMY  // timing experiment: LDS reads are issued but no MFMA depends on them
  {
    // FSN_ABL_LDSDUMMY = 1: the kernel's two 16-byte reads; 2: one 16-byte read; 3: two 8-byte reads; 4: four 8-byte
#if FSN_ABL_LDSDUMMY == 1
    const s16x8 dh = *reinterpret_cast<const s16x8*>(ubase);
    const s16x8 dl = *reinterpret_cast<const s16x8*>(ubase + 1024);
#elif FSN_ABL_LDSDUMMY == 2
    const s16x8 dh = *reinterpret_cast<const s16x8*>(ubase);
    const int dl = 0;
#elif FSN_ABL_LDSDUMMY == 3
    typedef __attribute__((ext_vector_type(2))) int i32x2;
    const i32x2 dh = *reinterpret_cast<const i32x2*>(ubase - (FSN_TIDX & 63) * 8);
    const i32x2 dl = *reinterpret_cast<const i32x2*>(ubase - (FSN_TIDX & 63) * 8 + 1024);
#else
    typedef __attribute__((ext_vector_type(2))) int i32x2;
    const i32x2 dh = *reinterpret_cast<const i32x2*>(ubase - (FSN_TIDX & 63) * 8);
    const i32x2 dl = *reinterpret_cast<const i32x2*>(ubase - (FSN_TIDX & 63) * 8 + 512);
    const i32x2 d2 = *reinterpret_cast<const i32x2*>(ubase - (FSN_TIDX & 63) * 8 + 1024);
    const i32x2 d3 = *reinterpret_cast<const i32x2*>(ubase - (FSN_TIDX & 63) * 8 + 1536);
    asm volatile("" ::"v"(d2), "v"(d3));
#endif
    acc = mfma16<F16>(b.lo, b.hi, acc);
    if (X3) {
      acc = mfma16<F16>(b.hi, b.hi, acc);
      acc = mfma16<F16>(b.lo, b.lo, acc);
    }
    asm volatile("" ::"v"(dh), "v"(dl));
    return;
  }
#endif
  const s16x8 ah = *reinterpret_cast<const s16x8*>(ubase);
  acc = mfma16<F16>(ah, b.hi, acc);
  if (X3) {
    f32x4& c = prec_one_acc(PREC) ? acc : cor;
    if (PREC != 6) {
      const s16x8 al = *reinterpret_cast<const s16x8*>(ubase + 1024);
      c = mfma16<F16>(al, b.hi, c);
    }
    c = mfma16<F16>(ah, b.lo, c);
  }
}

__device__ __forceinline__ float relu_f32(float v) {  // on the sign bit: one v_max_i32, no canonicalising pre-max
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, b < 0 ? 0 : b);
}

// Epilogue hooks.  A hook sees each finished output pair (8 fp32 values per lane, features
// 32 tp + 16 (j>>2) + 4 g + (j&3) of this lane's sample) after the activation and before the 16-bit split.
// Inference uses NoHook (no code); the training kernels (train_fused.hip) save / mask / store there.
struct NoHook {
  static constexpr bool kZeroInit = false;  // accumulators start from the bias
  static constexpr bool kPacked = false;    // no store<X3>(tp, frag) of the pair's 16-bit parts (training savers: true)
  static constexpr bool kLayerEnd = false;  // no layer_end(fmax) call with the layer's packed |high part| maximum
  __device__ __forceinline__ void pre(int) {}
  __device__ __forceinline__ void post(int, float (&)[8]) {}
};

// Epilogue of one finished output pair (shared by the compiler-scheduled and the hand-scheduled k-loops).
// acc0 / acc1: main sums of the pair's two tiles; cor0 / cor1: their correction sums (x3 modes, else unused).
template <int PREC, int NP_OUT, int EPI, int NOUT, class HK>
__device__ __forceinline__ void pair_epilogue(const NetDev& net, int tp, const f32x4& acc0, const f32x4& acc1,
                                              const f32x4& cor0, const f32x4& cor1, Frag (&out)[NOUT], Heads& heads,
                                              int g, HK& hk) {
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0, LS = prec_lo_scaled_k(PREC);
#if defined(FSN_PRIO) && FSN_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#elif defined(FSN_PRIO) && FSN_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#elif defined(FSN_PRIO) && FSN_PRIO == 4
    __builtin_amdgcn_s_setprio(2);
#endif
    float v[8];
    if constexpr (X3 && !prec_one_acc(PREC)) {
      // value = main + 2^-11 x corrections (fp16 modes: exact power-of-two unscaling inside the fma); bf16: scale 1
      constexpr float IK = F16 ? 1.0f / kLoScaleF16 : 1.0f;
#ifndef FSN_EPI_PK_FMA
      // eight v_fma_f32: measured 0.6 % faster on the frame than four v_pk_fma_f32 (418.7 -> 416.2 ms), as
      // MI355X_MICROARCH.md prices packed fp32 VALU beside MFMAs
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = __builtin_fmaf(cor0[j], IK, acc0[j]); v[4 + j] = __builtin_fmaf(cor1[j], IK, acc1[j]); }
#else
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      const f32x2 ik2 = {IK, IK};
#pragma unroll
      for (int j = 0; j < 4; j += 2) {  // v_pk_fma_f32: two values per instruction
        const f32x2 a0 = {acc0[j], acc0[j + 1]}, c0 = {cor0[j], cor0[j + 1]};
        const f32x2 a1 = {acc1[j], acc1[j + 1]}, c1 = {cor1[j], cor1[j + 1]};
        const f32x2 r0 = __builtin_elementwise_fma(c0, ik2, a0), r1 = __builtin_elementwise_fma(c1, ik2, a1);
        v[j] = r0[0]; v[j + 1] = r0[1];
        v[4 + j] = r1[0]; v[4 + j + 1] = r1[1];
      }
#endif
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = acc0[j]; v[4 + j] = acc1[j]; }
    }
    if constexpr (!X3 && EPI == EPI_RELU_CVT && std::is_same<HK, NoHook>::value) {
      // single-pass modes: convert first, ReLU on the packed 16-bit values (sign bit = top bit of either format, so a
      // signed 16-bit max with 0 is the ReLU; rounding keeps the sign, so relu(round(x)) == round(relu(x))):
      // 4 + 4 instructions per pair instead of 8 + 4
      Frag& o = out[tp < NOUT ? tp : 0];
      split_store<F16, false>(v, o);
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
      u32x4 w = __builtin_bit_cast(u32x4, o.hi);
#pragma unroll
      for (int i = 0; i < 4; ++i) asm("v_pk_max_i16 %0, %0, 0" : "+v"(w[i]));
      o.hi = __builtin_bit_cast(s16x8, w);
      if constexpr (F16) range_track<false>(heads.rs.fmax, o.hi);
      asm volatile("" : "+v"(o.hi));
      return;
    }
    if (EPI != EPI_CVT && EPI != EPI_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = relu_f32(v[j]);
    }
    hk.post(tp, v);
    if (EPI == EPI_LAST_FULL || EPI == EPI_LAST_DENS) {
      const float* ws = net.aux + (net.n_layers + 2) * (NP_OUT * 32) + 32 * tp + 4 * g;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws), w1 = *reinterpret_cast<const f32x4*>(ws + 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        heads.sigma = __builtin_fmaf(w0[j], v[j], heads.sigma);
        heads.sigma = __builtin_fmaf(w1[j], v[4 + j], heads.sigma);
      }
    }
    if (EPI == EPI_RGB) {
      const float* wr = net.aux + (net.n_layers + 3) * (NP_OUT * 64) + 32 * tp + 4 * g;  // D = 64 NP_OUT here
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + c * (NP_OUT * 32));
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + c * (NP_OUT * 32) + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          heads.rgb[c] = __builtin_fmaf(w0[j], v[j], heads.rgb[c]);
          heads.rgb[c] = __builtin_fmaf(w1[j], v[4 + j], heads.rgb[c]);
        }
      }
    }
    // The empty asm statements pin each pair's results here.  Without them LLVM sinks the pure conversion
    // arithmetic to its first use (the next layer): all accumulator pairs of the layer then stay alive until
    // its end (64 more registers) and the conversions run in one block after the last MFMA.
    if (EPI == EPI_LAST_FULL || EPI == EPI_LAST_DENS) asm volatile("" : "+v"(heads.sigma));
    if (EPI == EPI_RGB) asm volatile("" : "+v"(heads.rgb[0]), "+v"(heads.rgb[1]), "+v"(heads.rgb[2]));
    if (EPI == EPI_RELU_CVT || EPI == EPI_CVT || EPI == EPI_LAST_FULL) {
      Frag& o = out[tp < NOUT ? tp : 0];
#ifdef FSN_ABL_NOCVT  // timing experiment: skip the fp32 -> hi/lo split
      asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
      o = out[0];
#else
      // The asm form of the split must not read MFMA results directly (hipcc pads the XDL-write -> VALU-read wait
      // states only for instructions it knows; found as 1e-3 gradient errors in the backward chain).  In the x3
      // modes v[] comes out of the merge fma above, a compiler-visible VALU instruction; the single-pass modes'
      // split is plain C++.
      // (FSN_PREC_FP16X3U has no merge: where the epilogue has no ReLU either - the connection layer - v[] ARE the
      // accumulators, and the asm split must not read them: the C++ form there)
      if constexpr (prec_one_acc(PREC) && (EPI == EPI_CVT || EPI == EPI_NONE)) split_store_cpp<F16, X3, LS>(v, o);
      else split_store<F16, X3, LS>(v, o);
      if constexpr (F16) range_track<EPI == EPI_CVT || EPI == EPI_NONE>(heads.rs.fmax, o.hi);
#endif
      if (X3) asm volatile("" : "+v"(o.hi), "+v"(o.lo));
      else asm volatile("" : "+v"(o.hi));
    }
    // training savers keep the pair's 16-bit high / low parts (the operands the weight-gradient GEMM consumes), not
    // the fp32 values: where the epilogue has just formed them for the next layer they cost nothing extra
    if constexpr (HK::kPacked) {
      if constexpr (EPI == EPI_RELU_CVT || EPI == EPI_CVT || EPI == EPI_LAST_FULL) {
        hk.template store<X3>(tp, out[tp < NOUT ? tp : 0]);
      } else {
        Frag t;
        split_store_cpp<F16, X3, LS>(v, t);
        hk.template store<X3>(tp, t);
      }
    }
}

// ---------------------------------------------------------------- hand-scheduled k-loop blocks (kloop_gen.hpp)
// FSN_KLOOP_ASM (defined by the translation unit) selects them for the x3 / x2 modes of 256-wide networks: one
// inline-asm block per output pair with the A operands two units ahead of their MFMAs (three register sets in
// rotation), counted waits, and the weight-phase openings inside the stream.  tools/gen_kloop.py documents the
// operand contract.  The blocks exist for (units per pair, offset of the pair's first unit inside a phase):
// 16/0 (256->256), 20/0,4 (skip layer), 18/0,2,4,6 (branch), 4/0,4 (first layer).
#ifdef FSN_KLOOP_ASM
constexpr bool kKloopAsm = true;
#else
constexpr bool kKloopAsm = false;
#endif

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// The pinned accumulator sets of the hand-scheduled path: main sums E = v[240:247], O = v[248:255] (a pair uses one,
// the other receives the next pair's bias), corrections C = v[232:239]; tile 0 / tile 1 of each.
struct AccSets {
  f32x4 e0, e1, o0, o1, c0, c1;
};

#define FSN_KLOOP_PIN_X3                                                                                           \
  "+{v[240:243]}"(acc.e0), "+{v[244:247]}"(acc.e1), "+{v[248:251]}"(acc.o0), "+{v[252:255]}"(acc.o1),              \
      "=&{v[232:235]}"(acc.c0), "=&{v[236:239]}"(acc.c1)
#define FSN_KLOOP_PIN_X2 FSN_KLOOP_PIN_X3
// X3S (FSN_PREC_FP16X3U): no correction set - v[232:239] stay with the register allocator
#define FSN_KLOOP_PIN_X3S \
  "+{v[240:243]}"(acc.e0), "+{v[244:247]}"(acc.e1), "+{v[248:251]}"(acc.o0), "+{v[252:255]}"(acc.o1)
#define FSN_KLOOP_SETS_X3                                                                                        \
  [s0h] "+v"(s0.hi), [s0l] "+v"(s0.lo), [s1h] "+v"(s1.hi), [s1l] "+v"(s1.lo), [s2h] "+v"(s2.hi), [s2l] "+v"(s2.lo), \
      [keep] "=&s"(keep)
#define FSN_KLOOP_SETS_X2 [s0h] "+v"(s0.hi), [s1h] "+v"(s1.hi), [s2h] "+v"(s2.hi), [keep] "=&s"(keep)
#define FSN_KLOOP_SETS_X3S FSN_KLOOP_SETS_X3
#define FSN_B(k) bsel<k, KS_ACT, KS_ENC>(act, enc)
#define FSN_KLOOP_INS                                                                                            \
  [b0h] "v"(FSN_B(0).hi), [b0l] "v"(FSN_B(0).lo), [b1h] "v"(FSN_B(1).hi), [b1l] "v"(FSN_B(1).lo),                \
      [b2h] "v"(FSN_B(2).hi), [b2l] "v"(FSN_B(2).lo), [b3h] "v"(FSN_B(3).hi), [b3l] "v"(FSN_B(3).lo),            \
      [b4h] "v"(FSN_B(4).hi), [b4l] "v"(FSN_B(4).lo), [b5h] "v"(FSN_B(5).hi), [b5l] "v"(FSN_B(5).lo),            \
      [b6h] "v"(FSN_B(6).hi), [b6l] "v"(FSN_B(6).lo), [b7h] "v"(FSN_B(7).hi), [b7l] "v"(FSN_B(7).lo),            \
      [b8h] "v"(FSN_B(8).hi), [b8l] "v"(FSN_B(8).lo), [b9h] "v"(FSN_B(9).hi), [b9l] "v"(FSN_B(9).lo),            \
      [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [abn] "v"(abn), [voff] "v"(voff),          \
      [mv0] "s"(mv[0]), [gb0] "s"(gb[0]), [mv1] "s"(mv[1]), [gb1] "s"(gb[1]), [mv2] "s"(mv[2]), [gb2] "s"(gb[2])
// one asm statement: text variant TXT (both MFMA element types), output list OUTS.  "scc": the phase openings compare
// (s_cmp_ge_u32); without the clobber hipcc may carry a condition of its own across the block (found when the standalone
// MLP kernel was built on these blocks: its stream pointers were then selected by the block's comparison)
#define FSN_KLOOP_EMIT(TXT, OUTS)                                                                       \
  do {                                                                                                  \
    if constexpr (F16) asm volatile(TXT("v_mfma_f32_16x16x32_f16") : OUTS : FSN_KLOOP_INS : "memory", "scc");  \
    else asm volatile(TXT("v_mfma_f32_16x16x32_bf16") : OUTS : FSN_KLOOP_INS : "memory", "scc");               \
  } while (0)
#define FSN_KLOOP_CASE(MODE, NU_, OFF_)                                                                         \
  if constexpr (NU == NU_ && OFF == OFF_) {                                                                     \
    constexpr int EV = FSN_KLOOP_##NU_##_##OFF_##_EVENTS, NP = FSN_KLOOP_##NU_##_##OFF_##_PHASES;               \
    kloop_plan<EV, NP>(st, a, mv, gb);                                                                          \
    if constexpr (PAR == 0)                                                                                     \
      FSN_KLOOP_EMIT(FSN_KLOOP_##MODE##_##NU_##_##OFF_##_N0, FSN_KLOOP_PIN_##MODE FSN_COMMA FSN_KLOOP_SETS_##MODE); \
    else                                                                                                        \
      FSN_KLOOP_EMIT(FSN_KLOOP_##MODE##_##NU_##_##OFF_##_N1, FSN_KLOOP_PIN_##MODE FSN_COMMA FSN_KLOOP_SETS_##MODE); \
    st.opened(EV);                                                                                              \
  }
#define FSN_COMMA ,

// phase addresses and stage descriptors of a block with EV openings touching NP phases; unused slots repeat slot 0
template <int EV, int NP>
__device__ __forceinline__ void kloop_plan(WStream& st, uint32_t (&a)[4], uint32_t (&mv)[3], uint64_t (&gb)[3]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = st.phase_lds(k < NP ? k : 0);
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    if (e < EV) st.next_stage(mv[e], gb[e]);
    else { mv[e] = mv[0]; gb[e] = gb[0]; }
  }
  if (EV == 0) { mv[0] = mv[1] = mv[2] = 0; gb[0] = gb[1] = gb[2] = 0; }
}

// B operand of k-step K: activations first, then the encoding; K past the last k-step repeats the last one (the
// block's text does not reference it)
template <int K, int KS_ACT, int KS_ENC, int NACT, int NENC>
__device__ __forceinline__ const Frag& bsel(const Frag (&act)[NACT], const Frag (&enc)[NENC]) {
  constexpr int kk = K < KS_ACT + KS_ENC ? K : KS_ACT + KS_ENC - 1;
  if constexpr (kk < KS_ACT) return act[kk];
  else return enc[kk - KS_ACT];
}

// One output pair: NU = 2 x k-steps units starting OFF units into the current phase.  The pair's main sums accumulate
// in set E (PAR 0) or O (PAR 1), which holds its bias on entry; the other set receives the next pair's bias (abn =
// its LDS address); the correction sums are written to set C from zero.
// s0,s1: A sets holding units 0,1 on entry; on exit units NU, NU+1 sit in sets (NU % 3), ((NU+1) % 3) of (s0,s1,s2).
// MODE: 0 = X3 (correction set), 1 = X2, 2 = X3S (one accumulator)
template <bool F16, int MODE, int KS_ACT, int KS_ENC, int OFF, int PAR, int NACT, int NENC>
__device__ __forceinline__ void kloop_block(WStream& st, const Frag (&act)[NACT], const Frag (&enc)[NENC],
                                            AccSets& acc, uint32_t abn, AFrag& s0, AFrag& s1, AFrag& s2) {
  constexpr int NU = 2 * (KS_ACT + KS_ENC);
  uint32_t a[4], mv[3], keep;
  uint64_t gb[3];
  const uint32_t voff = ((FSN_TIDX >> 6) ^ FSN_LOADER_XOR) * (kGldsPerWave * 1024) + (FSN_TIDX & 63) * 16;
  if constexpr (MODE == 2) {
    FSN_KLOOP_CASE(X3S, 16, 0)
    FSN_KLOOP_CASE(X3S, 20, 0)
    FSN_KLOOP_CASE(X3S, 20, 4)
    FSN_KLOOP_CASE(X3S, 18, 0)
    FSN_KLOOP_CASE(X3S, 18, 2)
    FSN_KLOOP_CASE(X3S, 18, 4)
    FSN_KLOOP_CASE(X3S, 18, 6)
    FSN_KLOOP_CASE(X3S, 4, 0)
    FSN_KLOOP_CASE(X3S, 4, 4)
  } else if constexpr (MODE == 0) {
    FSN_KLOOP_CASE(X3, 16, 0)
    FSN_KLOOP_CASE(X3, 20, 0)
    FSN_KLOOP_CASE(X3, 20, 4)
    FSN_KLOOP_CASE(X3, 18, 0)
    FSN_KLOOP_CASE(X3, 18, 2)
    FSN_KLOOP_CASE(X3, 18, 4)
    FSN_KLOOP_CASE(X3, 18, 6)
    FSN_KLOOP_CASE(X3, 4, 0)
    FSN_KLOOP_CASE(X3, 4, 4)
  } else {
    FSN_KLOOP_CASE(X2, 16, 0)
    FSN_KLOOP_CASE(X2, 20, 0)
    FSN_KLOOP_CASE(X2, 20, 4)
    FSN_KLOOP_CASE(X2, 18, 0)
    FSN_KLOOP_CASE(X2, 18, 2)
    FSN_KLOOP_CASE(X2, 18, 4)
    FSN_KLOOP_CASE(X2, 18, 6)
    FSN_KLOOP_CASE(X2, 4, 0)
    FSN_KLOOP_CASE(X2, 4, 4)
  }
  (void)keep;
}

// NP_OUT output pairs (32 features = two 16-row tiles); KS_ACT k-steps (of 32) from `act`, KS_ENC
// from `enc`; units are consumed in (pair, k-step, half) order starting phase-aligned.  Epilogue
// per finished pair (8 values per lane: features 32tp + 16(j>>2) + 4g + (j&3)):
//   EPI_RELU_CVT : out[tp] = split(relu(acc))
//   EPI_LAST_*   : heads.sigma += w_sigma . relu(acc); _FULL also converts  (last hidden layer)
//   EPI_CVT      : out[tp] = split(acc)                                        (connection)
//   EPI_RGB      : heads.rgb[c] += w_rgb[c] . relu(acc)                        (branch)
//   EPI_NONE     : nothing (the hook consumes the values; last GEMM of the backward chain)
template <int PREC, int NP_OUT, int KS_ACT, int KS_ENC, int EPI, int NACT, int NENC, int NOUT, class HK>
__device__ __forceinline__ void gemm_layer(WStream& st, const NetDev& net, int aux_bias, const Frag (&act)[NACT],
                                           const Frag (&enc)[NENC], Frag (&out)[NOUT], Heads& heads, ARing& ring,
                                           int g, HK& hk) {
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
  // range guard (fp16 modes): layers whose epilogue forms 16-bit parts close their per-layer maximum; the split modes'
  // forward passes (bias-initialised accumulators; the backward chain starts from zero) also check the layer's scale
  constexpr bool kConverts = EPI == EPI_RELU_CVT || EPI == EPI_CVT || EPI == EPI_LAST_FULL;
  constexpr bool kCheckSmall = X3 && !HK::kZeroInit;
  constexpr uint32_t kSmallThr = prec_one_acc(PREC) ? kSmallBitsU : kSmallBits;
  static_assert(kLead >= 2, "the A-operand prefetch distance (one k-step = two units) must not exceed the phase lead");
  constexpr bool PREFETCH = !X3;
#ifdef FSN_X3_PF1
  constexpr bool kX3Pf1 = X3;
#else
  constexpr bool kX3Pf1 = false;
#endif
  constexpr int UPP = X3 ? 8 : 16;
  constexpr int UB = X3 ? 2048 : 1024;
  constexpr int KS = KS_ACT + KS_ENC;
  constexpr int TOTAL = 2 * NP_OUT * KS;
  static_assert(KS_ACT <= NACT && KS_ENC <= NENC, "operand arrays too small");
  const float* bias = net.aux + aux_bias;
  if constexpr (kKloopAsm && X3 && (KS == 8 || KS == 10 || KS == 9 || KS == 2) &&
                (KS_ACT == 8 || (KS_ACT == 0 && NP_OUT == 8)) && !HK::kZeroInit) {
    // hand-scheduled path (256-wide networks): ring.cur[0], cur[1] hold this GEMM's units 0 and 1 on entry and the
    // next GEMM's on exit; inside, the three A sets rotate by NU units per pair (compile-time indices).  The pairs
    // alternate between two pinned accumulator sets; each pair's epilogue runs right after its block.  (Running the
    // epilogue of pair tp-1 inside the instruction stream of pair tp - interleaved, or as one burst for waves 4..7
    // only - was measured slower, DESIGN.md 4.3, and is no longer generated.)
    constexpr int NU = 2 * KS;
    static_assert(TOTAL % UPP == 0, "a GEMM of the hand-scheduled path ends on a phase boundary");
    static_assert(kNS == 3, "three A register sets");
    AccSets acc;
    acc.e0 = *reinterpret_cast<const f32x4*>(bias + 4 * g);
    acc.e1 = *reinterpret_cast<const f32x4*>(bias + 16 + 4 * g);
    acc.o0 = acc.e0;
    acc.o1 = acc.e1;
    const uint32_t bias_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)(bias + 4 * g);
    static_for<NP_OUT>([&](auto TP) __attribute__((always_inline)) {
      constexpr int tp = decltype(TP)::value;
      constexpr int NS = kNS;  // A register sets
      constexpr int R0 = (tp * NU) % NS, OFF = (tp * NU) % UPP, PAR = tp & 1;
      hk.pre(tp);
#if defined(FSN_PRIO) && FSN_PRIO == 4
      if (__builtin_amdgcn_readfirstlane(FSN_TIDX) & 256) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
#ifdef FSN_STAMP
      const uint64_t ts0 = __builtin_amdgcn_s_memtime();
#endif
      const uint32_t abn = bias_lds + 128u * (tp + 1 < NP_OUT ? tp + 1 : tp);  // (last pair: a harmless reload)
      kloop_block<F16, PREC == 6 ? 1 : (prec_one_acc(PREC) ? 2 : 0), KS_ACT, KS_ENC, OFF, PAR>(st, act, enc, acc, abn, ring.cur[R0], ring.cur[(R0 + 1) % NS],
                                                            ring.cur[(R0 + 2) % NS]);
#ifdef FSN_STAMP
      const uint64_t ts1 = __builtin_amdgcn_s_memtime();
#endif
      if constexpr (PAR == 0) pair_epilogue<PREC, NP_OUT, EPI>(net, tp, acc.e0, acc.e1, acc.c0, acc.c1, out, heads, g, hk);
      else pair_epilogue<PREC, NP_OUT, EPI>(net, tp, acc.o0, acc.o1, acc.c0, acc.c1, out, heads, g, hk);
#ifdef FSN_STAMP
      {
        const uint64_t ts2 = __builtin_amdgcn_s_memtime();
        if (KS == 8) {  // the 256 -> 256 GEMMs
          st.t_k += ts1 - ts0;
          st.t_e += ts2 - ts1;
          st.t_n += 1;
        } else {  // first layer, skip layer, branch
          st.t_ko += ts1 - ts0;
          st.t_eo += ts2 - ts1;
        }
      }
#endif
    });
    // bring the sets holding the next GEMM's first kKD units back to cur[0..kKD-1]
    constexpr int RE = TOTAL % kNS;
    if constexpr (RE != 0) {
      AFrag t[kNS];
#pragma unroll
      for (int i = 0; i < kNS; ++i) t[i] = ring.cur[(RE + i) % kNS];
#pragma unroll
      for (int i = 0; i < kNS; ++i) ring.cur[i] = t[i];
    }
    if constexpr (F16 && kConverts) {
      if constexpr (HK::kLayerEnd) hk.layer_end(heads.rs.fmax);
      range_layer_end<kCheckSmall, kSmallThr>(heads.rs);
    }
    return;
  }
#pragma unroll
  for (int tp = 0; tp < NP_OUT; ++tp) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 cor0 = {0.f, 0.f, 0.f, 0.f}, cor1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!HK::kZeroInit) {
      acc0 = *reinterpret_cast<const f32x4*>(bias + 32 * tp + 4 * g);
      acc1 = *reinterpret_cast<const f32x4*>(bias + 32 * tp + 16 + 4 * g);
    }
    hk.pre(tp);
#if defined(FSN_PRIO) && FSN_PRIO == 1
    __builtin_amdgcn_s_setprio(1);
#elif defined(FSN_PRIO) && FSN_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#elif defined(FSN_PRIO) && FSN_PRIO == 4
    if (__builtin_amdgcn_readfirstlane(FSN_TIDX) & 256) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const Frag& b = ks < KS_ACT ? act[ks < KS_ACT ? ks : 0] : enc[ks >= KS_ACT ? ks - KS_ACT : 0];
      if constexpr (PREFETCH) {
        const int u = (tp * KS + ks) * 2;  // first of this k-step's two units; compile-time after unrolling
        // open the phase that starts two units from here (the one after this layer's last unit too)
        if (((u + kLead) % UPP == 0 && u + kLead <= TOTAL) || (u + kLead == TOTAL && TOTAL % UPP != 0)) st.open_next();
        if (u % UPP == 0) st.enter_phase();
        // A operands of the NEXT k-step (of the next GEMM / pass at the tail) are read from LDS in front of
        // this k-step's MFMAs, one scheduling region per k-step so that hipcc cannot sink the reads back to
        // their use.  Measured (bench_mlp, 8x256): single-pass modes 37.6 -> 42.4 % matrix-pipe busy; the
        // x3 modes lose (55.2 -> 52.8 %: 32 more live registers tip the wide layers into scratch), so they
        // keep the compiler's own read placement and rely on the partner wave of the SIMD.
        AFrag nxt[2];
        const int v = u + 2;
        const char* src = (v < TOTAL) ? ((v / UPP == u / UPP) ? st.c_base : st.n_base) + (v % UPP) * UB
                                      : st.n_base + (v - TOTAL) * UB;
        load_afrag<PREC>(src, nxt[0]);
        load_afrag<PREC>(src + UB, nxt[1]);
        unit_mfma_r<PREC>(ring.cur[0], b, acc0, cor0);
        unit_mfma_r<PREC>(ring.cur[1], b, acc1, cor1);
        ring.cur[0] = nxt[0];
        ring.cur[1] = nxt[1];
        __builtin_amdgcn_sched_group_barrier(0x100, X3 ? 4 : 2, 0);  // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, X3 ? 6 : 2, 0);  // MFMAs
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (kX3Pf1) {
        // x3 modes: A operands of the NEXT unit are read in front of this unit's three MFMAs (8 registers in flight)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const int u = (tp * KS + ks) * 2 + sub;
          if (((u + kLead) % UPP == 0 && u + kLead <= TOTAL) || (u + kLead == TOTAL && TOTAL % UPP != 0)) st.open_next();
          if (u % UPP == 0) st.enter_phase();
          const int v = u + 1;  // (two units ahead was measured too: no faster)
          const char* src = (v < TOTAL) ? ((v / UPP == u / UPP) ? st.c_base : st.n_base) + (v % UPP) * UB
                                        : st.n_base + (v - TOTAL) * UB;
          AFrag nxt;
          load_afrag<PREC>(src, nxt);
          if (sub == 0) unit_mfma_r<PREC>(ring.cur[0], b, acc0, cor0);
          else unit_mfma_r<PREC>(ring.cur[0], b, acc1, cor1);
          ring.cur[0] = nxt;
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS reads
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);  // MFMAs
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const int u = (tp * KS + ks) * 2 + sub;  // compile-time after unrolling
          // open the phase that starts kLead units from here (the one after this layer's last unit too)
          if (((u + kLead) % UPP == 0 && u + kLead <= TOTAL) || (u + kLead == TOTAL && TOTAL % UPP != 0)) st.open_next();
          if (u % UPP == 0) st.enter_phase();
          const char* ub = st.c_base + (u % UPP) * UB;
          if (sub == 0) unit_mfma<PREC>(ub, b, acc0, cor0);
          else unit_mfma<PREC>(ub, b, acc1, cor1);
        }
      }
    }
    pair_epilogue<PREC, NP_OUT, EPI>(net, tp, acc0, acc1, cor0, cor1, out, heads, g, hk);
  }
  if constexpr (F16 && kConverts) {
    if constexpr (HK::kLayerEnd) hk.layer_end(heads.rs.fmax);
    range_layer_end<kCheckSmall, kSmallThr>(heads.rs);
  }
}

template <int PREC, int NP_OUT, int KS_ACT, int KS_ENC, int EPI, int NACT, int NENC, int NOUT>
__device__ __forceinline__ void gemm_layer(WStream& st, const NetDev& net, int aux_bias, const Frag (&act)[NACT],
                                           const Frag (&enc)[NENC], Frag (&out)[NOUT], Heads& heads, ARing& ring,
                                           int g) {
  NoHook hk;
  gemm_layer<PREC, NP_OUT, KS_ACT, KS_ENC, EPI>(st, net, aux_bias, act, enc, out, heads, ring, g, hk);
}

// ---------------------------------------------------------------- two sample groups per wave (single-pass modes)
// 32 samples per wave as two 16-sample groups that share every A operand: one ds_read_b128 feeds two MFMAs, and a
// 16-KiB weight phase (one barrier, one round of LDS-DMA) covers 256 samples of the workgroup instead of 128.  The
// single-pass modes carry no low parts, so both groups' activations (2 x 64 registers in, 2 x 64 out) still fit two
// waves per SIMD.  Same blob, same unit order, same epilogue arithmetic per group as gemm_layer.
template <int PREC, int NP_OUT, int KS_ACT, int KS_ENC, int EPI, int NACT, int NENC, int NOUT>
__device__ __forceinline__ void gemm_layer2(WStream& st, const NetDev& net, int aux_bias, const Frag (&act0)[NACT],
                                            const Frag (&act1)[NACT], const Frag (&enc0)[NENC], const Frag (&enc1)[NENC],
                                            Frag (&out0)[NOUT], Frag (&out1)[NOUT], Heads& heads0, Heads& heads1,
                                            ARing& ring, int g) {
  constexpr bool F16 = PREC >= 2;
  static_assert((PREC & 1) == 1, "two groups per wave: single-pass modes only");
  constexpr int UPP = 16, UB = 1024;
  constexpr int KS = KS_ACT + KS_ENC;
  constexpr int TOTAL = 2 * NP_OUT * KS;
  static_assert(KS_ACT <= NACT && KS_ENC <= NENC, "operand arrays too small");
  const float* bias = net.aux + aux_bias;
  NoHook hk;
#pragma unroll
  for (int tp = 0; tp < NP_OUT; ++tp) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 32 * tp + 4 * g);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 32 * tp + 16 + 4 * g);
    f32x4 a00 = b0, a01 = b1, a10 = b0, a11 = b1;  // [group][tile]
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const Frag& x0 = ks < KS_ACT ? act0[ks < KS_ACT ? ks : 0] : enc0[ks >= KS_ACT ? ks - KS_ACT : 0];
      const Frag& x1 = ks < KS_ACT ? act1[ks < KS_ACT ? ks : 0] : enc1[ks >= KS_ACT ? ks - KS_ACT : 0];
      const int u = (tp * KS + ks) * 2;  // compile-time after unrolling
      if (((u + kLead) % UPP == 0 && u + kLead <= TOTAL) || (u + kLead == TOTAL && TOTAL % UPP != 0)) st.open_next();
      if (u % UPP == 0) st.enter_phase();
      // A operands of the next k-step in front of this k-step's four MFMAs (as the one-group prefetch path)
      AFrag nxt[2];
      const int v = u + 2;
      const char* src = (v < TOTAL) ? ((v / UPP == u / UPP) ? st.c_base : st.n_base) + (v % UPP) * UB
                                    : st.n_base + (v - TOTAL) * UB;
      load_afrag<PREC>(src, nxt[0]);
      load_afrag<PREC>(src + UB, nxt[1]);
      a00 = mfma16<F16>(ring.cur[0].hi, x0.hi, a00);
      a10 = mfma16<F16>(ring.cur[0].hi, x1.hi, a10);
      a01 = mfma16<F16>(ring.cur[1].hi, x0.hi, a01);
      a11 = mfma16<F16>(ring.cur[1].hi, x1.hi, a11);
      ring.cur[0] = nxt[0];
      ring.cur[1] = nxt[1];
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS reads
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    // (Issuing the conversion of pair tp-1 beside pair tp's MFMAs - one packed convert + ReLU per k-step, as inline
    // asm - gained 0.9 % and was NOT safe: hipcc does not see the XDL hazards of registers an asm statement touches,
    // and the bf16 density pass came out different from run to run.  tests/test_parity_fp64.py now renders the
    // same rays alone and inside a frame in this mode too.)
    pair_epilogue<PREC, NP_OUT, EPI>(net, tp, a00, a01, a00, a01, out0, heads0, g, hk);  // (single pass: no corrections)
    pair_epilogue<PREC, NP_OUT, EPI>(net, tp, a10, a11, a10, a11, out1, heads1, g, hk);
  }
}

// Whole network for one tile of 256 samples (two groups per wave); src0 / src1 supply this lane's sample of each group.
template <int NT, int PREC, bool FULL, class Src>
__device__ __forceinline__ void mlp_tile2(WStream& st, const NetDev& net, const Src& src0, const Src& src1, ARing& ring,
                                          float (&sigma)[2], float (&rgb)[2][3]) {
  constexpr int NA = NT;
  constexpr bool F16 = PREC >= 2;
  const int g = (FSN_TIDX >> 4) & 3;
  constexpr int D = 32 * NT;
  const int L = net.n_layers;
  const float* misc = net.aux + (L + 5) * D;
  Frag A0[NA], A1[NA], B0[NA], B1[NA];
  Frag none[1];
  Heads h0{0.f, {0.f, 0.f, 0.f}, {0u, 0u, 0u}}, h1{0.f, {0.f, 0.f, 0.f}, {0u, 0u, 0u}};
  auto enc_pos = [&](Frag (&p0)[kKsPos], Frag (&p1)[kKsPos]) __attribute__((always_inline)) {
    float x, y, z;
    src0.pos(x, y, z);
    encode<kKsPos, F16, false, false>(x, y, z, net.n_freqs_pos, misc + 4, net.pos_mask, g, p0);
    src1.pos(x, y, z);
    encode<kKsPos, F16, false, false>(x, y, z, net.n_freqs_pos, misc + 4, net.pos_mask, g, p1);
  };
  {
    Frag p0[kKsPos], p1[kKsPos];
    enc_pos(p0, p1);
    gemm_layer2<PREC, NT, 0, kKsPos, EPI_RELU_CVT>(st, net, 0, none, none, p0, p1, A0, A1, h0, h1, ring, g);
  }
#define FSN_HIDDEN2(EPI, I0, I1, O0, O1, LIDX)                                                              \
  do {                                                                                                      \
    if ((net.skip_mask >> ((LIDX)-1)) & 1u) {                                                               \
      Frag p0[kKsPos], p1[kKsPos];                                                                          \
      enc_pos(p0, p1);                                                                                      \
      gemm_layer2<PREC, NT, NA, kKsPos, EPI>(st, net, (LIDX)*D, I0, I1, p0, p1, O0, O1, h0, h1, ring, g);    \
    } else                                                                                                  \
      gemm_layer2<PREC, NT, NA, 0, EPI>(st, net, (LIDX)*D, I0, I1, none, none, O0, O1, h0, h1, ring, g);    \
  } while (0)
  for (int l = 1; l <= L - 2; l += 2) {
    FSN_HIDDEN2(EPI_RELU_CVT, A0, A1, B0, B1, l);
    if (l + 1 <= L - 2) {
      FSN_HIDDEN2(EPI_RELU_CVT, B0, B1, A0, A1, l + 1);
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) { A0[i] = B0[i]; A1[i] = B1[i]; }
    }
  }
  FSN_HIDDEN2((FULL ? EPI_LAST_FULL : EPI_LAST_DENS), A0, A1, B0, B1, L - 1);
#undef FSN_HIDDEN2
  {
    float s0 = h0.sigma, s1 = h1.sigma;
    s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64);
    s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
    sigma[0] = s0 + misc[0];
    sigma[1] = s1 + misc[0];
  }
  if (FULL) {
    gemm_layer2<PREC, NT, NA, 0, EPI_CVT>(st, net, L * D, B0, B1, none, none, A0, A1, h0, h1, ring, g);
    Frag d0[kKsDir], d1[kKsDir];
    float x, y, z;
    src0.dir(x, y, z);
    encode<kKsDir, F16, false, false>(x, y, z, net.n_freqs_dir, misc + 20, net.dir_mask, g, d0);
    src1.dir(x, y, z);
    encode<kKsDir, F16, false, false>(x, y, z, net.n_freqs_dir, misc + 20, net.dir_mask, g, d1);
    gemm_layer2<PREC, NT / 2, NA, kKsDir, EPI_RGB>(st, net, (L + 1) * D, A0, A1, d0, d1, B0, B1, h0, h1, ring, g);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float z0 = h0.rgb[c], z1 = h1.rgb[c];
      z0 += __shfl_xor(z0, 16, 64); z1 += __shfl_xor(z1, 16, 64);
      z0 += __shfl_xor(z0, 32, 64); z1 += __shfl_xor(z1, 32, 64);
      rgb[0][c] = 1.0f / (1.0f + expf(-(z0 + misc[1 + c])));
      rgb[1][c] = 1.0f / (1.0f + expf(-(z1 + misc[1 + c])));
    }
  }
  if constexpr (F16) {
    // (single pass: one running maximum per group for the whole tile, no per-layer scale check)
    asm("v_pk_max_u16 %0, %0, %1" : "+v"(h0.rs.fmax) : "v"(h1.rs.fmax));
    range_layer_end<false>(h0.rs);
    range_report(net.status, h0.rs);
  }
}

// A-operand pair primed for the very first GEMM of a kernel (after WStream::init opened phase 0)
template <int PREC, int NT = 0>
__device__ __forceinline__ void prime_ring(const WStream& st, ARing& ring) {
  constexpr int UB = (PREC & 1) == 0 ? 2048 : 1024;
  if (kKloopAsm && (PREC & 1) == 0 && NT == 8) {  // hand-scheduled x3 / x2 path: units 0 and 1 of the first GEMM
#pragma unroll
    for (int i = 0; i < kKD; ++i) load_afrag<PREC>(st.n_base + i * UB, ring.cur[i]);
#pragma unroll
    for (int i = kKD; i < (kNS > 3 ? kNS : 3); ++i) ring.cur[i] = ring.cur[0];
    return;
  }
#ifdef FSN_X3_PF1
  if ((PREC & 1) == 0) {
    load_afrag<PREC>(st.n_base, ring.cur[0]);
    return;
  }
#else
  if ((PREC & 1) == 0) return;  // the x3 modes read their operands in place
#endif
  load_afrag<PREC>(st.n_base, ring.cur[0]);
  load_afrag<PREC>(st.n_base + UB, ring.cur[1]);
  if ((PREC & 1) != 0) ring.cur[0].lo = ring.cur[1].lo = ring.cur[0].hi;  // single pass: no low parts
}

// ---------------------------------------------------------------- whole network, one tile
// `src` supplies this lane's sample on demand: src.pos(x,y,z) and src.dir(x,y,z) (the four lanes
// l, l+16, l+32, l+48 hold the same sample).  The direction is read only in front of the branch
// layer.  Outputs (valid in all lanes): sigma, and rgb when FULL.
// Saver of the inference kernels: nothing is kept.
struct NoSave {
  static constexpr bool kSave = false;
  using Hook = NoHook;
  __device__ __forceinline__ Hook hidden(int) const { return {}; }  // hidden layer l / connection (l = n_layers)
  __device__ __forceinline__ Hook branch() const { return {}; }
  __device__ __forceinline__ uint32_t* enc_pos(int) const { return nullptr; }
  __device__ __forceinline__ uint32_t* enc_dir(int) const { return nullptr; }
  __device__ __forceinline__ void layer_done(int, Hook&) const {}  // after GEMM l (kernel order) of a tile
};

template <int NT, int PREC, bool FULL, class Src, class SV>
__device__ __forceinline__ void mlp_tile(WStream& st, const NetDev& net, const Src& src, ARing& ring, float& sigma,
                                         float (&rgb)[3], const SV& sv) {
  constexpr int NA = NT;  // k-steps of 32 across the hidden width
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0, LS = prec_lo_scaled_k(PREC);
  const int g = (FSN_TIDX >> 4) & 3;
  constexpr int D = 32 * NT;
  const int L = net.n_layers;
  const float* misc = net.aux + (L + 5) * D;
  Frag A[NA], B[NA];
  Frag none[1];
  Heads heads{0.f, {0.f, 0.f, 0.f}, {0u, 0u, 0u}};
  {
    Frag pe[kKsPos];
    float px, py, pz;
    src.pos(px, py, pz);
    encode<kKsPos, F16, X3, SV::kSave, LS>(px, py, pz, net.n_freqs_pos, misc + 4, net.pos_mask, g, pe, sv.enc_pos(g));
    typename SV::Hook hk = sv.hidden(0);
    gemm_layer<PREC, NT, 0, kKsPos, EPI_RELU_CVT>(st, net, 0, none, pe, A, heads, ring, g, hk);
    sv.layer_done(0, hk);
  }
  // A wide (skip) layer re-encodes the position instead of keeping the 16 registers of `pe` alive across the layers
  // in between (same function of the same inputs: identical values; fused kernel: 360 -> 256 B of scratch, +1.5 %).
#define FSN_HIDDEN(EPI, IN, OUT, LIDX)                                                        \
  do {                                                                                        \
    typename SV::Hook hk = sv.hidden(LIDX);                                                   \
    if ((net.skip_mask >> ((LIDX)-1)) & 1u) {                                                 \
      Frag pe2[kKsPos];                                                                       \
      float qx, qy, qz;                                                                       \
      src.pos(qx, qy, qz);                                                                    \
      encode<kKsPos, F16, X3, false, LS>(qx, qy, qz, net.n_freqs_pos, misc + 4, net.pos_mask, g, pe2); \
      gemm_layer<PREC, NT, NA, kKsPos, EPI>(st, net, (LIDX)*D, IN, pe2, OUT, heads, ring, g, hk); \
    } else                                                                                    \
      gemm_layer<PREC, NT, NA, 0, EPI>(st, net, (LIDX)*D, IN, none, OUT, heads, ring, g, hk);     \
    sv.layer_done(LIDX, hk);                                                                  \
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
  {
    float sg = heads.sigma;
    sg += __shfl_xor(sg, 16, 64);
    sg += __shfl_xor(sg, 32, 64);
    sigma = sg + misc[0];
  }
  if (FULL) {
    // connection (no activation, models.py:130), then branch on [feat, dir_enc] (models.py:131-133)
    {
      typename SV::Hook hk = sv.hidden(L);
      gemm_layer<PREC, NT, NA, 0, EPI_CVT>(st, net, L * D, B, none, A, heads, ring, g, hk);
      sv.layer_done(L, hk);
    }
    Frag de[kKsDir];
    float dx, dy, dz;
    src.dir(dx, dy, dz);
    encode<kKsDir, F16, X3, SV::kSave, LS>(dx, dy, dz, net.n_freqs_dir, misc + 20, net.dir_mask, g, de, sv.enc_dir(g));
    {
      typename SV::Hook hk = sv.branch();
      gemm_layer<PREC, NT / 2, NA, kKsDir, EPI_RGB>(st, net, (L + 1) * D, A, de, B, heads, ring, g, hk);
      sv.layer_done(L + 1, hk);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float z = heads.rgb[c];
      z += __shfl_xor(z, 16, 64);
      z += __shfl_xor(z, 32, 64);
      z += misc[1 + c];
      rgb[c] = 1.0f / (1.0f + expf(-z));  // sigmoid (models.py:135)
    }
  }
  if constexpr (F16) {
    range_layer_end<false>(heads.rs);  // (single-pass modes keep one running maximum)
    range_report(net.status, heads.rs);
  }
}

template <int NT, int PREC, bool FULL, class Src>
__device__ __forceinline__ void mlp_tile(WStream& st, const NetDev& net, const Src& src, ARing& ring, float& sigma,
                                         float (&rgb)[3]) {
  mlp_tile<NT, PREC, FULL>(st, net, src, ring, sigma, rgb, NoSave{});
}

// Copy a blob's aux region and the two frequency masks into LDS (all threads of the workgroup;
// the caller synchronises afterwards) and describe the net.  lds: aux_floats + 96 floats.
__device__ __forceinline__ void load_net(const NetParams& p, const float* __restrict__ pos_mask_g,
                                         const float* __restrict__ dir_mask_g, float* lds, NetDev& net) {
  const f32x4* src = reinterpret_cast<const f32x4*>(p.blob + p.aux_off);
  f32x4* dst = reinterpret_cast<f32x4*>(lds);
  for (int i = FSN_TIDX; i < p.aux_floats / 4; i += blockDim.x) dst[i] = src[i];
  float* pm = lds + p.aux_floats;
  float* dm = pm + 64;
  const int npe = 3 * (1 + 2 * p.n_freqs_pos), nde = 3 * (1 + 2 * p.n_freqs_dir);
  for (int i = FSN_TIDX; i < 64; i += blockDim.x) pm[i] = (pos_mask_g && i < npe) ? pos_mask_g[i] : 1.0f;
  for (int i = FSN_TIDX; i < 32; i += blockDim.x) dm[i] = (dir_mask_g && i < nde) ? dir_mask_g[i] : 1.0f;
  net.aux = lds;
  net.pos_mask = pm;
  net.dir_mask = dm;
  net.n_layers = p.n_layers;
  net.skip_mask = p.skip_mask;
  net.n_freqs_pos = p.n_freqs_pos;
  net.n_freqs_dir = p.n_freqs_dir;
  net.status = p.status;
}

}  // namespace fsn

// train_fused.hip — the training step of NeRF.forward on the matrix cores.
//
//   k_train_fwd   : the inference MLP kernel (mlp_dev.hpp) with a saver hooked into every epilogue: the fp32
//                   encodings, post-activation hidden layers, connection output and branch output go to HBM.
//   k_train_bwd   : the dgrad chain in the same register-resident form.  d(out) -> rgb/sigma heads (VALU) ->
//                   branch^T -> connection^T -> layers L-1..1 transposed, each an MFMA GEMM whose A operands are
//                   the TRANSPOSED weights streamed through LDS exactly like the forward's; the ReLU mask comes
//                   from the saved activations; every pre-activation gradient dPre_l is stored for the wgrad.
//   k_wgrad       : dW_l = dPre_l . H_{l-1}^T, a GEMM whose contraction runs over ALL samples (K = millions,
//                   output 256x256): split-K over sample tiles, v_mfma_f32_32x32x16, one 256x256 (or smaller)
//                   fp32 accumulator block per workgroup in registers, partial blocks reduced in fixed order
//                   (deterministic) by k_wgrad_reduce, which also undoes the slot permutation of the encodings.
//   k_heads_wgrad : sigma / rgb head weights (1 and 3 output rows): HBM-bound row dot products.
//
// "Packed T-layout" of every saved matrix with R rows (features): tiles of 128 samples (one workgroup tile),
// [tile][chunk of 16 samples][R/2 pair-rows][16 samples][parts] x 32 bit (mlp_dev.hpp, t_layout_off).  A dword packs
// one 16-bit part of two neighbouring rows: part(row 2q, s) | part(row 2q+1, s) << 16; the x3 modes keep the HIGH and
// the LOW part of the value (in the mode's 16-bit format) side by side, the single-pass modes only have the first.
// These are the words the forward / backward epilogues form anyway for the next GEMM's B operand (a packed pair of
// neighbouring features of one sample), so saving costs the stores only - and the weight-gradient GEMM, whose
// contraction runs along the samples, has both rows' MFMA operands after four v_perm_b32 each, instead of splitting
// fp32 values itself.  Same bytes as fp32 in the x3 modes, half in the single-pass modes.
// Round 3 changed the order inside a tile twice (rounds 1-2: [plane][pair-row][128 samples]):
//   * chunk-major: a wave of the forward / backward kernels owns 16 samples, so its epilogue stores of neighbouring
//     pair-rows are neighbours in memory, and what the wgrad kernels consume per k-step - all rows of 16 samples - is
//     ONE contiguous run (16 KiB for 256 rows) instead of 64 bytes out of every 512;
//   * parts side by side: a lane stores the two parts of a pair as one 8-byte store (the savers cost 0.5-0.6 ms in
//     each of the two kernels, per instruction as much as per byte: DESIGN.md 6).
//
// Precision: as the forward, split 16-bit x 3 passes with fp32 accumulation.  In the fp16 modes d(out) is
// multiplied by a power-of-two `grad_scale` on entry (and the result divided by it in the reduce) so that the
// gradients sit in fp16's range; bf16 modes need none.
//
// reference: src/core/models.py:111-143 (forward), src/run-nerf.py:243-285 (loss.backward()).
// x3 modes: pinned one-unit-ahead LDS prefetch of the A operands (mlp_dev.hpp).  +6 % on these kernels; the fused
// render kernel keeps the compiler's own read placement (with its larger live state the pinned form spills more).
#define FSN_X3_PF1
#include <type_traits>

#include "mlp_dev.hpp"
#include "mlp_layout.hpp"
#include "train_internal.hpp"

namespace fsn {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kTC = kTileCols;

// ------------------------------------------------------------------ workspace
struct FusedLayout {
  int L, D, NT, d_pe, d_de, prec;
  int64_t T;  // sample tiles
  int64_t blob_f, blob_b;
  int64_t pe, de, h, h_stride, bo, dhead, dbo, dp, part, hpart, total;  // float offsets
  int64_t mask, mask_stride;  // ReLU sign bits: [L hidden layers + branch][tile][4 lane groups][128 samples] x 2 words
  int nsplit[4], nsplit_heads, n_bwd_units, nph_bwd;
  int64_t blob_f_bytes;
};

// The four wgrad launches (jobs of one shape each) and their split-K counts: every launch should put about one
// workgroup on every CU.
enum { WG_BIG = 0, WG_ENC = 1, WG_BR = 2, WG_BD = 3 };
static int n_jobs_of(const fsn_mlp_desc& d, int kind) {
  if (kind == WG_BIG) return d.n_layers;  // layers 1..L-1, connection
  if (kind == WG_ENC) {
    int n = 1;
    for (int l = 1; l < d.n_layers; ++l) n += (d.skip_mask >> (l - 1)) & 1u;
    return n;
  }
  return 1;
}
static int split_of(const fsn_mlp_desc& d, int kind, int cus, int64_t T) {
  int64_t ns = cus / n_jobs_of(d, kind);
  if (ns > T) ns = T;
  return ns < 1 ? 1 : (int)ns;
}
static int64_t part_floats(const fsn_mlp_desc& d, const int nsplit[4]) {
  const int64_t D = d.d_hidden;
  return (int64_t)n_jobs_of(d, WG_BIG) * nsplit[WG_BIG] * (D * D + D) + (int64_t)n_jobs_of(d, WG_ENC) * nsplit[WG_ENC] * (D * 64 + D) +
         (int64_t)nsplit[WG_BR] * ((D / 2) * D + D / 2) + (int64_t)nsplit[WG_BD] * ((D / 2) * 32 + D / 2);
}

static int make_fused_layout(const fsn_mlp_desc& d, int prec, int64_t n, FusedLayout& F, const char** why) {
  NetGeom G;
  const int rc = build_geom(d, prec, G, why);
  if (rc != FSN_OK) return rc;
  F.L = d.n_layers; F.D = d.d_hidden; F.NT = F.D / 32; F.prec = prec;
  F.d_pe = 3 * (1 + 2 * d.n_freqs_pos); F.d_de = 3 * (1 + 2 * d.n_freqs_dir);
  F.T = (n + kTC - 1) / kTC;
  int cus = fsn_device_cus();
  if (cus <= 0) cus = 256;  // sizing query without a device
  for (int k = 0; k < 4; ++k) F.nsplit[k] = split_of(d, k, cus, F.T);
  F.nsplit_heads = (int)(F.T < cus ? (F.T > 0 ? F.T : 1) : cus);
  const int NT = F.NT;
  F.n_bwd_units = NT * NT + 2 * NT * NT + (F.L - 1) * 2 * NT * NT;
  const int upp = units_per_phase(prec);
  F.nph_bwd = (F.n_bwd_units + upp - 1) / upp;
  F.blob_f_bytes = G.total_bytes;
  auto al = [](int64_t floats) { return (floats + 1023) / 1024 * 1024; };  // 4-KiB granules
  int64_t o = 0;
  F.blob_f = o; o += al(G.total_bytes / 4 + 1);
  F.blob_b = o; o += al((int64_t)F.nph_bwd * kPhaseBytes / 4);
  const int64_t T = F.T, D = F.D;
  F.pe = o; o += T * 64 * kTC;
  F.de = o; o += T * 32 * kTC;
  F.h_stride = T * D * kTC;
  F.h = o; o += (F.L + 1) * F.h_stride;  // H_0..H_{L-1}, connection output
  F.bo = o; o += T * (D / 2) * kTC;
  F.mask_stride = T * 4 * kTC * 2;
  F.mask = o; o += (F.L + 1) * F.mask_stride;
  F.dhead = o; o += T * 4 * kTC;
  F.dbo = o; o += T * (D / 2) * kTC;
  F.dp = o; o += (F.L + 1) * F.h_stride;  // dPre_0..dPre_{L-1}, d(connection output)
  F.part = o; o += al(part_floats(d, F.nsplit));
  F.hpart = o; o += al((int64_t)F.nsplit_heads * (D + 3 * (D / 2) + 4));
  F.total = o;
  return FSN_OK;
}

// saved activations / gradients are written once and read by a later kernel: streaming stores
#ifdef FSN_NO_NT
#define FSN_STREAM_STORE(v, p) (*(p) = (v))
#else
#define FSN_STREAM_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// Store the 16-bit parts of output pair tp (accumulator order: dword i of a part = rows 32 tp + 16 (i>>1) + 4 g +
// 2 (i&1), +1 = pair-row 16 tp + 8 (i>>1) + 2 g + (i&1)).  p: this lane's sample at pair-row 2g of its tile
// (t_layout_off).
template <bool X3>
__device__ __forceinline__ void store_pair_parts(uint32_t* p, int tp, const Frag& o) {
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  const u32x4 h = __builtin_bit_cast(u32x4, o.hi), l = __builtin_bit_cast(u32x4, o.lo);
#ifdef FSN_ABL_SAVE_NOLOADER  // timing experiment: the loader waves of the weight stream store nothing
  if (((threadIdx.x >> 6) ^ FSN_LOADER_XOR) < (uint32_t)kLoaders) return;
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pr = 16 * tp + 8 * (i >> 1) + (i & 1);
#ifdef FSN_ABL_SAVE_NONE  // timing experiment: no saved parts at all
    continue;
#endif
    if (X3) FSN_STREAM_STORE(((u32x2){h[i], l[i]}), reinterpret_cast<u32x2*>(p + pr * kTRow * 2));
    else FSN_STREAM_STORE(h[i], p + pr * kTRow);
  }
}

// ------------------------------------------------------------------ forward with saved activations
struct FwdSaver {
  static constexpr bool kSave = true;
  // Stores the 16-bit parts (packed T-layout) and, for ReLU layers, the sign bits: bit 8 tp + j of this lane's 64-bit
  // word (byte tp, bit j) <=> element j of output pair tp is > 0.  The dgrad chain reads only the bits.
  struct Hook {
    static constexpr bool kZeroInit = false;
    static constexpr bool kPacked = true;
    static constexpr bool kLayerEnd = false;
    uint32_t* p;    // this lane's sample at pair-row 2g of the layer's tile
    uint8_t* mk;    // this lane's 8 mask bytes (one per output pair), or null (layer without activation)
#ifdef FSN_MASK_WORDS  // experiment (measured: no change, 9.86 ms per step either way): four pairs' bytes gathered in a
    // register, one 4-byte store per four pairs instead of four byte stores
    uint32_t acc = 0u;
    int last = 7;   // index of the layer's last output pair
#endif
    __device__ __forceinline__ void pre(int) {}
    __device__ __forceinline__ void post(int tp, float (&v)[8]) {
#ifdef FSN_ABL_SAVE_NOLOADER
      if (((threadIdx.x >> 6) ^ FSN_LOADER_XOR) < (uint32_t)kLoaders) return;
#endif
      if (mk) {
        uint32_t b = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) b |= (v[j] > 0.f ? 1u : 0u) << j;
#ifdef FSN_MASK_WORDS
        acc |= b << (8 * (tp & 3));
        if ((tp & 3) == 3 || tp == last) {
          if ((tp & 3) == 3) *reinterpret_cast<uint32_t*>(mk + (tp & ~3)) = acc;
          else for (int q = 0; q <= (tp & 3); ++q) mk[(tp & ~3) + q] = (uint8_t)(acc >> (8 * q));
          acc = 0u;
        }
#else
        mk[tp] = (uint8_t)b;  // byte stores: accumulating the 64-bit word in registers tips the x3 modes into scratch
#endif
      }
    }
    template <bool X3>
    __device__ __forceinline__ void store(int tp, const Frag& o) { store_pair_parts<X3>(p, tp, o); }
  };
  uint32_t* h0;  // this lane's column, pair-row 2g, of H_0's tile
  int64_t hstride;
  uint32_t *bo, *pe, *de;
  uint32_t* mk0;  // this lane's mask words of layer 0
  int64_t mstride;
  int n_layers;
  __device__ __forceinline__ Hook hidden(int l) const {
    return Hook{h0 + l * hstride, l < n_layers ? reinterpret_cast<uint8_t*>(mk0 + l * mstride) : nullptr};
  }
  __device__ __forceinline__ Hook branch() const {
    Hook h{bo, reinterpret_cast<uint8_t*>(mk0 + n_layers * mstride)};
#ifdef FSN_MASK_WORDS
    h.last = branch_pairs - 1;
#endif
    return h;
  }
  int branch_pairs;  // output pairs of the branch layer (D / 64)
  __device__ __forceinline__ uint32_t* enc_pos(int) const { return pe; }
  __device__ __forceinline__ uint32_t* enc_dir(int) const { return de; }
  __device__ __forceinline__ void layer_done(int, Hook&) const {}
};

struct TrainFwdArgs {
  NetParams net;
  const float *x, *dirs, *pos_mask, *dir_mask;
  int64_t n;
  float* out;
  float* ws;
  int64_t off_h, h_stride, off_bo, off_pe, off_de, off_mask, mask_stride;
  int32_t D;
  // ray form (fsn_nerf_train_fwd_rays): sample s = midpoint of [t0[s], t1[s]) on ray ri[s] (x / dirs are not read)
  const float *rays_o, *rays_d, *t0, *t1;
  const int64_t* ri;
};

struct TileSrcT {
  const float* p;
  __device__ __forceinline__ void pos(float& x, float& y, float& z) const { x = p[0]; y = p[1]; z = p[2]; }
  __device__ __forceinline__ void dir(float& x, float& y, float& z) const { x = p[3]; y = p[4]; z = p[5]; }
};

template <int NT, int PREC>
__global__ __launch_bounds__(kThreads) void k_train_fwd(TrainFwdArgs a) {
  __shared__ __attribute__((aligned(1024))) char smem[kRingBytes + (kAuxCapFloats + 96) * 4 + 128 * 6 * 4];
  float* aux_lds = reinterpret_cast<float*>(smem + kRingBytes);
  float* in_lds = aux_lds + kAuxCapFloats + 96;
  NetDev net;
  load_net(a.net, a.pos_mask, a.dir_mask, aux_lds, net);
  __syncthreads();
  constexpr int D = 32 * NT;
  const int64_t ntiles = (a.n + kTC - 1) / kTC;
  WStream st;
  st.init(smem, nullptr, 0, 0, a.net.blob + a.net.stream_off, (uint32_t)a.net.nph_full, 1);
  ARing ring;
  prime_ring<PREC>(st, ring);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // (thread id laundered per tile: lane-derived addresses are recomputed, not hoisted and spilled - render.hip)
    int t_ = threadIdx.x;
#ifndef FSN_NO_LAUNDER_TID
    asm volatile("" : "+v"(t_));
#endif
    const int wave = t_ >> 6, lane = t_ & 63, g = lane >> 4;
    const int col = wave * 16 + (lane & 15);
    const int64_t s = tile * kTC + col;
    const int64_t sc = s < a.n ? s : a.n - 1;
    if (lane < 16) {
      float* q = in_lds + col * 6;
      if (a.ri) {  // x = o + d (t0 + t1) / 2, the reference's operation order (rendering.py:77-79); dirs = d
        const int64_t r = a.ri[sc];
        const float tm = a.t0[sc] + a.t1[sc];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float dc = a.rays_d[3 * r + c];
          q[c] = a.rays_o[3 * r + c] + dc * tm / 2.0f;
          q[3 + c] = dc;
        }
      } else {
        q[0] = a.x[3 * sc]; q[1] = a.x[3 * sc + 1]; q[2] = a.x[3 * sc + 2];
        q[3] = a.dirs[3 * sc]; q[4] = a.dirs[3 * sc + 1]; q[5] = a.dirs[3 * sc + 2];
      }
    }
    __builtin_amdgcn_wave_barrier();
    const TileSrcT src{in_lds + col * 6};
    FwdSaver sv;
    uint32_t* wsu = reinterpret_cast<uint32_t*>(a.ws);
    constexpr int NPL = (PREC & 1) == 0 ? 2 : 1;
    sv.h0 = wsu + a.off_h + tile * D * kTC + t_layout_off(NPL, D / 2, 2 * g, col);
    sv.hstride = a.h_stride;
    sv.bo = wsu + a.off_bo + tile * (D / 2) * kTC + t_layout_off(NPL, D / 4, 2 * g, col);
    sv.pe = wsu + a.off_pe + tile * 64 * kTC + t_layout_off(NPL, 32, 4 * g, col);
    sv.de = wsu + a.off_de + tile * 32 * kTC + t_layout_off(NPL, 16, 4 * g, col);
    sv.mk0 = reinterpret_cast<uint32_t*>(a.ws + a.off_mask) + ((tile * 4 + g) * kTC + col) * 2;
    sv.mstride = a.mask_stride;
    sv.n_layers = net.n_layers;
    sv.branch_pairs = D / 64;
    float sigma, rgb[3] = {0.f, 0.f, 0.f};
    mlp_tile<NT, PREC, true>(st, net, src, ring, sigma, rgb, sv);
    if (lane < 16 && s < a.n) {
      f32x4 o = {rgb[0], rgb[1], rgb[2], sigma};
      *reinterpret_cast<f32x4*>(a.out + 4 * s) = o;
    }
  }
  st.drain();
}

// ------------------------------------------------------------------ backward blob (transposed weights)
struct BwdPackArgs {
  const float* W[kMaxLayers + 2];
  int32_t ld[kMaxLayers + 2], ks[kMaxLayers + 2], unit0[kMaxLayers + 2];
  int32_t n_gemm, units_total, prec, NT;
};

// A operand element (row i = input feature of the layer, k = o = output feature) = W[o][i]; the k order inside a
// k-step is the accumulator-as-operand order of mlp_layout.hpp (o = 32 ks + 16 (j>>2) + 4 g + (j&3)).
__global__ void k_pack_bwd(BwdPackArgs a, char* __restrict__ stream, int64_t n_pieces) {
  const int64_t piece = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (piece >= n_pieces) return;
  const int ub = unit_bytes(a.prec), ppu = ub / 16;
  const int unit = (int)(piece / ppu);
  const int rem = (int)(piece - (int64_t)unit * ppu);
  const int part = rem >> 6, lane = rem & 63;
  uint16_t o8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (unit < a.units_total) {
    int gi = 0;
    while (gi + 1 < a.n_gemm && a.unit0[gi + 1] <= unit) ++gi;
    const int lu = unit - a.unit0[gi];
    const int sub = lu & 1, kst = a.ks[gi];
    const int t = (lu >> 1) / kst, ks = (lu >> 1) - t * kst;
    const int r = lane & 15, grp = lane >> 4;
    const int row = 32 * t + 16 * sub + r;
    const float* W = a.W[gi];
    const int ld = a.ld[gi];
    const bool f16 = prec_is_f16(a.prec);
    for (int j = 0; j < 8; ++j) {
      const int oc = 32 * ks + 16 * (j >> 2) + 4 * grp + (j & 3);
      const float w = W[(int64_t)oc * ld + row];
      const uint16_t hi = half_rne(w, f16);
      o8[j] = part == 0 ? hi : half_rne((w - half_to_f32(hi, f16)) * lo_scale(a.prec), f16);
    }
  }
  uint4 v;
  v.x = o8[0] | ((uint32_t)o8[1] << 16);
  v.y = o8[2] | ((uint32_t)o8[3] << 16);
  v.z = o8[4] | ((uint32_t)o8[5] << 16);
  v.w = o8[6] | ((uint32_t)o8[7] << 16);
  *reinterpret_cast<uint4*>(stream + piece * 16) = v;
}

// ------------------------------------------------------------------ backward chain (dgrad)
// Per-stage gradient scales (fp16 modes).  One global power of two (`grad_scale`, from max |d out|) puts the LARGEST
// gradient of the chain into fp16's range; the pre-activation gradients of earlier layers can be orders of magnitude
// smaller (a network shrunk by the weight-norm regulariser loses ~2 decades per layer), and below 2^-14 the high parts
// are fp16 subnormals.  So every stored stage (dBo, d feat, dPre_{L-1} .. dPre_0) carries its own power-of-two factor
// bs[stage] on top of grad_scale: the chain multiplies by bs[next] / bs[this] (exact) where it hands a gradient on,
// the reduce kernels divide a stage's weight gradients by grad_scale * bs[stage].  The factors are DELAYED: each
// backward measures every stage's largest stored |high part| (atomic maximum of the fp16 bit patterns) and
// k_bwd_rescale moves the factor so that the next call's maximum lands at 2^5 (core/models.py keeps the two small
// device arrays per model and calibrates them with one extra backward on first use).
// (the maxima are collected in LDS - one ds_max per wave, stage and tile - and go to memory once, when the workgroup
// has drained its weight stream: a global atomic inside the tile loop is a VMEM operation among the hand-counted
// LDS-DMA loads, and made the kernel 2.3 x slower)
__device__ __forceinline__ void stage_amax(uint32_t* amax_lds, uint32_t fmax) {
  uint32_t m = max(fmax & 0xffffu, fmax >> 16);
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, k, 64));
  if ((threadIdx.x & 63) == 0 && m) __hip_atomic_fetch_max(amax_lds, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct BwdStoreHook {
  static constexpr bool kZeroInit = true;
  static constexpr bool kPacked = true;
  static constexpr bool kLayerEnd = true;
  uint32_t* d;     // this lane's sample at pair-row 2g of the gradient's tile
  float r;         // bs[this stage] / bs[previous stage of the chain]
  uint32_t* amax;  // this stage's maximum word (LDS), or null
  __device__ __forceinline__ void pre(int) {}
  __device__ __forceinline__ void post(int, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= r;
  }
  template <bool X3>
  __device__ __forceinline__ void store(int tp, const Frag& o) { store_pair_parts<X3>(d, tp, o); }
  __device__ __forceinline__ void layer_end(uint32_t fmax) { if (amax) stage_amax(amax, fmax); }
};

template <bool ADD_SIGMA>
struct BwdMaskHook {
  static constexpr bool kZeroInit = true;
  static constexpr bool kPacked = true;
  static constexpr bool kLayerEnd = true;
  uint32_t b0, b1;    // sign bits of the layer whose pre-activation gradient this is (FwdSaver::Hook)
  uint32_t* d;        // dPre destination (as BwdStoreHook)
  float r;            // bs[this stage] / bs[previous stage of the chain]
  uint32_t* amax;     // this stage's maximum word (LDS), or null
  float dsig;         // d sigma of this lane's sample x bs[this stage]
  const float* wsig;  // LDS: w_sigma + 4g
  __device__ __forceinline__ void pre(int) {}
  __device__ __forceinline__ void post(int tp, float (&v)[8]) {
    if (ADD_SIGMA) {  // sigma = w_sigma . h_{L-1} + b (models.py:127)
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wsig + 32 * tp);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(wsig + 32 * tp + 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = v[j] * r + dsig * w0[j];
        v[4 + j] = v[4 + j] * r + dsig * w1[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= r;
    }
    const uint32_t bits = (tp < 4 ? b0 : b1) >> (8 * (tp & 3));
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ((bits >> j) & 1u) ? v[j] : 0.f;
  }
  template <bool X3>
  __device__ __forceinline__ void store(int tp, const Frag& o) { store_pair_parts<X3>(d, tp, o); }
  __device__ __forceinline__ void layer_end(uint32_t fmax) { if (amax) stage_amax(amax, fmax); }
};

struct TrainBwdArgs {
  NetParams net;       // forward blob: aux region (w_sigma, w_rgb)
  const char* bstream;  // transposed-weight stream
  int32_t nph_bwd;
  int64_t n;
  const float *out, *d_out, *scale;
  float* ws;
  int64_t off_h, h_stride, off_bo, off_dhead, off_dbo, off_dp, off_mask, mask_stride;
  const float* bscale;  // per-stage factors bs[0..L] (dPre_0..dPre_{L-1}, d feat), bs[L+1] (dBo); null: all 1
  uint32_t* bamax;      // per-stage maxima of the stored |high parts| (fp16 bit patterns), or null
};

template <int NT, int PREC>
__global__ __launch_bounds__(kThreads) void k_train_bwd(TrainBwdArgs a) {
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
  __shared__ __attribute__((aligned(1024))) char smem[kRingBytes + (kAuxCapFloats + 96 + 2 * (kMaxLayers + 2)) * 4];
  float* aux_lds = reinterpret_cast<float*>(smem + kRingBytes);
  float* bs_lds = aux_lds + kAuxCapFloats + 96;                                // per-stage factors
  uint32_t* am_lds = reinterpret_cast<uint32_t*>(bs_lds + (kMaxLayers + 2));  // per-stage maxima of this workgroup
  NetDev net;
  load_net(a.net, nullptr, nullptr, aux_lds, net);
  if (threadIdx.x < kMaxLayers + 2) {
    bs_lds[threadIdx.x] = (F16 && a.bscale && (int)threadIdx.x < a.net.n_layers + 2) ? a.bscale[threadIdx.x] : 1.0f;
    am_lds[threadIdx.x] = 0u;
  }
  __syncthreads();
  constexpr int D = 32 * NT;
  const int L = net.n_layers;
  const int64_t ntiles = (a.n + kTC - 1) / kTC;
  WStream st;
  st.init(smem, nullptr, 0, 0, a.bstream, (uint32_t)a.nph_bwd, 1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4;
  const int col = wave * 16 + (lane & 15);
  const float scale = a.scale ? a.scale[0] : 1.0f;
  ARing ring;
  prime_ring<PREC>(st, ring);
  Heads heads{0.f, {0.f, 0.f, 0.f}, {0u, 0u, 0u}};
  Frag none[1];
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t s = tile * kTC + col;
    const bool valid = s < a.n;
    f32x4 o4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      o4 = *reinterpret_cast<const f32x4*>(a.out + 4 * s);
      d4 = *reinterpret_cast<const f32x4*>(a.d_out + 4 * s);
    }
    // rgb = sigmoid(z): dz = d rgb * rgb (1 - rgb) (models.py:135); sigma head is linear (models.py:127)
    float dz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) dz[c] = (d4[c] * scale) * o4[c] * (1.0f - o4[c]);
    const float dsig = d4[3] * scale;
    if (lane < 16) {
      float* dh = a.ws + a.off_dhead + tile * 4 * kTC + col;
      dh[0] = dz[0]; dh[kTC] = dz[1]; dh[2 * kTC] = dz[2]; dh[3 * kTC] = dsig;
    }
    uint32_t* wsu = reinterpret_cast<uint32_t*>(a.ws);
    constexpr int NPL = X3 ? 2 : 1;
    const int64_t lane_off = tile * D * kTC + t_layout_off(NPL, D / 2, 2 * g, col);          // pair-row 2g of a D-row tile
    const int64_t lane_off_h = tile * (D / 2) * kTC + t_layout_off(NPL, D / 4, 2 * g, col);  // ... of a D/2-row tile
    Frag A[NT], B[NT];
    const uint32_t* mk = reinterpret_cast<const uint32_t*>(a.ws + a.off_mask) + ((tile * 4 + g) * kTC + col) * 2;
    // per-stage factors (wave-uniform scalar loads; all 1 in the bf16 modes / without the arrays)
    auto bs = [&](int i) { return bs_lds[i]; };
    auto am = [&](int i) { return (F16 && a.bamax) ? am_lds + i : nullptr; };
    {  // branch output: d Bo = W_rgb^T dz, masked by Bo > 0 (models.py:133-134), VALU
      const uint32_t bbits = mk[L * a.mask_stride];
      uint32_t* dbo = wsu + a.off_dbo + lane_off_h;
      const float* wr = net.aux + (L + 3) * D + 4 * g;
      const float s_bo = bs(L + 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) dz[c] *= s_bo;  // (dhead above holds the unscaled dz)
      uint32_t fm = 0u;
#pragma unroll
      for (int ks = 0; ks < NT / 2; ++ks) {
        float v[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + 32 * ks + 16 * hh);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + (D / 2) + 32 * ks + 16 * hh);
          const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 2 * (D / 2) + 32 * ks + 16 * hh);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float val = (dz[0] * w0[j] + dz[1] * w1[j]) + dz[2] * w2[j];
            v[4 * hh + j] = ((bbits >> (8 * ks + 4 * hh + j)) & 1u) ? val : 0.f;
          }
        }
        split_store<F16, X3>(v, B[ks]);
        if constexpr (F16) range_track<true>(fm, B[ks].hi);
        store_pair_parts<X3>(dbo, ks, B[ks]);
      }
      if constexpr (F16) {
        if (uint32_t* w = am(L + 1)) stage_amax(w, fm);
        asm("v_pk_max_u16 %0, %0, %1" : "+v"(heads.rs.fall) : "v"(fm));  // (an overflow here is reported like any other)
      }
    }
    {  // d feat = W_branch[:, :D]^T dBo  -> "dPre" of the connection (no activation, models.py:130)
      BwdStoreHook hk{wsu + a.off_dp + L * a.h_stride + lane_off, bs(L) / bs(L + 1), am(L)};
      gemm_layer<PREC, NT, NT / 2, 0, EPI_CVT>(st, net, 0, B, none, A, heads, ring, g, hk);
    }
    {  // d h_{L-1} = W_conn^T d feat + d sigma w_sigma, masked by h_{L-1} > 0
      BwdMaskHook<true> hk;
      hk.b0 = mk[(L - 1) * a.mask_stride];
      hk.b1 = mk[(L - 1) * a.mask_stride + 1];
      hk.d = wsu + a.off_dp + (L - 1) * a.h_stride + lane_off;
      hk.r = bs(L - 1) / bs(L);
      hk.amax = am(L - 1);
      hk.dsig = dsig * bs(L - 1);
      hk.wsig = net.aux + (L + 2) * D + 4 * g;
      gemm_layer<PREC, NT, NT, 0, EPI_CVT>(st, net, 0, A, none, B, heads, ring, g, hk);
    }
    // B = dPre_{l}; layers l = L-1 .. 1: dPre_{l-1} = (W_l[:, :D]^T dPre_l) * (h_{l-1} > 0)
    for (int l = L - 1; l >= 1; l -= 2) {
      {
        BwdMaskHook<false> hk;
        hk.b0 = mk[(l - 1) * a.mask_stride];
        hk.b1 = mk[(l - 1) * a.mask_stride + 1];
        hk.d = wsu + a.off_dp + (l - 1) * a.h_stride + lane_off;
        hk.r = bs(l - 1) / bs(l);
        hk.amax = am(l - 1);
        gemm_layer<PREC, NT, NT, 0, EPI_CVT>(st, net, 0, B, none, A, heads, ring, g, hk);
      }
      if (l - 1 >= 1) {
        BwdMaskHook<false> hk;
        hk.b0 = mk[(l - 2) * a.mask_stride];
        hk.b1 = mk[(l - 2) * a.mask_stride + 1];
        hk.d = wsu + a.off_dp + (l - 2) * a.h_stride + lane_off;
        hk.r = bs(l - 2) / bs(l - 1);
        hk.amax = am(l - 2);
        gemm_layer<PREC, NT, NT, 0, EPI_CVT>(st, net, 0, A, none, B, heads, ring, g, hk);
      }
    }
    if constexpr (F16) {  // a scaled gradient left the fp16 range
      range_layer_end<false>(heads.rs);
      // with per-stage scales an overflowing gradient is an event of the delayed scaling (FSN_STATUS_GRAD_RANGE: the
      // step is skipped, k_bwd_rescale lowers the stage), without them a range failure like the forward's
      range_report(net.status, heads.rs, a.bscale ? FSN_STATUS_GRAD_RANGE : FSN_STATUS_FP16_RANGE);
    }
  }
  st.drain();
  if constexpr (F16) {
    if (a.bamax) {
      __syncthreads();
      if ((int)threadIdx.x < L + 2 && am_lds[threadIdx.x]) atomicMax(a.bamax + threadIdx.x, am_lds[threadIdx.x]);
    }
  }
}

// ------------------------------------------------------------------ wgrad
struct WgJob {
  const uint32_t* A;  // dPre, packed T-layout with a_rows rows
  const uint32_t* B;  // layer input, packed T-layout with b_rows rows
  float* part;        // [nsplit][a_rows][b_rows]
  float* bpart;       // [nsplit][a_rows] row sums of A (bias gradient)
  int32_t b_rows, a_rows;
};
constexpr int kMaxJobs = kMaxLayers + 2;
struct WgArgs {
  WgJob job[kMaxJobs];
  int64_t T;
  int32_t nsplit;
};

template <bool F16>
__device__ __forceinline__ f32x16 mfma32(const s16x8& a, const s16x8& b, const f32x16& c) {
#ifdef FSN_WGRAD_NOMFMA  // timing experiment: the memory side of k_wgrad by itself (one VALU op keeps the operands live)
  f32x16 r = c;
  r[0] += (float)a[0] * (float)b[0];
  return r;
#endif
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The savers store the fp16 modes' low parts scaled by 2^11 (mlp_layout.hpp, kLoScaleF16).  This kernel keeps ONE
// accumulator block per workgroup (128 of its 256 registers), so it cannot sum the correction products separately:
// it multiplies the low-part dwords by 2^-11 as packed fp16 on load (exact unless the result is subnormal, in which
// case it is the value rounds 1-2 stored).  Gradients are asserted to 2e-4 of a tensor's largest entry, far above that.
template <bool F16X3>
__device__ __forceinline__ u32x4 unscale_lo(u32x4 w) {
#ifdef FSN_WGRAD_NOUNZIP
  return w;
#endif
  if constexpr (F16X3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_pk_mul_f16 %0, %0, %1" : "+v"(w[i]) : "s"(0x10001000u));  // 2^-11 | 2^-11
  }
  return w;
}

// the two rows of a pair-row out of 8 packed dwords (8 samples): low halves -> row 2q, high halves -> row 2q+1
__device__ __forceinline__ void unzip_rows(const u32x4& d0, const u32x4& d1, s16x8& even, s16x8& odd) {
#ifdef FSN_WGRAD_NOUNZIP  // timing experiment: the k_wgrad stream without its VALU work (results are garbage)
  even = __builtin_bit_cast(s16x8, d0);
  odd = __builtin_bit_cast(s16x8, d1);
  return;
#endif
  u32x4 e, o;
  e[0] = __builtin_amdgcn_perm(d0[1], d0[0], 0x05040100u); o[0] = __builtin_amdgcn_perm(d0[1], d0[0], 0x07060302u);
  e[1] = __builtin_amdgcn_perm(d0[3], d0[2], 0x05040100u); o[1] = __builtin_amdgcn_perm(d0[3], d0[2], 0x07060302u);
  e[2] = __builtin_amdgcn_perm(d1[1], d1[0], 0x05040100u); o[2] = __builtin_amdgcn_perm(d1[1], d1[0], 0x07060302u);
  e[3] = __builtin_amdgcn_perm(d1[3], d1[2], 0x05040100u); o[3] = __builtin_amdgcn_perm(d1[3], d1[2], 0x07060302u);
  even = __builtin_bit_cast(s16x8, e);
  odd = __builtin_bit_cast(s16x8, o);
}

// sum of the 8 16-bit elements of a fragment, accumulated in fp32 (v_dot2c with ones)
template <bool F16>
__device__ __forceinline__ float dot2_ones(uint32_t w, float acc) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  if (F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w), h2{(_Float16)1.0f, (_Float16)1.0f}, acc, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w), b2{(__bf16)1.0f, (__bf16)1.0f}, acc, false);
}
template <bool F16>
__device__ __forceinline__ float frag_sum(const s16x8& f, float acc) {
  const u32x4 w = __builtin_bit_cast(u32x4, f);
  acc = dot2_ones<F16>(w.x, acc);
  acc = dot2_ones<F16>(w.y, acc);
  acc = dot2_ones<F16>(w.z, acc);
  acc = dot2_ones<F16>(w.w, acc);
  return acc;
}

// Workgroup = 8 waves as MG row groups (64 rows of A each: two 32-row MFMA tiles) x CG = 8/MG column groups (BT
// 32-row tiles of B each); blockIdx.z selects the workgroup's block of A_ROWS = 64 MG rows of A.  The contraction runs
// over the samples in chunks of 16 (one MFMA k-step).
//
// This kernel is bound by HBM, not by the matrix pipe (12.9 GB per step at float32-grade storage; timing the memory
// side and the matrix side of the register-prefetch form of rounds 1-2 separately gave 2.66 ms and 1.97 ms of its
// 3.08 ms), so the operands travel HBM -> LDS by LDS-DMA (`global_load_lds_dwordx4`), never through
// registers: a ring of NSTAGE = 4 chunk slots keeps two to three chunks (64-96 KiB per CU) in flight.
//   * A chunk slot holds the RAW packed pair-rows of the workgroup's rows of A and B, one section per 64-row block: 32
//     pair-rows x 16 samples x parts = 2 KiB per part, contiguous in the T-layout, fetched as pieces of 1 KiB (one
//     LDS-DMA instruction each).  The LDS side of such an instruction is lane-linear by construction; the global side is
//     free, so the granules (16 bytes) of a piece are permuted on the way such that the consumers' ds_read_b128 touch
//     every bank quad once per 16-lane group (conflict-free).  x3 modes: a piece = 8 pair-rows x 8 granules (samples
//     2c, 2c+1 with both parts); consumer lane (kg, m) wants granules 4 kg .. 4 kg + 3 of pair-row m and finds granule
//     4 kg + r of piece m >> 3 at slot 16 r + (m & 7) + 8 (kg ^ (m >> 4)).  Single-pass modes: a piece = 16 pair-rows x
//     4 granules; granule 2 kg + r of piece m >> 4 at slot 32 r + 16 kg + (m & 15).
//   * Both halves of every dword are used by the lane that reads it: MFMA tile 0 of a 64-row block takes its EVEN
//     rows, tile 1 the ODD rows (A: the wave's two row tiles; B: tiles 2j and 2j+1 of the workgroup).  Eight v_perm
//     unzip two granules into the two tiles' operands.  B tile g therefore holds rows 64 (g>>1) + 2 q + (g&1).
//   * One workgroup barrier per chunk: before it every wave waits (counted vmcnt) for its own pieces of the NEXT
//     chunk, after it the slot of the chunk consumed LAST is refilled.  The raw granules of chunk c+1 are read and
//     unzipped beside the MFMAs of chunk c (software pipeline over registers), so the two waves of a SIMD, which the
//     barrier keeps in step, do not both stand in front of an LDS latency at the same time.
// No fp32 -> 16-bit conversion happens here: the savers stored the parts.  Bias gradient = row sums of A (v_dot2c).
template <bool TWO>
__device__ __forceinline__ void wg_dma(uint32_t voff, const void* gbase, uint32_t m0v) {
  // one or two 1-KiB pieces of a section (the immediate offset advances the global and the LDS address alike).
  // The scalar operands are wave-uniform by construction; readfirstlane pins them to SGPRs (an "s" INPUT operand may
  // otherwise arrive in a VGPR when the compiler formed the value with vector instructions).
  uint32_t keep;
  m0v = __builtin_amdgcn_readfirstlane(m0v);
  const uint64_t gb = (uint64_t)gbase;
  const uint64_t gs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)gb);
  if (TWO)
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(m0v), "s"(gs)
        : "memory");
  else
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(m0v), "s"(gs)
        : "memory");
}

template <int NW, int MG, int BT, int PREC>
__global__ __launch_bounds__(64 * NW, 2) void k_wgrad(WgArgs a) {  // (2 waves per SIMD: <= 256 registers)
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
  constexpr int NPL = X3 ? 2 : 1;  // parts per value: high (, low)
  constexpr int CG = NW / MG;
  constexpr int A_ROWS = 64 * MG;
  constexpr int NTB = CG * BT;          // 32-row MFMA tiles of B per workgroup
  constexpr int NBB = (NTB + 1) / 2;    // 64-row blocks of B
  constexpr int NBP = (BT + 1) / 2;     // blocks of B a wave reads
  constexpr int NQ = (MG + NBB) * NPL;  // 2-KiB halves of the sections (64-row blocks) of a chunk slot
  constexpr int SLOT = NQ * 2048;
  constexpr int NSTAGE = 4;
  constexpr int PAIRS = (NQ + NW - 1) / NW;  // halves a wave loads per chunk (two LDS-DMA instructions each)
  static_assert(NW == 8 && (BT == 1 || BT % 2 == 0) && PAIRS <= 3 && NSTAGE * SLOT <= 160 * 1024, "k_wgrad geometry");
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * SLOT];
  const WgJob jb = a.job[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave % MG, cg = wave / MG;
  const int m = lane & 31, kg = lane >> 5;
  const int b_rows = jb.b_rows, b_pairs = b_rows >> 1;
  const int nbb = (b_rows + 63) >> 6;  // blocks of B that exist (the last one may be half full: 32 direction rows)
  const int a_tot = jb.a_rows, rb = blockIdx.z;  // this workgroup's rows: [rb A_ROWS, (rb + 1) A_ROWS) of a_tot
  const int g0 = cg * BT;                        // the wave's first B tile
  const bool active = (g0 >> 1) < nbb;
  const bool want_bias = jb.bpart && cg == 0;
  const int64_t t0 = (int64_t)blockIdx.x * a.T / a.nsplit, t1 = (int64_t)(blockIdx.x + 1) * a.T / a.nsplit;
  const int64_t nchunk = (t1 - t0) * (kTC / 16);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  f32x16 acc[2][BT];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][bt][r] = 0.f;
  float bsum[2] = {0.f, 0.f};

  // ---- loader side: this wave's section halves q = wave + 8 j of every chunk
  const int nq = (MG + nbb) * NPL;
  const bool half_last = (b_pairs & 31) != 0;  // the last block of B holds 16 pair-rows: half a section
  auto pieces_of = [&](int q) {                // LDS-DMA instructions of half q (wave-uniform)
    if (half_last && q / NPL == MG + nbb - 1) return X3 ? (q % NPL == 0 ? 2 : 0) : 1;
    return 2;
  };
  int my_inst = 0;  // ... of this wave per chunk
#pragma unroll
  for (int j = 0; j < PAIRS; ++j) {
    const int q = wave + NW * j;
    if (q < nq) my_inst += pieces_of(q);
  }
  // granule of the 1-KiB piece this lane fetches (it lands at LDS slot `lane` of the piece); x3: the second half of a
  // section holds pieces 2, 3, whose slots swap the kg halves (voff ^ 64)
  const uint32_t voff = X3 ? (uint32_t)(((lane & 7) * 8 + ((lane >> 3) & 1) * 4 + (lane >> 4)) * 16)
                           : (uint32_t)(((lane & 15) * 4 + ((lane >> 4) & 1) * 2 + (lane >> 5)) * 16);
  auto issue = [&](int64_t cj) {
    const int64_t t = t0 + cj / (kTC / 16);
    const int64_t c16 = cj % (kTC / 16);
    const uint32_t slot = lds_base + (uint32_t)(cj % NSTAGE) * SLOT;
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
      const int q = wave + NW * j;
      if (q < nq) {
        const int sec = q / NPL, hf = q % NPL;
        const bool isA = sec < MG;
        const uint32_t* base = isA ? jb.A : jb.B;
        const int64_t pairs_tot = isA ? a_tot / 2 : b_pairs;
        const int64_t pair0 = isA ? rb * (A_ROWS / 2) + sec * 32 : (sec - MG) * 32;
        const uint32_t* g = base + t * 2 * pairs_tot * kTC + (c16 * pairs_tot + pair0) * (kTRow * NPL) + hf * 512;
        const uint32_t vo = voff ^ (uint32_t)(hf * 64);
        const int np = pieces_of(q);
        if (np == 2) wg_dma<true>(vo, g, slot + (uint32_t)q * 2048u);
        else if (np == 1) wg_dma<false>(vo, g, slot + (uint32_t)q * 2048u);
      }
    }
  };
  // own pieces of the oldest chunk in flight landed: vmcnt counts in issue order, NSTAGE - 2 younger chunks may stay
  auto wait_counted = [&]() {
    static_assert(NSTAGE == 4 && PAIRS <= 3, "the immediates below are 2 chunks x instructions per chunk");
    switch (my_inst) {
      case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: break;
    }
  };

  // this lane's first granule inside a section; the following ones RSTEP bytes apart
  constexpr int NG = 2 * NPL, RSTEP = X3 ? 256 : 512;
  const int lane_off = X3 ? (m >> 3) * 1024 + ((m & 7) + 8 * (kg ^ ((m >> 4) & 1))) * 16 : (m >> 4) * 1024 + (kg * 16 + (m & 15)) * 16;
  Frag af[2];    // current chunk: the wave's A rows, tile 0 (even rows) / tile 1 (odd rows)
  Frag bfr[BT];  // ... its B tiles
  s16x8 ahs[2];  // fp16 x3: A's high parts x 2^-11 (operand of the A-high x B-low product)
  u32x4 rawA[NG], rawB[NBP][NG];  // samples 8 kg .. + 8 of the lane's pair-row: NG granules per section
  // One step = [wait + barrier: chunk ci+1 landed, slot of chunk ci free] [refill that slot] [raw reads of chunk ci+1]
  // [MFMAs of chunk ci] [unzip chunk ci+1].  ACT / BIAS (wave-uniform) are compile-time inside the loops so that the
  // steady-state step is ONE basic block after its scalar head: the scheduler is then told to place the ~4.5 VALU
  // instructions per MFMA of the unzip (v_perm, the 2^-11 of the low parts, row sums) and the 12 reads BETWEEN the 24
  // MFMAs - left alone it emits the MFMAs back to back and the unzip after them, and since the barrier keeps the two
  // waves of a SIMD in step both then leave the matrix pipe idle together.
  auto reads = [&](int64_t cj, auto ACT) {
    const char* sl = lds + (int)(cj % NSTAGE) * SLOT + lane_off;
#pragma unroll
    for (int r = 0; r < NG; ++r) {
      rawA[r] = *reinterpret_cast<const u32x4*>(sl + rg * (2048 * NPL) + r * RSTEP);
      if constexpr (decltype(ACT)::value) {
#pragma unroll
        for (int jj = 0; jj < NBP; ++jj)
          rawB[jj][r] = *reinterpret_cast<const u32x4*>(sl + (MG + (g0 >> 1) + jj) * (2048 * NPL) + r * RSTEP);
      }
    }
  };
  // the operands of the even-row and the odd-row tile out of a section's granules
  // fp16 x3: the low parts arrive scaled by 2^11.  A's are unscaled here (A = the gradients, which the backward keeps
  // near 2^5: its low parts stay normal numbers).  B's - the forward's activations, of any size down to 2^-14 - are
  // NOT: their product takes the 2^-11 on the A side instead (ahs = A's high parts x 2^-11, at most 2^-6; what that
  // loses below 2^-14 is 2^-30 of the largest product), so a layer with activations of 1e-3 keeps its low parts' bits.
  auto unzip_section = [&](const u32x4 (&g)[NG], Frag& ev, Frag& od, auto IS_A) {
    constexpr bool kUnscale = F16 && decltype(IS_A)::value;
    if constexpr (X3) {  // granule r = samples 2r, 2r+1: (high, low, high, low) dwords
      u32x4 eh, oh, el, ol;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        eh[r] = __builtin_amdgcn_perm(g[r][2], g[r][0], 0x05040100u);
        oh[r] = __builtin_amdgcn_perm(g[r][2], g[r][0], 0x07060302u);
        el[r] = __builtin_amdgcn_perm(g[r][3], g[r][1], 0x05040100u);
        ol[r] = __builtin_amdgcn_perm(g[r][3], g[r][1], 0x07060302u);
      }
      ev.hi = __builtin_bit_cast(s16x8, eh); od.hi = __builtin_bit_cast(s16x8, oh);
      ev.lo = __builtin_bit_cast(s16x8, unscale_lo<kUnscale>(el)); od.lo = __builtin_bit_cast(s16x8, unscale_lo<kUnscale>(ol));
    } else {
      unzip_rows(g[0], g[1], ev.hi, od.hi);
      ev.lo = ev.hi; od.lo = od.hi;
    }
  };
  auto mfmas = [&](auto ACT, auto BIAS) {
    if constexpr (decltype(BIAS)::value) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        bsum[ti] = frag_sum<F16>(af[ti].hi, bsum[ti]);
        if (X3) bsum[ti] = frag_sum<F16>(af[ti].lo, bsum[ti]);
      }
    }
    if constexpr (decltype(ACT)::value) {
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          acc[ti][bt] = mfma32<F16>(af[ti].hi, bfr[bt].hi, acc[ti][bt]);
          if (X3) {
            // experiment switches (profiles/EXPERIMENTS_r4.md 11): bit 0 drops (dPre low) x (input high), bit 1 drops
            // (dPre high) x (input low); _BIG for the 256x256 jobs only, _ALL for every job
#ifndef FSN_EXP_WGRAD_DROP_BIG
#define FSN_EXP_WGRAD_DROP_BIG 0
#endif
#ifndef FSN_EXP_WGRAD_DROP_ALL
#define FSN_EXP_WGRAD_DROP_ALL 0
#endif
            constexpr int kDrop = FSN_EXP_WGRAD_DROP_ALL | ((MG == 4 && BT == 4) ? FSN_EXP_WGRAD_DROP_BIG : 0);
            if constexpr (!(kDrop & 1)) acc[ti][bt] = mfma32<F16>(af[ti].lo, bfr[bt].hi, acc[ti][bt]);
#ifndef FSN_WGRAD_NOBLO  // experiment: drop the (dPre high) x (input low) product
            if constexpr (!(kDrop & 2)) acc[ti][bt] = mfma32<F16>(F16 ? ahs[ti] : af[ti].hi, bfr[bt].lo, acc[ti][bt]);
#endif
          }
        }
    }
  };
  auto unzip = [&](auto ACT) {
    unzip_section(rawA, af[0], af[1], std::true_type{});
    if constexpr (F16 && X3) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) ahs[ti] = __builtin_bit_cast(s16x8, unscale_lo<true>(__builtin_bit_cast(u32x4, af[ti].hi)));
    }
    if constexpr (decltype(ACT)::value) {
#pragma unroll
      for (int jj = 0; jj < NBP; ++jj) {
        Frag ev, od;
        unzip_section(rawB[jj], ev, od, std::false_type{});
        if constexpr (BT == 1) {  // one tile per wave: the even or the odd rows of the block it shares with its neighbour
          const bool odd = g0 & 1;
          bfr[0].hi = odd ? od.hi : ev.hi;
          bfr[0].lo = odd ? od.lo : ev.lo;
        } else {
          bfr[2 * jj] = ev;
          bfr[2 * jj + 1] = od;
        }
      }
    }
  };
  auto run = [&](auto ACT, auto BIAS) {
    constexpr bool kAct = decltype(ACT)::value;
    auto edge_step = [&](int64_t ci) {  // first and last steps: any of the parts may be missing
      const bool more = ci + 1 < nchunk;
      if (more) {
        if (nchunk - 2 - ci >= NSTAGE - 2) wait_counted();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last chunks: nothing younger was issued
        asm volatile("s_barrier" ::: "memory");
        if (ci + NSTAGE < nchunk) issue(ci + NSTAGE);  // into the slot of chunk ci, which every wave has left
        reads(ci + 1, ACT);
      }
      if (ci >= 0) mfmas(ACT, BIAS);
      if (more) unzip(ACT);
    };
    const int64_t steady_end = nchunk - NSTAGE;  // steps [0, steady_end): every part present
    edge_step(-1);
    for (int64_t ci = 0; ci < steady_end; ++ci) {
      wait_counted();
      asm volatile("s_barrier" ::: "memory");
      issue(ci + NSTAGE);
      __builtin_amdgcn_sched_barrier(0);
      reads(ci + 1, ACT);
      mfmas(ACT, BIAS);
      unzip(ACT);
      if constexpr (kAct) {
        constexpr int NM = 2 * BT * (X3 ? 3 : 1);
        constexpr int NR = NG * (1 + NBP);
#pragma unroll
        for (int k = 0; k < NM; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // one MFMA
          if (k < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one raw read of the next chunk
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);              // its share of the VALU work
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int64_t ci = steady_end < 0 ? 0 : steady_end; ci < nchunk; ++ci) edge_step(ci);
  };
  using T1 = std::true_type;
  using T0 = std::false_type;
  for (int c = 0; c < NSTAGE - 1 && c < nchunk; ++c) issue(c);
  if (active) {
    if (want_bias) run(T1{}, T1{});
    else run(T1{}, T0{});
  } else {
    run(T0{}, T0{});
  }
  // partial block: C layout of the 32x32 tile: column = lane&31, tile row = (r&3) + 8 (r>>2) + 4 (lane>>5); tile ti's
  // row q is row 2 q + ti of the wave's 64-row block, B tile g's column q is row 64 (g>>1) + 2 q + (g&1) of B
  if (active) {
    float* part = jb.part + ((int64_t)blockIdx.x * a_tot + rb * A_ROWS) * b_rows;
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      const int g = g0 + bt;
      const int cc = 64 * (g >> 1) + 2 * m + (g & 1);
      if (cc < b_rows) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 64 * rg + 2 * ((r & 3) + 8 * (r >> 2) + 4 * kg) + ti;
            part[(int64_t)row * b_rows + cc] = acc[ti][bt][r];
          }
      }
    }
  }
  if (want_bias) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const float v = bsum[ti] + __shfl_xor(bsum[ti], 32, 64);
      if (kg == 0) jb.bpart[(int64_t)blockIdx.x * a_tot + rb * A_ROWS + 64 * rg + 2 * m + ti] = v;
    }
  }
}

struct RdJob {
  const float* part;
  const float* bpart;
  float* dW;
  float* db;
  int32_t a_rows, b_rows, ld, col0, mode, n_freqs;  // mode 0: column = col0 + c; 1/2: position/direction slots
  int32_t nsplit;
  int32_t stage;  // index of the job's gradient operand in the per-stage factors (RdArgs::bscale)
};
struct RdArgs {
  RdJob job[2 * kMaxLayers + 4];
  const float* scale;
  const uint32_t* status;  // fp16 modes: this call's range-guard word (null in the bf16 modes); bit 0 -> zero gradients
  int32_t accumulate;      // add to dW / db instead of overwriting them
  const float* bscale;     // per-stage factors of the backward chain (fp16 modes) or null
};

// Sum of the split-K partials of one element in a FIXED order (deterministic): eight independent running sums over
// p = i (mod 8), combined as a tree.  (One running sum = one load in flight per thread: the reduce kernels took 0.3 ms
// per step for 70 MB.)
__device__ __forceinline__ float sum_partials(const float* __restrict__ p0, int64_t stride, int nsplit) {
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int p = 0;
  for (; p + 8 <= nsplit; p += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += p0[(int64_t)(p + k) * stride];
  }
  for (int k = 0; p < nsplit; ++p, ++k) s[k] += p0[(int64_t)p * stride];
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

__global__ void k_wgrad_reduce(RdArgs a) {
  const RdJob jb = a.job[blockIdx.y];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const float inv = (a.scale ? 1.0f / a.scale[0] : 1.0f) / (a.bscale ? a.bscale[jb.stage] : 1.0f);  // (powers of two)
  // The forward / backward launch of THIS call reported values outside the fp16 range (bit 0 of the per-call word):
  // the sums are inf / NaN.  The gradients are zeroed ON THE DEVICE, without a host sync; fsn_adam_step skips the
  // update on the same word, and the host learns of it at its next (amortised) look and continues in bf16x3.
  const bool skip = a.status && (a.status[0] & (FSN_STATUS_FP16_RANGE | FSN_STATUS_GRAD_RANGE));
  const int tot = jb.a_rows * jb.b_rows;
  if (e < tot) {
    const int row = e / jb.b_rows, c = e - row * jb.b_rows;
    int colo;
    if (jb.mode == 0) {
      colo = jb.col0 + c;
    } else {
      const int slots = jb.mode == 1 ? 8 * kKsPos : 8 * kKsDir;
      const int ks = c >> 5, gq = (c >> 3) & 3, j = c & 7;
      const int f = enc_slot_feature(8 * ks + j, gq, jb.n_freqs, slots);
      colo = f < 0 ? -1 : jb.col0 + f;
    }
    if (colo >= 0) {
      const float sum = sum_partials(jb.part + e, tot, jb.nsplit);
      float* dst = jb.dW + (int64_t)row * jb.ld + colo;
      const float v = skip ? 0.f : sum * inv;
      *dst = a.accumulate ? *dst + v : v;
    }
  }
  if (jb.bpart && e < jb.a_rows) {
    const float sum = sum_partials(jb.bpart + e, jb.a_rows, jb.nsplit);
    const float v = skip ? 0.f : sum * inv;
    jb.db[e] = a.accumulate ? jb.db[e] + v : v;
  }
}

// ------------------------------------------------------------------ head weights
// dW_sigma[f] = sum_s dsigma_s h_{L-1}[f,s];  dW_rgb[c,f] = sum_s dz_c,s Bo[f,s];  biases = sums of dsigma / dz.
// Wave w owns 2 NT pair-rows of h_{L-1} and NT pair-rows of Bo (packed T-layout: value = high part + low part).
// Four lanes per pair-row, four samples each: the wave's pair-rows of one 16-sample chunk are one contiguous run of the
// T-layout (2 KiB for 16 pair-rows with both parts); with fewer pair-rows per wave the lane groups left over take the
// following chunks.
struct HeadsArgs {
  const uint32_t *H, *Bo;
  const float* dhead;
  float* hpart;  // [nsplit][D + 3 D/2 + 4]
  int64_t T;
  int32_t nsplit;
};

template <int NT, int PREC>
__global__ __launch_bounds__(kThreads) void k_heads_wgrad(HeadsArgs a) {
  constexpr bool F16 = PREC >= 2, X3 = (PREC & 1) == 0;
  constexpr int D = 32 * NT, PH = 2 * NT, PB = NT;     // pair-rows of H / Bo per wave
  constexpr int CH = 16 / PH, CB = 16 / PB;             // chunks the wave's lanes cover side by side
  constexpr int NCH = kTC / 16;
  static_assert(PH <= 16 && 16 % PH == 0 && 16 % PB == 0 && NCH % CB == 0, "k_heads_wgrad geometry");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t t0 = (int64_t)blockIdx.x * a.T / a.nsplit, t1 = (int64_t)(blockIdx.x + 1) * a.T / a.nsplit;
  const int s4 = (lane & 3) * 4, idx = lane >> 2;
  const int qh = idx % PH, ch = idx / PH, qb = idx % PB, cb = idx / PB;
  float accS[2] = {0.f, 0.f}, accR[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, accB[4] = {0.f, 0.f, 0.f, 0.f};
  // the two rows of a pair-row at four samples: v[row parity][sample]; p: first part of the first sample
  constexpr int NPL = X3 ? 2 : 1;
  auto load_pair = [&](const uint32_t* p, float (&v)[2][4]) {
    uint32_t h[4], l[4] = {0u, 0u, 0u, 0u};
    if constexpr (X3) {  // (high, low) side by side
      const u32x4 w0 = *reinterpret_cast<const u32x4*>(p), w1 = *reinterpret_cast<const u32x4*>(p + 4);
      h[0] = w0[0]; l[0] = w0[1]; h[1] = w0[2]; l[1] = w0[3];
      h[2] = w1[0]; l[2] = w1[1]; h[3] = w1[2]; l[3] = w1[3];
    } else {
      const u32x4 w0 = *reinterpret_cast<const u32x4*>(p);
      h[0] = w0[0]; h[1] = w0[1]; h[2] = w0[2]; h[3] = w0[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[0][i] = from_h<F16>((short)(h[i] & 0xffffu));
      v[1][i] = from_h<F16>((short)(h[i] >> 16));
      if (X3) {  // (fp16: the low parts are stored scaled by 2^11)
        constexpr float IK = F16 ? 1.0f / kLoScaleF16 : 1.0f;
        v[0][i] = __builtin_fmaf(from_h<F16>((short)(l[i] & 0xffffu)), IK, v[0][i]);
        v[1][i] = __builtin_fmaf(from_h<F16>((short)(l[i] >> 16)), IK, v[1][i]);
      }
    }
  };
  for (int64_t t = t0; t < t1; ++t) {
    const float* dht = a.dhead + t * 4 * kTC;
    if (wave == 0) {  // biases: sums of d sigma / d z over the tile's samples
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x2 d2 = *reinterpret_cast<const f32x2*>(dht + c * kTC + 2 * lane);
        accB[c] += d2[0] + d2[1];
      }
    }
    const uint32_t* hp = a.H + t * D * kTC + ((PH * wave + qh) * kTRow + s4) * NPL;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += CH) {
      const int c = c0 + ch;
      const f32x4 ds = *reinterpret_cast<const f32x4*>(dht + 3 * kTC + 16 * c + s4);
      float h[2][4];
      load_pair(hp + (int64_t)c * (D / 2) * kTRow * NPL, h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        accS[0] += ds[i] * h[0][i];
        accS[1] += ds[i] * h[1][i];
      }
    }
    const uint32_t* bp = a.Bo + t * (D / 2) * kTC + ((PB * wave + qb) * kTRow + s4) * NPL;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += CB) {
      const int c = c0 + cb;
      float b[2][4];
      load_pair(bp + (int64_t)c * (D / 4) * kTRow * NPL, b);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f32x4 dz = *reinterpret_cast<const f32x4*>(dht + k * kTC + 16 * c + s4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          accR[k][0] += dz[i] * b[0][i];
          accR[k][1] += dz[i] * b[1][i];
        }
      }
    }
  }
  float* out = a.hpart + (int64_t)blockIdx.x * (D + 3 * (D / 2) + 4);
  // lanes of one pair-row: the four sample granules (lane bits 0-1) and the chunk groups (lane bits above 2 + log2 P)
  auto rsum = [&](float v, int P) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    for (int mm = 4 * P; mm < 64; mm <<= 1) v += __shfl_xor(v, mm, 64);
    return v;
  };
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const float v = rsum(accS[par], PH);
    if ((lane & 3) == 0 && ch == 0) out[2 * PH * wave + 2 * qh + par] = v;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const float v = rsum(accR[k][par], PB);
      if ((lane & 3) == 0 && cb == 0) out[D + k * (D / 2) + 2 * PB * wave + 2 * qb + par] = v;
    }
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = accB[c];
#pragma unroll
      for (int mm = 32; mm >= 1; mm >>= 1) v += __shfl_xor(v, mm, 64);
      if (lane == 0) out[D + 3 * (D / 2) + c] = v;
    }
  }
}

struct HeadsRdArgs {
  const float* hpart;
  int32_t nsplit, D;
  const float* scale;
  float *dWs, *dbs, *dWr, *dbr;
  const uint32_t* status;  // as in RdArgs
  int32_t accumulate;
};
__global__ void k_heads_reduce(HeadsRdArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.D + 3 * (a.D / 2) + 4;
  if (e >= n) return;
  const float inv = a.scale ? 1.0f / a.scale[0] : 1.0f;
  float sum = sum_partials(a.hpart + e, n, a.nsplit) * inv;
  if (a.status && (a.status[0] & (FSN_STATUS_FP16_RANGE | FSN_STATUS_GRAD_RANGE))) sum = 0.f;
  float* dst;
  if (e < a.D) dst = a.dWs + e;
  else if (e < a.D + 3 * (a.D / 2)) dst = a.dWr + (e - a.D);
  else if (e < n - 1) dst = a.dbr + (e - a.D - 3 * (a.D / 2));
  else dst = a.dbs;
  *dst = a.accumulate ? *dst + sum : sum;
}

// Delayed per-stage scaling (see BwdStoreHook): after a backward, move every stage's factor so that the largest stored
// |high part| it just produced would have been 2^5 (eleven octaves of headroom to 65504 for the next batch - a stage's
// maximum moved by 2^8 between two consecutive batches early in a run - and 19 below it to 2^-14, under which the
// scaled low parts still resolve 2^-41 of the maximum); a stage that overflowed (maximum = infinity, the step is a
// skipped one) drops by 2^8, a stage that stored only zeros keeps its factor.  Factors stay powers of two in [2^-24, 2^40]; the maxima are cleared for the next call.
__global__ void k_bwd_rescale(float* __restrict__ bs, uint32_t* __restrict__ amax, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const uint32_t m = amax[i];
  amax[i] = 0u;
  if (m == 0u) return;
  float f = bs[i];
  if (m >= 0x7c00u) {
    f *= 0.00390625f;
  } else {
    // floor(log2) of the fp16 value with bit pattern m (subnormals: below 2^-14, counted from their leading bit)
    const int e = (m >> 10) ? (int)(m >> 10) - 15 : (31 - __builtin_clz(m)) - 24;
    f = ldexpf(f, 5 - e);
  }
  bs[i] = fminf(fmaxf(f, 5.9604644775390625e-08f), 1.099511627776e12f);
}

// ------------------------------------------------------------------ host side
static NetParams net_params(const fsn_mlp_desc& d, const NetGeom& G, const void* blob, uint32_t* status) {
  NetParams p;
  p.blob = static_cast<const char*>(blob);
  p.aux_off = (int32_t)G.aux_off; p.aux_floats = G.aux_floats; p.stream_off = (int32_t)G.stream_off;
  p.nph_density = G.nph_density; p.nph_full = G.nph_full;
  p.n_layers = d.n_layers; p.skip_mask = d.skip_mask;
  p.n_freqs_pos = d.n_freqs_pos; p.n_freqs_dir = d.n_freqs_dir;
  p.status = status;
  return p;
}

int64_t fused_train_workspace_floats(const fsn_mlp_desc& d, int prec, int64_t n) {
  FusedLayout F;
  const char* why;
  const int rc = make_fused_layout(d, prec, n, F, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "training path: %s", why);
  return F.total;
}

template <int NT, int PREC>
static int launch_fwd(const TrainFwdArgs& a, unsigned grid, hipStream_t s) {
  k_train_fwd<NT, PREC><<<grid, kThreads, 0, s>>>(a);
  FSN_LAUNCH_CHECK("k_train_fwd");
  return FSN_OK;
}
template <int NT, int PREC>
static int launch_bwd(const TrainBwdArgs& a, unsigned grid, hipStream_t s) {
  k_train_bwd<NT, PREC><<<grid, kThreads, 0, s>>>(a);
  FSN_LAUNCH_CHECK("k_train_bwd");
  return FSN_OK;
}
template <int NW, int MG, int BT>
static int launch_wgrad(int prec, const WgArgs& a, int njobs, hipStream_t s) {
  if (njobs == 0) return FSN_OK;
  const int a_rows = a.job[0].a_rows;  // (jobs of one launch share their shape)
  FSN_REQUIRE(a_rows % (64 * MG) == 0, FSN_E_HIP, "internal: wgrad row blocks");
  dim3 grid((unsigned)a.nsplit, (unsigned)njobs, (unsigned)(a_rows / (64 * MG)));
  switch (prec) {
    case 0: k_wgrad<NW, MG, BT, 0><<<grid, 64 * NW, 0, s>>>(a); break;
    case 1: k_wgrad<NW, MG, BT, 1><<<grid, 64 * NW, 0, s>>>(a); break;
    case 2: k_wgrad<NW, MG, BT, 2><<<grid, 64 * NW, 0, s>>>(a); break;
    default: k_wgrad<NW, MG, BT, 3><<<grid, 64 * NW, 0, s>>>(a); break;
  }
  FSN_LAUNCH_CHECK("k_wgrad");
  return FSN_OK;
}

int fused_train_fwd(const fsn_mlp_desc* d, int prec, const float* const* W, const float* const* b, const float* x,
                    const float* dirs, const float* pos_mask, const float* dir_mask, int64_t n, float* ws, float* out,
                    uint32_t* status, hipStream_t s, const TrainRays* rays) {
  FusedLayout F;
  const char* why;
  int rc = make_fused_layout(*d, prec, n, F, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_nerf_train_fwd: %s", why);
  NetGeom G;
  build_geom(*d, prec, G, &why);
  FSN_REQUIRE(G.aux_floats <= kAuxCapFloats, FSN_E_UNSUPPORTED, "fsn_nerf_train_fwd: network too deep for the LDS aux area");
  void* blob = ws + F.blob_f;
  rc = fsn_mlp_pack(d, prec, W, b, blob, (fsn_stream_t)s);
  if (rc != FSN_OK) return rc;
  const int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  TrainFwdArgs a{net_params(*d, G, blob, status), x, dirs, pos_mask, dir_mask, n, out, ws, F.h, F.h_stride, F.bo, F.pe, F.de,
                 F.mask, F.mask_stride, F.D, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (rays) { a.rays_o = rays->rays_o; a.rays_d = rays->rays_d; a.t0 = rays->t0; a.t1 = rays->t1; a.ri = rays->ri; }
  const unsigned grid = (unsigned)(F.T < cus ? F.T : cus);
  const int key = (F.D == 256 ? 4 : 0) + prec;
  switch (key) {
    case 0: return launch_fwd<4, 0>(a, grid, s);
    case 1: return launch_fwd<4, 1>(a, grid, s);
    case 2: return launch_fwd<4, 2>(a, grid, s);
    case 3: return launch_fwd<4, 3>(a, grid, s);
    case 4: return launch_fwd<8, 0>(a, grid, s);
    case 5: return launch_fwd<8, 1>(a, grid, s);
#ifdef FSN_EXP_TRAIN_ONEACC  // timing experiment (round 4): forward / backward in the one-accumulator arithmetic (results garbage)
    case 6: return launch_fwd<8, 4>(a, grid, s);
#else
    case 6: return launch_fwd<8, 2>(a, grid, s);
#endif
    default: return launch_fwd<8, 3>(a, grid, s);
  }
}

int fused_train_bwd(const fsn_mlp_desc* d, int prec, const float* const* W, int64_t n, float* ws, const float* out,
                    const float* d_out, const float* grad_scale_dev, float* const* dW, float* const* db,
                    bool accumulate, float* bscale, uint32_t* bamax, uint32_t* status, hipStream_t s) {
  FusedLayout F;
  const char* why;
  int rc = make_fused_layout(*d, prec, n, F, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_nerf_train_bwd: %s", why);
  NetGeom G;
  build_geom(*d, prec, G, &why);
  const int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  const int L = F.L, D = F.D, NT = F.NT;
  // ---- transposed-weight stream
  {
    BwdPackArgs p{};
    int u = 0, gi = 0;
    p.W[gi] = W[L + 2]; p.ld[gi] = D + F.d_de; p.ks[gi] = NT / 2; p.unit0[gi] = u; u += 2 * NT * (NT / 2); ++gi;
    p.W[gi] = W[L + 1]; p.ld[gi] = D; p.ks[gi] = NT; p.unit0[gi] = u; u += 2 * NT * NT; ++gi;
    for (int l = L - 1; l >= 1; --l) {
      const bool wide = (d->skip_mask >> (l - 1)) & 1u;
      p.W[gi] = W[l]; p.ld[gi] = D + (wide ? F.d_pe : 0); p.ks[gi] = NT; p.unit0[gi] = u; u += 2 * NT * NT; ++gi;
    }
    p.n_gemm = gi; p.units_total = u; p.prec = prec; p.NT = NT;
    FSN_REQUIRE(u == F.n_bwd_units, FSN_E_HIP, "internal: backward unit count");
    const int64_t n_pieces = (int64_t)F.nph_bwd * kPhaseBytes / 16;
    k_pack_bwd<<<(unsigned)((n_pieces + 255) / 256), 256, 0, s>>>(p, reinterpret_cast<char*>(ws + F.blob_b), n_pieces);
    FSN_LAUNCH_CHECK("k_pack_bwd");
  }
  // ---- dgrad chain
  {
    TrainBwdArgs a{net_params(*d, G, ws + F.blob_f, status), reinterpret_cast<const char*>(ws + F.blob_b), F.nph_bwd, n, out, d_out,
                   grad_scale_dev, ws, F.h, F.h_stride, F.bo, F.dhead, F.dbo, F.dp, F.mask, F.mask_stride,
                   prec_is_f16(prec) ? bscale : nullptr, prec_is_f16(prec) ? bamax : nullptr};
    const unsigned grid = (unsigned)(F.T < cus ? F.T : cus);
    const int key = (D == 256 ? 4 : 0) + prec;
    switch (key) {
      case 0: rc = launch_bwd<4, 0>(a, grid, s); break;
      case 1: rc = launch_bwd<4, 1>(a, grid, s); break;
      case 2: rc = launch_bwd<4, 2>(a, grid, s); break;
      case 3: rc = launch_bwd<4, 3>(a, grid, s); break;
      case 4: rc = launch_bwd<8, 0>(a, grid, s); break;
      case 5: rc = launch_bwd<8, 1>(a, grid, s); break;
#ifdef FSN_EXP_TRAIN_ONEACC
      case 6: rc = launch_bwd<8, 4>(a, grid, s); break;
#else
      case 6: rc = launch_bwd<8, 2>(a, grid, s); break;
#endif
      default: rc = launch_bwd<8, 3>(a, grid, s); break;
    }
    if (rc != FSN_OK) return rc;
  }
  // ---- wgrad jobs
  WgArgs wa[4] = {};
  int cnt[4] = {0, 0, 0, 0};
  RdArgs rd{};
  int nrd = 0;
  float* part = ws + F.part;
  auto U = [](const float* q) { return reinterpret_cast<const uint32_t*>(q); };  // saved tensors: packed T-layout
  auto H = [&](int i) { return ws + F.h + i * F.h_stride; };
  auto dP = [&](int i) { return ws + F.dp + i * F.h_stride; };
  auto add = [&](int kind, int stage, const float* A, const float* B, int a_rows, int b_rows, bool bias, float* dWp, float* dbp,
                 int ld, int col0, int mode, int n_freqs) {
    const int64_t ns = F.nsplit[kind];
    WgJob& j = wa[kind].job[cnt[kind]++];
    j.A = U(A); j.B = U(B); j.b_rows = b_rows; j.a_rows = a_rows;
    j.part = part; part += ns * a_rows * b_rows;
    j.bpart = nullptr;
    if (bias) { j.bpart = part; part += ns * a_rows; }
    RdJob& r = rd.job[nrd++];
    r.part = j.part; r.bpart = j.bpart; r.dW = dWp; r.db = dbp; r.a_rows = a_rows; r.b_rows = b_rows; r.ld = ld;
    r.col0 = col0; r.mode = mode; r.n_freqs = n_freqs; r.nsplit = (int)ns; r.stage = stage;
  };
  for (int l = 1; l < L; ++l) {
    const bool wide = (d->skip_mask >> (l - 1)) & 1u;
    add(WG_BIG, l, dP(l), H(l - 1), D, D, true, dW[l], db[l], D + (wide ? F.d_pe : 0), 0, 0, 0);
    if (wide) add(WG_ENC, l, dP(l), ws + F.pe, D, 64, false, dW[l], nullptr, D + F.d_pe, D, 1, d->n_freqs_pos);
  }
  add(WG_BIG, L, dP(L), H(L - 1), D, D, true, dW[L + 1], db[L + 1], D, 0, 0, 0);
  add(WG_ENC, 0, dP(0), ws + F.pe, D, 64, true, dW[0], db[0], F.d_pe, 0, 1, d->n_freqs_pos);
  add(WG_BR, L + 1, ws + F.dbo, H(L), D / 2, D, true, dW[L + 2], db[L + 2], D + F.d_de, 0, 0, 0);
  add(WG_BD, L + 1, ws + F.dbo, ws + F.de, D / 2, 32, false, dW[L + 2], nullptr, D + F.d_de, D, 2, d->n_freqs_dir);
  FSN_REQUIRE(part - (ws + F.part) <= part_floats(*d, F.nsplit), FSN_E_HIP, "internal: wgrad partial area");
  for (int k = 0; k < 4; ++k) { wa[k].T = F.T; wa[k].nsplit = F.nsplit[k]; }
  const WgArgs &big = wa[WG_BIG], &enc = wa[WG_ENC], &br = wa[WG_BR], &bd = wa[WG_BD];
  const int nbig = cnt[WG_BIG], nenc = cnt[WG_ENC], nbr = cnt[WG_BR], nbd = cnt[WG_BD];
  if (D == 256) {
#ifdef FSN_WGRAD_4W  // experiment: two 4-wave workgroups per CU, 128 rows of A each (twice the chunks in flight, but
    // B is read and staged twice): 3.55 ms against 2.80 ms for the 8-wave form
    if ((rc = launch_wgrad<4, 2, 4>(prec, big, nbig, s)) != FSN_OK) return rc;
#else
    if ((rc = launch_wgrad<8, 4, 4>(prec, big, nbig, s)) != FSN_OK) return rc;
#endif
    if ((rc = launch_wgrad<8, 4, 1>(prec, enc, nenc, s)) != FSN_OK) return rc;
    if ((rc = launch_wgrad<8, 2, 2>(prec, br, nbr, s)) != FSN_OK) return rc;
    if ((rc = launch_wgrad<8, 2, 1>(prec, bd, nbd, s)) != FSN_OK) return rc;
  } else {
    if ((rc = launch_wgrad<8, 2, 1>(prec, big, nbig, s)) != FSN_OK) return rc;
    if ((rc = launch_wgrad<8, 2, 1>(prec, enc, nenc, s)) != FSN_OK) return rc;
    if ((rc = launch_wgrad<8, 1, 1>(prec, br, nbr, s)) != FSN_OK) return rc;
    if ((rc = launch_wgrad<8, 1, 1>(prec, bd, nbd, s)) != FSN_OK) return rc;
  }
  rd.scale = grad_scale_dev;
  rd.status = prec_is_f16(prec) ? status : nullptr;  // (only the fp16 modes can overflow)
  rd.accumulate = accumulate ? 1 : 0;
  rd.bscale = prec_is_f16(prec) ? bscale : nullptr;
  {
    dim3 grid((unsigned)((D * D + 255) / 256), (unsigned)nrd);
    k_wgrad_reduce<<<grid, 256, 0, s>>>(rd);
    FSN_LAUNCH_CHECK("k_wgrad_reduce");
  }
  // ---- heads
  {
    HeadsArgs ha{U(H(L - 1)), U(ws + F.bo), ws + F.dhead, ws + F.hpart, F.T, F.nsplit_heads};
    const unsigned hg = (unsigned)F.nsplit_heads;
    switch ((D == 256 ? 4 : 0) + prec) {
      case 0: k_heads_wgrad<4, 0><<<hg, kThreads, 0, s>>>(ha); break;
      case 1: k_heads_wgrad<4, 1><<<hg, kThreads, 0, s>>>(ha); break;
      case 2: k_heads_wgrad<4, 2><<<hg, kThreads, 0, s>>>(ha); break;
      case 3: k_heads_wgrad<4, 3><<<hg, kThreads, 0, s>>>(ha); break;
      case 4: k_heads_wgrad<8, 0><<<hg, kThreads, 0, s>>>(ha); break;
      case 5: k_heads_wgrad<8, 1><<<hg, kThreads, 0, s>>>(ha); break;
      case 6: k_heads_wgrad<8, 2><<<hg, kThreads, 0, s>>>(ha); break;
      default: k_heads_wgrad<8, 3><<<hg, kThreads, 0, s>>>(ha); break;
    }
    FSN_LAUNCH_CHECK("k_heads_wgrad");
    HeadsRdArgs hr{ws + F.hpart, F.nsplit_heads, D, grad_scale_dev, dW[L], db[L], dW[L + 3], db[L + 3],
                   prec_is_f16(prec) ? status : nullptr, accumulate ? 1 : 0};
    const int nn = D + 3 * (D / 2) + 4;
    k_heads_reduce<<<(unsigned)((nn + 255) / 256), 256, 0, s>>>(hr);
    FSN_LAUNCH_CHECK("k_heads_reduce");
  }
  if (prec_is_f16(prec) && bscale && bamax) {
    k_bwd_rescale<<<1, 64, 0, s>>>(bscale, bamax, L + 2);
    FSN_LAUNCH_CHECK("k_bwd_rescale");
  }
  return FSN_OK;
}

}  // namespace fsn

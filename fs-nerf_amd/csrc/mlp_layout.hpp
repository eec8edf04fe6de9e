// mlp_layout.hpp — the packed-weights blob: geometry shared by the host packer, the device
// packer and the MFMA kernels.  (DESIGN.md "weight blob" describes this in prose.)
//
// The NeRF MLP (src/core/models.py:96-143) is evaluated TRANSPOSED on the matrix cores:
//   H_out^T [features x samples] = W [out x in] . H_in^T [in x samples]
// with v_mfma_f32_16x16x32_{f16,bf16}: A = a 16(out) x 32(in) weight fragment, B = a 32(in) x 16
// (samples) activation fragment, D = a 16(out) x 16(samples) fp32 accumulator tile with
// column = lane&15 = sample and row = 4(lane>>4) + reg = out feature.  Two such tiles (32 features)
// hold, lane for lane, exactly the 8 elements a lane needs as B operand of one k-step of the next
// layer, so activations never leave registers.
//
// A "unit" is the A operand of one (layer, 32-feature output pair tp, k-step ks, half sub): one
// 1-KiB fragment of the 16-bit (bf16 or fp16) high parts and, in the x3 modes, one 1-KiB fragment
// of the low parts, lane-linear (lane l owns bytes [16 l, 16 l + 16)).  Units are stored in the
// order the kernel consumes them (layer, pair, k-step, half) and streamed through LDS in 16-KiB
// "phases".
#pragma once
#include <cstdint>

#include "fsnerf_hip.h"

#if defined(__HIPCC__)
#define FSN_HD __host__ __device__ __forceinline__
#else
#define FSN_HD inline
#endif

namespace fsn {

constexpr uint32_t kBlobMagic = 0x4e53460au;  // "\nFSN"
constexpr int kPhaseBytes = 16384;
constexpr int kKsPos = 2;  // k-steps (of 32) reserved for the positional encoding: 64 slots
constexpr int kKsDir = 1;  // k-steps reserved for the direction encoding: 32 slots
constexpr int kMaxLayers = 16;

FSN_HD bool prec_is_x3(int prec) { return (prec & 1) == 0; }
FSN_HD bool prec_is_f16(int prec) { return prec >= 2; }
// Low parts of the split fp16 modes (weights in the blob, activations in registers and in the training workspace) are
// stored SCALED: low = fp16((v - high) * 2^11).  The unscaled remainder of a value below ~0.1 is an fp16 subnormal
// (fixed 2^-24 resolution: an activation of 1e-3 kept ~15 bits, a sigma error of 4e-4); scaled it is a normal fp16
// number down to |v| ~ 6e-5 (and still carries 2^-36 absolute below that).  The two correction products al.wh, ah.wl
// are summed in their own accumulator and folded in as 2^-11 * corr by the epilogue.  bf16 has float32's exponent
// range: scale 1.
constexpr float kLoScaleF16 = 2048.0f;
// FSN_PREC_FP16X3U (round 4): low parts UNSCALED, all three products of a unit in one accumulator - float32-grade because
// per-layer power-of-two scales folded into the packed weights (mlp_pack.hpp, PackArgs::sc_*) keep every layer's
// activations at 2^4 .. 2^10, where an activation's unscaled low part is a normal fp16 number or negligible against the
// layer's scale.
FSN_HD bool prec_lo_scaled(int prec) { return prec == FSN_PREC_FP16X3 || prec == FSN_PREC_FP16X2; }
FSN_HD float lo_scale(int prec) { return prec_lo_scaled(prec) ? kLoScaleF16 : 1.0f; }
FSN_HD int unit_bytes(int prec) { return prec_is_x3(prec) ? 2048 : 1024; }
FSN_HD int units_per_phase(int prec) { return kPhaseBytes / unit_bytes(prec); }

// One GEMM of the network as the kernel sees it.
struct LayerGeom {
  int32_t nt_out;      // 32-feature output pairs (two 16-row MFMA tiles each)
  int32_t ks_act;      // k-steps (of 32) fed by the previous layer's activations
  int32_t ks_enc;      // k-steps fed by an encoding (kKsPos / kKsDir) or 0
  int32_t enc_is_dir;  // which encoding
  int32_t n_freqs;     // of that encoding
  int32_t d_act;       // input columns coming from activations (0 for layer 0)
  int32_t ld;          // row stride of the source weight matrix (= in_features)
  int32_t unit0;       // first unit of this layer in the stream
};

struct NetGeom {
  int32_t n_gemm;  // n_layers + 2 (connection, branch)
  LayerGeom g[kMaxLayers + 2];
  int32_t units_hidden;  // units of layers 0..n_layers-1 (density-only pass stops here)
  int32_t units_total;
  int32_t nph_density, nph_full;  // phases per pass
  // aux region (float32 offsets)
  int32_t aux_bias[kMaxLayers + 2];
  int32_t aux_wsigma, aux_wrgb, aux_misc, aux_floats;
  int64_t aux_off, stream_off, total_bytes;  // byte offsets in the blob
};

// Encoding slot -> feature index of the reference's PositionalEncoder layout
// (src/core/models.py:28,37-39: [x, sin f0 x, cos f0 x, sin f1 x, ...], blocks d_in=3 wide), or -1.
// The four lanes g = lane>>4 that share a sample own `slots` = 8 * k-steps slots each, q = 8*ks + j.
// Pair p = 3*band+coord goes to lane group p&3 at slots (2i, 2i+1) = (sin, cos), i = p>>2; the
// identity features x, y sit in group 2's last two slots and z in group 3's second to last.
FSN_HD int enc_slot_feature(int q, int g, int n_freqs, int slots) {
  const int P = 3 * n_freqs;
  const int np = P > g ? (P - g + 3) / 4 : 0;  // pairs owned by this lane group
  if (q < 2 * np) {
    const int p = 4 * (q >> 1) + g;
    const int band = p / 3, coord = p - 3 * band;
    return 3 + band * 6 + (q & 1) * 3 + coord;
  }
  if (g == 2 && q >= slots - 2) return q - (slots - 2);  // x, y
  if (g == 3 && q == slots - 2) return 2;                // z
  return -1;
}

// Source column of weight fragment element (k-step ks, lane group g, element j) for a layer;
// -1 = zero padding.  Activation k-steps use the accumulator-as-operand order: element j of lane
// group g is register j&3 of output tile 2ks + (j>>2) of the previous layer, i.e.
//   feature = 32 ks + 16 (j>>2) + 4 g + (j&3).
FSN_HD int unit_src_col(const LayerGeom& L, int ks, int g, int j) {
  if (ks < L.ks_act) return 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
  const int e = ks - L.ks_act;
  const int slots = 8 * L.ks_enc;
  const int f = enc_slot_feature(8 * e + j, g, L.n_freqs, slots);
  return f < 0 ? -1 : L.d_act + f;
}

FSN_HD uint16_t bf16_rne(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  if ((v.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((v.u >> 16) | 0x40);  // NaN stays NaN
  return (uint16_t)((v.u + 0x7fffu + ((v.u >> 16) & 1u)) >> 16);
}
FSN_HD float bf16_to_f32(uint16_t b) {
  union { float f; uint32_t u; } v;
  v.u = (uint32_t)b << 16;
  return v.f;
}

// IEEE binary16, round to nearest even, subnormals kept
FSN_HD uint16_t f16_rne(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  const uint32_t sign = (v.u >> 16) & 0x8000u;
  uint32_t x = v.u & 0x7fffffffu;
  if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // >= 65520 rounds to infinity
  if (x < 0x38800000u) {                                      // < 2^-14: subnormal half or zero
    v.u = x;
    const float a = v.f * 16777216.0f;  // * 2^24, exact
    float r = (float)(int)a;            // a < 1024: truncate, then round half to even by hand
    const float frac = a - r;
    if (frac > 0.5f || (frac == 0.5f && ((int)r & 1))) r += 1.0f;
    return (uint16_t)(sign | (uint32_t)(int)r);
  }
  x += 0xfffu + ((x >> 13) & 1u);
  return (uint16_t)(sign | ((x - 0x38000000u) >> 13));
}
FSN_HD float f16_to_f32(uint16_t h) {
  union { float f; uint32_t u; } v;
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu;
  if (e == 0) {
    v.f = (float)m * 5.9604644775390625e-08f;  // m * 2^-24
    v.u |= sign;
    return v.f;
  }
  v.u = sign | (e == 31 ? 0x7f800000u | (m << 13) : ((e + 112u) << 23) | (m << 13));
  return v.f;
}
FSN_HD uint16_t half_rne(float f, bool f16) { return f16 ? f16_rne(f) : bf16_rne(f); }
FSN_HD float half_to_f32(uint16_t h, bool f16) { return f16 ? f16_to_f32(h) : bf16_to_f32(h); }

// Fills `G` from the descriptor; returns 0 or an FSN_E_* code (message via set_error on host).
inline int build_geom(const fsn_mlp_desc& d, int prec, NetGeom& G, const char** why) {
  *why = "";
  if (prec < 0 || (prec > FSN_PREC_FP16X3U && prec != FSN_PREC_FP16X2)) { *why = "unknown precision mode"; return FSN_E_INVALID; }
  if (d.d_hidden != 256 && d.d_hidden != 128) { *why = "d_hidden must be 128 or 256"; return FSN_E_UNSUPPORTED; }
  if (d.n_layers < 2 || d.n_layers > kMaxLayers) { *why = "n_layers must be in [2,16]"; return FSN_E_UNSUPPORTED; }
  if (d.n_freqs_pos < 0 || d.n_freqs_pos > 10) { *why = "n_freqs (position) must be <= 10"; return FSN_E_UNSUPPORTED; }
  if (d.n_freqs_dir < 0 || d.n_freqs_dir > 4) { *why = "n_freqs (direction) must be <= 4"; return FSN_E_UNSUPPORTED; }
  if (d.skip_mask >> (d.n_layers - 1)) { *why = "skip index >= n_layers-1 (the reference model cannot run it either)"; return FSN_E_INVALID; }
  const int D = d.d_hidden, NT = D / 32, L = d.n_layers;
  const int d_pe = 3 * (1 + 2 * d.n_freqs_pos), d_de = 3 * (1 + 2 * d.n_freqs_dir);
  int u = 0;
  for (int l = 0; l < L; ++l) {
    LayerGeom& g = G.g[l];
    const bool wide = l > 0 && ((d.skip_mask >> (l - 1)) & 1u);
    g.nt_out = NT;
    g.ks_act = l == 0 ? 0 : NT;
    g.ks_enc = (l == 0 || wide) ? kKsPos : 0;
    g.enc_is_dir = 0;
    g.n_freqs = d.n_freqs_pos;
    g.d_act = l == 0 ? 0 : D;
    g.ld = g.d_act + (g.ks_enc ? d_pe : 0);
    g.unit0 = u;
    u += 2 * g.nt_out * (g.ks_act + g.ks_enc);
  }
  G.units_hidden = u;
  LayerGeom& c = G.g[L];  // connection
  c = LayerGeom{NT, NT, 0, 0, 0, D, D, u};
  u += 2 * NT * NT;
  LayerGeom& b = G.g[L + 1];  // branch
  b = LayerGeom{NT / 2, NT, kKsDir, 1, d.n_freqs_dir, D, D + d_de, u};
  u += 2 * (NT / 2) * (NT + kKsDir);
  G.n_gemm = L + 2;
  G.units_total = u;
  const int upp = units_per_phase(prec);
  if (G.units_hidden % upp) { *why = "internal: hidden units not phase aligned"; return FSN_E_UNSUPPORTED; }
  G.nph_density = G.units_hidden / upp;
  G.nph_full = (G.units_total + upp - 1) / upp;
  int a = 0;
  for (int l = 0; l < L + 2; ++l) { G.aux_bias[l] = a; a += D; }
  G.aux_wsigma = a; a += D;
  G.aux_wrgb = a; a += 2 * D;  // [3][D/2]
  G.aux_misc = a; a += 4 + 32;  // b_sigma, b_rgb[3], freqs_pos[16], freqs_dir[16]
  G.aux_floats = (a + 63) / 64 * 64;
  G.aux_off = 256;
  G.stream_off = (G.aux_off + (int64_t)G.aux_floats * 4 + 4095) / 4096 * 4096;
  G.total_bytes = G.stream_off + (int64_t)G.nph_full * kPhaseBytes;
  return FSN_OK;
}

}  // namespace fsn

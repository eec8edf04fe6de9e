// mlp_pack.hpp — the weight packer's index code (reference state_dict -> packed blob, mlp_layout.hpp), shared by the
// device packer kernels (mlp.hip), the host packer fsn_mlp_pack_host and the sanitizer build of the host side
// (csrc/host_pack.cpp: plain C++, no HIP).
#pragma once
#include "mlp_layout.hpp"

namespace fsn {

struct PackArgs {
  NetGeom G;
  const float* W[kMaxLayers + 4];  // layers.0.., sigma, connection, branch, rgb (reference order)
  const float* b[kMaxLayers + 4];
  int32_t n_layers, d_hidden, prec;
  uint32_t skip_mask;
  int32_t n_freqs_pos, n_freqs_dir;
  float freqs_pos[16], freqs_dir[16];
  // per-layer activation scales folded into the packed network (fsn_mlp_pack_scaled; all 1 = the plain network): GEMM g
  // produces 2^exps[g] x the reference's activations.  Factors on GEMM g's weight columns fed by activations / by an
  // encoding, on its bias, and on the two float32 heads.
  int32_t exps[kMaxLayers + 2];
  float sc_act[kMaxLayers + 2], sc_enc[kMaxLayers + 2], sc_bias[kMaxLayers + 2], sc_sigma, sc_rgb;
};

FSN_HD float pow2f(int e) {  // 2^e as float32, |e| <= 126
  union { float f; uint32_t u; } v;
  v.u = (uint32_t)(e + 127) << 23;
  return v.f;
}

// exps: n_layers + 2 exponents in kernel GEMM order (hidden 0..L-1, connection, branch) or null (all zero)
inline void fill_scales(PackArgs& a, const int32_t* exps) {
  const int L = a.n_layers;
  for (int g = 0; g < L + 2; ++g) a.exps[g] = exps ? exps[g] : 0;
  for (int g = L + 2; g < kMaxLayers + 2; ++g) a.exps[g] = 0;
  for (int g = 0; g < L + 2; ++g) {
    const int prev = g == 0 ? 0 : (g <= L ? a.exps[g == L ? L - 1 : g - 1] : a.exps[L]);  // connection reads layer L-1, branch the connection
    a.sc_act[g] = pow2f(a.exps[g] - prev);
    a.sc_enc[g] = pow2f(a.exps[g]);
    a.sc_bias[g] = pow2f(a.exps[g]);
  }
  a.sc_sigma = pow2f(-a.exps[L - 1]);
  a.sc_rgb = pow2f(-a.exps[L + 1]);
}

// weight index (state_dict order) of GEMM g (kernel order: hidden 0..L-1, connection, branch)
FSN_HD int gemm_to_sd(int g, int L) { return g < L ? g : g + 1; }  // skips "sigma" at index L

// One 16-byte piece = 8 bf16 of (unit, part hi/lo, lane).  Shared by host and device packers.
FSN_HD void pack_piece(const PackArgs& a, int64_t piece, uint16_t out8[8]) {
  const int ub = unit_bytes(a.prec);
  const int ppu = ub / 16;  // pieces per unit
  const int unit = (int)(piece / ppu);
  const int rem = (int)(piece - (int64_t)unit * ppu);
  const int part = rem >> 6, lane = rem & 63;
  for (int j = 0; j < 8; ++j) out8[j] = 0;
  if (unit >= a.G.units_total) return;  // tail padding of the last phase
  int g = 0;
  while (g + 1 < a.G.n_gemm && a.G.g[g + 1].unit0 <= unit) ++g;
  const LayerGeom& Lg = a.G.g[g];
  const int ks_tot = Lg.ks_act + Lg.ks_enc;
  const int lu = unit - Lg.unit0;          // ((pair * ks_tot) + ks) * 2 + half
  const int sub = lu & 1;
  const int t = (lu >> 1) / ks_tot, ks = (lu >> 1) - t * ks_tot;
  const int r = lane & 15, grp = lane >> 4;
  const float* W = a.W[gemm_to_sd(g, a.n_layers)];
  const int row = 32 * t + 16 * sub + r;
  for (int j = 0; j < 8; ++j) {
    const int col = unit_src_col(Lg, ks, grp, j);
    if (col < 0) continue;
    const float w = W[(int64_t)row * Lg.ld + col] * (ks < Lg.ks_act ? a.sc_act[g] : a.sc_enc[g]);  // (exact: powers of two)
    const bool f16 = prec_is_f16(a.prec);
    const uint16_t hi = half_rne(w, f16);
    out8[j] = part == 0 ? hi : half_rne((w - half_to_f32(hi, f16)) * lo_scale(a.prec), f16);
  }
}

// aux float `i` of the blob
FSN_HD float aux_value(const PackArgs& a, int i) {
  const int D = a.d_hidden, L = a.n_layers;
  const int blk = i / D, off = i - blk * D;
  if (blk < L) return a.b[blk][off] * a.sc_bias[blk];                          // hidden biases
  if (blk == L) return a.b[L + 1][off] * a.sc_bias[L];                         // connection bias
  if (blk == L + 1) return off < D / 2 ? a.b[L + 2][off] * a.sc_bias[L + 1] : 0.f;  // branch bias
  if (blk == L + 2) return a.W[L][off] * a.sc_sigma;                           // sigma.weight [1,D]
  if (blk == L + 3 || blk == L + 4) {                                          // rgb.weight [3,D/2]
    const int k = i - (L + 3) * D;
    return k < 3 * (D / 2) ? a.W[L + 3][k] * a.sc_rgb : 0.f;
  }
  const int m = i - (L + 5) * D;
  if (m == 0) return a.b[L][0];                 // sigma.bias
  if (m >= 1 && m <= 3) return a.b[L + 3][m - 1];  // rgb.bias
  if (m >= 4 && m < 20) return a.freqs_pos[m - 4];
  if (m >= 20 && m < 36) return a.freqs_dir[m - 20];
  return 0.f;
}

FSN_HD void header_words(const PackArgs& a, uint32_t hw[64]) {
  for (int i = 0; i < 64; ++i) hw[i] = 0;
  hw[0] = kBlobMagic; hw[1] = 3;  // layout version 3: low parts scaled by lo_scale(prec) (mlp_layout.hpp); per-layer
  for (int g = 0; g < a.n_layers + 2; ++g) hw[16 + g] = (uint32_t)a.exps[g];  // activation exponents at words 16..
  hw[2] = (uint32_t)a.prec; hw[3] = (uint32_t)a.n_layers;
  hw[4] = (uint32_t)a.d_hidden; hw[5] = a.skip_mask; hw[6] = (uint32_t)a.n_freqs_pos;
  hw[7] = (uint32_t)a.n_freqs_dir; hw[8] = (uint32_t)a.G.units_total; hw[9] = (uint32_t)a.G.nph_full;
  hw[10] = (uint32_t)a.G.nph_density; hw[11] = (uint32_t)a.G.aux_off; hw[12] = (uint32_t)a.G.aux_floats;
  hw[13] = (uint32_t)a.G.stream_off; hw[14] = (uint32_t)(a.G.total_bytes & 0xffffffffu);
}


// Fills `a` from the descriptor and the weight / bias pointer tables; returns 0 or an FSN_E_* code and a message.
inline int fill_pack_args_raw(const fsn_mlp_desc* d, int prec, const float* const* W, const float* const* b, PackArgs& a,
                              const char** why, const int32_t* exps = nullptr) {
  *why = "";
  if (!d || !W || !b) { *why = "null pointer"; return FSN_E_INVALID; }
  const int rc = build_geom(*d, prec, a.G, why);
  if (rc != FSN_OK) return rc;
  for (int i = 0; i < d->n_layers + 4; ++i) {
    if (!W[i] || !b[i]) { *why = "null weight / bias pointer"; return FSN_E_INVALID; }
    a.W[i] = W[i];
    a.b[i] = b[i];
  }
  a.n_layers = d->n_layers; a.d_hidden = d->d_hidden; a.prec = prec; a.skip_mask = d->skip_mask;
  a.n_freqs_pos = d->n_freqs_pos; a.n_freqs_dir = d->n_freqs_dir;
  for (int i = 0; i < 16; ++i) { a.freqs_pos[i] = d->freqs_pos[i]; a.freqs_dir[i] = d->freqs_dir[i]; }
  if (exps)
    for (int g = 0; g < d->n_layers + 2; ++g)
      if (exps[g] < -60 || exps[g] > 60) { *why = "layer exponent outside [-60, 60]"; return FSN_E_INVALID; }
  fill_scales(a, exps);
  return FSN_OK;
}

// The whole blob on the host (format tests; the sanitizer build runs exactly this).
inline int pack_blob_host(const fsn_mlp_desc* d, int prec, const float* const* W, const float* const* b, void* blob_host,
                          const char** why, const int32_t* exps = nullptr) {
  PackArgs a;
  const int rc = fill_pack_args_raw(d, prec, W, b, a, why, exps);
  if (rc != FSN_OK) return rc;
  char* blob = static_cast<char*>(blob_host);
  for (int64_t i = 0; i < a.G.total_bytes; ++i) blob[i] = 0;
  uint32_t hw[64];
  header_words(a, hw);
  for (int i = 0; i < 64; ++i) reinterpret_cast<uint32_t*>(blob)[i] = hw[i];
  float* aux = reinterpret_cast<float*>(blob + a.G.aux_off);
  for (int i = 0; i < a.G.aux_floats; ++i) aux[i] = aux_value(a, i);
  const int64_t n_pieces = (int64_t)a.G.nph_full * kPhaseBytes / 16;
  uint16_t* sp = reinterpret_cast<uint16_t*>(blob + a.G.stream_off);
  for (int64_t p = 0; p < n_pieces; ++p) pack_piece(a, p, sp + p * 8);
  return FSN_OK;
}

}  // namespace fsn

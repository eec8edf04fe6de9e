// mlp.hip — weight packing (host + device) and the standalone NeRF.forward kernel.
// reference: src/core/models.py:53-143 (NeRF), state_dict layout SURVEY.md 8a (a5).
#include "common.hpp"
// x3 modes: pinned one-unit-ahead LDS prefetch of the A operands (mlp_dev.hpp).  +6 % on these kernels; the fused
// render kernel keeps the compiler's own read placement (with its larger live state the pinned form spills more).
#define FSN_X3_PF1
#define FSN_BF16X3_ONEACC 1  // inference: bf16x3 accumulates its three products in one tile (mlp_dev.hpp)
#include "mlp_dev.hpp"
#include "mlp_layout.hpp"
#include "mlp_pack.hpp"

#include <cstring>
#include <vector>

namespace fsn {

__global__ void k_pack_stream(PackArgs a, char* __restrict__ blob, int64_t n_pieces) {
  const int64_t piece = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (piece >= n_pieces) return;
  uint16_t o[8];
  pack_piece(a, piece, o);
  uint4 v;
  v.x = o[0] | ((uint32_t)o[1] << 16);
  v.y = o[2] | ((uint32_t)o[3] << 16);
  v.z = o[4] | ((uint32_t)o[5] << 16);
  v.w = o[6] | ((uint32_t)o[7] << 16);
  *reinterpret_cast<uint4*>(blob + a.G.stream_off + piece * 16) = v;
}

__global__ void k_pack_aux(PackArgs a, char* __restrict__ blob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 64) {
    uint32_t hw[64];
    header_words(a, hw);
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) w = (k == i) ? hw[k] : w;
    reinterpret_cast<uint32_t*>(blob)[i] = w;
  }
  if (i < a.G.aux_floats) reinterpret_cast<float*>(blob + a.G.aux_off)[i] = aux_value(a, i);
}

static int fill_pack_args(const fsn_mlp_desc* d, int prec, const float* const* W, const float* const* b,
                          PackArgs& a, const int32_t* exps = nullptr) {
  const char* why;
  const int rc = fill_pack_args_raw(d, prec, W, b, a, &why, exps);
  FSN_REQUIRE(rc == FSN_OK, rc, "mlp_pack: %s", why);
  return FSN_OK;
}

static NetParams make_net_params(const fsn_mlp_desc& d, const NetGeom& G, const void* blob, uint32_t* status) {
  NetParams p;
  p.blob = static_cast<const char*>(blob);
  p.aux_off = (int32_t)G.aux_off; p.aux_floats = G.aux_floats; p.stream_off = (int32_t)G.stream_off;
  p.nph_density = G.nph_density; p.nph_full = G.nph_full;
  p.n_layers = d.n_layers; p.skip_mask = d.skip_mask;
  p.n_freqs_pos = d.n_freqs_pos; p.n_freqs_dir = d.n_freqs_dir;
  p.status = status;
  return p;
}

// ------------------------------------------------------------------ NeRF.forward kernel
struct MlpFwdArgs {
  NetParams net;
  const float* x;
  const float* dirs;  // null -> density only
  const float* pos_mask;
  const float* dir_mask;
  int64_t n;
  float* out;
  // ray form (fsn_mlp_fwd_rays): sample s is the midpoint of [t0[s], t1[s]) on ray ri[s]; x / dirs are not read
  const float *rays_o, *rays_d, *t0, *t1;
  const int64_t* ri;
  int32_t full;
};

// sample s of the ray form: x = o + d (t0 + t1) / 2 in the reference's operation order (rendering.py:59-61, 77-79:
// to + td * (t_starts + t_ends)[:, None] / 2.0), dirs = d
__device__ __forceinline__ void ray_sample(const float* __restrict__ ro, const float* __restrict__ rd,
                                           const int64_t* __restrict__ ri, const float* __restrict__ t0,
                                           const float* __restrict__ t1, int64_t s, float* q, bool with_dir) {
  const int64_t r = ri[s];
  const float tm = t0[s] + t1[s];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dc = rd[3 * r + c];
    q[c] = ro[3 * r + c] + dc * tm / 2.0f;
    if (with_dir) q[3 + c] = dc;
  }
}

// sample source of the standalone kernel: the tile's positions / directions staged in LDS
struct TileSrc {
  const float* p;  // this lane's [x,y,z,dx,dy,dz] in LDS
  __device__ __forceinline__ void pos(float& x, float& y, float& z) const { x = p[0]; y = p[1]; z = p[2]; }
  __device__ __forceinline__ void dir(float& x, float& y, float& z) const { x = p[3]; y = p[4]; z = p[5]; }
};

// Persistent workgroups; tile = 128 NG consecutive samples; wave w / lane (c = lane&15, g = lane>>4) owns samples
// tile0 + 16*(NG*w + q) + c, q < NG (NG = 2 sample groups per wave in the single-pass modes of 256-wide networks,
// mlp_dev.hpp gemm_layer2).  LDS: [weight ring 64 KiB][aux + masks][tile inputs 128 NG x 6 floats].
template <int NT, int PREC>
constexpr int groups_per_wave() { return ((PREC & 1) == 1 && NT == 8) ? 2 : 1; }

template <int NT, int PREC, bool FULL>
__global__ __launch_bounds__(kThreads) void k_mlp_fwd(MlpFwdArgs a) {
  constexpr int NG = groups_per_wave<NT, PREC>(), TILE = 128 * NG;
  __shared__ __attribute__((aligned(1024))) char smem[kRingBytes + (kAuxCapFloats + 96) * 4 + TILE * 6 * 4];
  float* aux_lds = reinterpret_cast<float*>(smem + kRingBytes);
  float* in_lds = aux_lds + kAuxCapFloats + 96;
  NetDev net;
  load_net(a.net, a.pos_mask, a.dir_mask, aux_lds, net);
  __syncthreads();
  constexpr bool full = FULL;
  const int64_t ntiles = (a.n + TILE - 1) / TILE;
  WStream st;
  const char* sbase = a.net.blob + a.net.stream_off;
  st.init(smem, nullptr, 0, 0, sbase, (uint32_t)(full ? a.net.nph_full : a.net.nph_density), 1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  ARing ring;
  prime_ring<PREC, NT>(st, ring);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // each wave stages (and later reads) only its own 16 NG samples: no workgroup barrier
    if (lane < 16 * NG) {
      const int64_t s = tile * TILE + wave * (16 * NG) + lane;
      const int64_t sc = s < a.n ? s : a.n - 1;
      float* q = in_lds + (wave * (16 * NG) + lane) * 6;
      if (a.ri) {
        ray_sample(a.rays_o, a.rays_d, a.ri, a.t0, a.t1, sc, q, full);
      } else {
        q[0] = a.x[3 * sc]; q[1] = a.x[3 * sc + 1]; q[2] = a.x[3 * sc + 2];
        if (full) { q[3] = a.dirs[3 * sc]; q[4] = a.dirs[3 * sc + 1]; q[5] = a.dirs[3 * sc + 2]; }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t s = tile * TILE + wave * (16 * NG) + (lane & 15);
    if constexpr (NG == 1) {
      const TileSrc src{in_lds + (wave * 16 + (lane & 15)) * 6};
      float sigma, rgb[3] = {0.f, 0.f, 0.f};
      mlp_tile<NT, PREC, FULL>(st, net, src, ring, sigma, rgb);
      if (lane < 16 && s < a.n) {
        if (full) {
          f32x4 o = {rgb[0], rgb[1], rgb[2], sigma};
          *reinterpret_cast<f32x4*>(a.out + 4 * s) = o;
        } else {
          a.out[s] = sigma;
        }
      }
    } else {
      const TileSrc src0{in_lds + (wave * 32 + (lane & 15)) * 6}, src1{in_lds + (wave * 32 + 16 + (lane & 15)) * 6};
      float sigma[2], rgb[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
      mlp_tile2<NT, PREC, FULL>(st, net, src0, src1, ring, sigma, rgb);
      if (lane < 32) {  // lanes 0-15 write group 0, lanes 16-31 group 1 (every lane holds both results)
        const int q = lane >> 4;
        const int64_t sq = s + 16 * q;
        if (sq < a.n) {
          if (full) {
            f32x4 o = {rgb[q][0], rgb[q][1], rgb[q][2], sigma[q]};
            *reinterpret_cast<f32x4*>(a.out + 4 * sq) = o;
          } else {
            a.out[sq] = sigma[q];
          }
        }
      }
    }
  }
  st.drain();
}

// ------------------------------------------------------------------ per-layer activation maxima (calibration)
// fsn_mlp_layer_maxima: the full forward of k_mlp_fwd with a hook that keeps, per GEMM, the largest |output after its
// activation| - what the host turns into the per-layer power-of-two scales of FSN_PREC_FP16X3U (fsn_mlp_pack_scaled).
struct MaxSave {
  static constexpr bool kSave = false;
  struct Hook {
    static constexpr bool kZeroInit = false;
    static constexpr bool kPacked = false;
    static constexpr bool kLayerEnd = false;
    float m;
    __device__ __forceinline__ void pre(int) {}
    __device__ __forceinline__ void post(int, float (&v)[8]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
    }
  };
  float* wave_max;  // LDS: this wave's kMaxLayers + 2 running maxima
  __device__ __forceinline__ Hook hidden(int) const { return Hook{0.f}; }
  __device__ __forceinline__ Hook branch() const { return Hook{0.f}; }
  __device__ __forceinline__ uint32_t* enc_pos(int) const { return nullptr; }
  __device__ __forceinline__ uint32_t* enc_dir(int) const { return nullptr; }
  __device__ __forceinline__ void layer_done(int l, Hook& hk) const {
    float m = hk.m;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) wave_max[l] = fmaxf(wave_max[l], m);
  }
};

template <int NT, int PREC>
__global__ __launch_bounds__(kThreads) void k_mlp_maxima(MlpFwdArgs a, float* __restrict__ maxima) {
  constexpr int TILE = 128, NSLOT = kMaxLayers + 2;
  __shared__ __attribute__((aligned(1024))) char smem[kRingBytes + (kAuxCapFloats + 96) * 4 + TILE * 6 * 4 + kWaves * NSLOT * 4];
  float* aux_lds = reinterpret_cast<float*>(smem + kRingBytes);
  float* in_lds = aux_lds + kAuxCapFloats + 96;
  float* max_lds = in_lds + TILE * 6;
  NetDev net;
  load_net(a.net, a.pos_mask, a.dir_mask, aux_lds, net);
  net.status = nullptr;  // (a calibration pass reports maxima, not range bits: inf shows as inf)
  for (int i = threadIdx.x; i < kWaves * NSLOT; i += blockDim.x) max_lds[i] = 0.f;
  __syncthreads();
  const int64_t ntiles = (a.n + TILE - 1) / TILE;
  WStream st;
  st.init(smem, nullptr, 0, 0, a.net.blob + a.net.stream_off, (uint32_t)a.net.nph_full, 1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  ARing ring;
  prime_ring<PREC, NT>(st, ring);
  const MaxSave sv{max_lds + wave * NSLOT};
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (lane < 16) {
      const int64_t s = tile * TILE + wave * 16 + lane;
      const int64_t sc = s < a.n ? s : a.n - 1;  // (tail slots repeat the last sample: maxima unchanged)
      float* q = in_lds + (wave * 16 + lane) * 6;
      q[0] = a.x[3 * sc]; q[1] = a.x[3 * sc + 1]; q[2] = a.x[3 * sc + 2];
      q[3] = a.dirs[3 * sc]; q[4] = a.dirs[3 * sc + 1]; q[5] = a.dirs[3 * sc + 2];
    }
    __builtin_amdgcn_wave_barrier();
    const TileSrc src{in_lds + (wave * 16 + (lane & 15)) * 6};
    float sigma, rgb[3] = {0.f, 0.f, 0.f};
    mlp_tile<NT, PREC, true>(st, net, src, ring, sigma, rgb, sv);
  }
  st.drain();
  __syncthreads();
  if ((int)threadIdx.x < a.net.n_layers + 2) {
    float m = 0.f;
    for (int w = 0; w < kWaves; ++w) m = fmaxf(m, max_lds[w * NSLOT + threadIdx.x]);
    // non-negative floats (and +inf) order like their bit patterns
    atomicMax(reinterpret_cast<unsigned int*>(maxima) + threadIdx.x, __float_as_uint(m));
  }
}

template <int NT, int PREC>
static int launch_mlp_maxima(const MlpFwdArgs& a, float* maxima, int cus, hipStream_t s) {
  const int64_t ntiles = (a.n + 127) / 128;
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  k_mlp_maxima<NT, PREC><<<grid, kThreads, 0, s>>>(a, maxima);
  FSN_LAUNCH_CHECK("k_mlp_maxima");
  return FSN_OK;
}

template <int NT, int PREC>
static int launch_mlp_fwd(const MlpFwdArgs& a, int cus, hipStream_t s) {
  constexpr int TILE = 128 * groups_per_wave<NT, PREC>();
  const int64_t ntiles = (a.n + TILE - 1) / TILE;
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  if (a.dirs || (a.ri && a.full)) k_mlp_fwd<NT, PREC, true><<<grid, kThreads, 0, s>>>(a);
  else k_mlp_fwd<NT, PREC, false><<<grid, kThreads, 0, s>>>(a);
  FSN_LAUNCH_CHECK("k_mlp_fwd");
  return FSN_OK;
}

}  // namespace fsn

using namespace fsn;

extern "C" int64_t fsn_mlp_blob_bytes(const fsn_mlp_desc* desc, int prec) {
  FSN_REQUIRE(desc, FSN_E_INVALID, "fsn_mlp_blob_bytes: null desc");
  NetGeom G;
  const char* why;
  const int rc = build_geom(*desc, prec, G, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_mlp_blob_bytes: %s", why);
  return G.total_bytes;
}

extern "C" int fsn_mlp_pack(const fsn_mlp_desc* desc, int prec, const float* const* weights, const float* const* biases,
                            void* blob, fsn_stream_t stream) {
  return fsn_mlp_pack_scaled(desc, prec, weights, biases, nullptr, blob, stream);
}

extern "C" int fsn_mlp_pack_scaled(const fsn_mlp_desc* desc, int prec, const float* const* weights,
                                   const float* const* biases, const int32_t* layer_exps, void* blob, fsn_stream_t stream) {
  FSN_REQUIRE(blob, FSN_E_INVALID, "fsn_mlp_pack: null blob");
  PackArgs a;
  const int rc = fill_pack_args(desc, prec, weights, biases, a, layer_exps);
  if (rc != FSN_OK) return rc;
  const int64_t n_pieces = (int64_t)a.G.nph_full * kPhaseBytes / 16;
  k_pack_stream<<<(unsigned)((n_pieces + 255) / 256), 256, 0, as_stream(stream)>>>(a, static_cast<char*>(blob), n_pieces);
  FSN_LAUNCH_CHECK("k_pack_stream");
  const int naux = a.G.aux_floats > 64 ? a.G.aux_floats : 64;
  k_pack_aux<<<(unsigned)((naux + 255) / 256), 256, 0, as_stream(stream)>>>(a, static_cast<char*>(blob));
  FSN_LAUNCH_CHECK("k_pack_aux");
  return FSN_OK;
}

extern "C" int fsn_mlp_pack_host(const fsn_mlp_desc* desc, int prec, const float* const* weights,
                                 const float* const* biases, void* blob_host) {
  FSN_REQUIRE(blob_host, FSN_E_INVALID, "fsn_mlp_pack_host: null blob");
  const char* why;
  const int rc = pack_blob_host(desc, prec, weights, biases, blob_host, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_mlp_pack_host: %s", why);
  return FSN_OK;
}

extern "C" int fsn_mlp_pack_scaled_host(const fsn_mlp_desc* desc, int prec, const float* const* weights,
                                        const float* const* biases, const int32_t* layer_exps, void* blob_host) {
  FSN_REQUIRE(blob_host, FSN_E_INVALID, "fsn_mlp_pack_scaled_host: null blob");
  const char* why;
  const int rc = pack_blob_host(desc, prec, weights, biases, blob_host, &why, layer_exps);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_mlp_pack_scaled_host: %s", why);
  return FSN_OK;
}

static int mlp_fwd_any(const char* who, const fsn_mlp_desc* desc, int prec, const void* blob, MlpFwdArgs a, uint32_t* status,
                       fsn_stream_t stream) {
  FSN_REQUIRE(desc && a.n >= 0, FSN_E_INVALID, "%s: bad arguments", who);
  NetGeom G;
  const char* why;
  const int rc = build_geom(*desc, prec, G, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "%s: %s", who, why);
  if (a.n == 0) return FSN_OK;
  FSN_REQUIRE(blob && a.out && (a.x || (a.ri && a.rays_o && a.rays_d && a.t0 && a.t1)), FSN_E_INVALID, "%s: null pointer", who);
  FSN_REQUIRE(G.aux_floats <= kAuxCapFloats, FSN_E_UNSUPPORTED, "%s: network too deep for the LDS aux area", who);
  const int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  a.net = make_net_params(*desc, G, blob, status);
  hipStream_t s = as_stream(stream);
  if (prec == FSN_PREC_FP16X2) return desc->d_hidden == 256 ? launch_mlp_fwd<8, 6>(a, cus, s) : launch_mlp_fwd<4, 6>(a, cus, s);
  if (prec == FSN_PREC_FP16X3U) return desc->d_hidden == 256 ? launch_mlp_fwd<8, 4>(a, cus, s) : launch_mlp_fwd<4, 4>(a, cus, s);
  const int key = (desc->d_hidden == 256 ? 4 : 0) + prec;
  switch (key) {
    case 0: return launch_mlp_fwd<4, 0>(a, cus, s);
    case 1: return launch_mlp_fwd<4, 1>(a, cus, s);
    case 2: return launch_mlp_fwd<4, 2>(a, cus, s);
    case 3: return launch_mlp_fwd<4, 3>(a, cus, s);
    case 4: return launch_mlp_fwd<8, 0>(a, cus, s);
    case 5: return launch_mlp_fwd<8, 1>(a, cus, s);
    case 6: return launch_mlp_fwd<8, 2>(a, cus, s);
    default: return launch_mlp_fwd<8, 3>(a, cus, s);
  }
}

extern "C" int fsn_mlp_fwd(const fsn_mlp_desc* desc, int prec, const void* blob, const float* x, const float* dirs,
                           const float* pos_mask, const float* dir_mask, int64_t n, float* out, uint32_t* status,
                           fsn_stream_t stream) {
  MlpFwdArgs a{};
  a.x = x; a.dirs = dirs; a.pos_mask = pos_mask; a.dir_mask = dir_mask; a.n = n; a.out = out;
  return mlp_fwd_any("fsn_mlp_fwd", desc, prec, blob, a, status, stream);
}

extern "C" int fsn_mlp_fwd_rays(const fsn_mlp_desc* desc, int prec, const void* blob, const float* rays_o,
                                const float* rays_d, const int64_t* ray_indices, const float* t_starts,
                                const float* t_ends, int full, const float* pos_mask, const float* dir_mask, int64_t n,
                                float* out, uint32_t* status, fsn_stream_t stream) {
  MlpFwdArgs a{};
  a.rays_o = rays_o; a.rays_d = rays_d; a.ri = ray_indices; a.t0 = t_starts; a.t1 = t_ends; a.full = full ? 1 : 0;
  a.pos_mask = pos_mask; a.dir_mask = dir_mask; a.n = n; a.out = out;
  return mlp_fwd_any("fsn_mlp_fwd_rays", desc, prec, blob, a, status, stream);
}

extern "C" int fsn_mlp_layer_maxima(const fsn_mlp_desc* desc, int prec, const void* blob, const float* x, const float* dirs,
                                    const float* pos_mask, const float* dir_mask, int64_t n, float* maxima,
                                    fsn_stream_t stream) {
  FSN_REQUIRE(desc && n >= 0, FSN_E_INVALID, "fsn_mlp_layer_maxima: bad arguments");
  FSN_REQUIRE(prec == FSN_PREC_BF16X3 || prec == FSN_PREC_FP16X3U, FSN_E_UNSUPPORTED,
              "fsn_mlp_layer_maxima: precision mode %d (FSN_PREC_BF16X3 or FSN_PREC_FP16X3U)", prec);
  NetGeom G;
  const char* why;
  const int rc = build_geom(*desc, prec, G, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_mlp_layer_maxima: %s", why);
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(blob && x && dirs && maxima, FSN_E_INVALID, "fsn_mlp_layer_maxima: null pointer");
  FSN_REQUIRE(G.aux_floats <= kAuxCapFloats, FSN_E_UNSUPPORTED, "fsn_mlp_layer_maxima: network too deep for the LDS aux area");
  const int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  MlpFwdArgs a{};
  a.x = x; a.dirs = dirs; a.pos_mask = pos_mask; a.dir_mask = dir_mask; a.n = n;
  a.net = make_net_params(*desc, G, blob, nullptr);
  hipStream_t s = as_stream(stream);
  if (prec == FSN_PREC_BF16X3) return desc->d_hidden == 256 ? launch_mlp_maxima<8, 0>(a, maxima, cus, s) : launch_mlp_maxima<4, 0>(a, maxima, cus, s);
  return desc->d_hidden == 256 ? launch_mlp_maxima<8, 4>(a, maxima, cus, s) : launch_mlp_maxima<4, 4>(a, maxima, cus, s);
}

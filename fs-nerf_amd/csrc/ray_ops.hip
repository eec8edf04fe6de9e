// ray_ops.hip — the HBM-bound pieces of the path as standalone kernels + their C-ABI:
// get_rays, to_ndc, posenc_fwd, stratified_edges, edges_to_packed, sample_pdf_merge,
// composite (dense and packed).  All are streaming kernels: coalesced row-major access,
// one thread per output element or one wavefront per ray.
#include "common.hpp"
#include "ray_dev.hpp"

namespace fsn {

// ---------------------------------------------------------------- get_rays
// reference: src/utils/utilities.py:36-82.  One thread per pixel.
struct Pose34 { float m[12]; };

__global__ void k_get_rays(Pose34 P, int W, float half_w, float half_h, float focal, int row0,
                           int64_t npix, float* __restrict__ ro, float* __restrict__ rd) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int h = row0 + (int)(p / W);
  const int w = (int)(p % W);
  float o[3], d[3];
  pinhole_ray(P.m, half_w, half_h, focal, h, w, o, d);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rd[3 * p + k] = d[k];
    ro[3 * p + k] = o[k];
  }
}

// ---------------------------------------------------------------- to_ndc
// reference: src/utils/utilities.py:84-120.  The one definition used by k_to_ndc and k_build_rays.
__device__ __forceinline__ void ndc_ray(float (&o)[3], float (&d)[3], float sx, float sy, float near, float two_near) {
  const float dx = d[0], dy = d[1], dz = d[2];
  float ox = o[0], oy = o[1], oz = o[2];
  const float t = -(near + oz) / dz;
  ox = ox + t * dx;
  oy = oy + t * dy;
  oz = oz + t * dz;
  o[0] = sx * ox / oz;
  o[1] = sy * oy / oz;
  o[2] = 1.0f + two_near / oz;
  d[0] = sx * (dx / dz - ox / oz);
  d[1] = sy * (dy / dz - oy / oz);
  d[2] = -two_near / oz;
}

__global__ void k_to_ndc(const float* __restrict__ ro, const float* __restrict__ rd, int64_t n,
                         float sx, float sy, float near, float two_near, float* __restrict__ no,
                         float* __restrict__ nd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[3] = {ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]}, d[3] = {rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]};
  ndc_ray(o, d, sx, sy, near, two_near);
#pragma unroll
  for (int k = 0; k < 3; ++k) { no[3 * i + k] = o[k]; nd[3 * i + k] = d[k]; }
}

// ---------------------------------------------------------------- dataset ray tables
// reference: src/nerfdata/datasets/blender.py:174-191, llff.py:59-90 (torch.stack of per-pose get_rays, to_ndc over all of
// them, min / max of {o, o + d} over all rays).  ONE launch for [n_poses, H, W]: thread = (pose, pixel); the rays go
// straight into the [n*H*W, 3] tables; the region of interest is reduced on the way (wave min / max by shuffles, one
// atomic per wave and component on order-preserving integer keys - min / max are exact in any order, so the result is
// bit for bit torch's).  keys: 6 uint32, [0..2] minima, [3..5] maxima; k_aabb_init / k_aabb_finish around the launch.
__device__ __forceinline__ uint32_t f32_key(float f) {  // monotone float -> uint32
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__global__ void k_aabb_init(uint32_t* __restrict__ keys) {
  if (threadIdx.x < 3) keys[threadIdx.x] = 0xffffffffu;
  else if (threadIdx.x < 6) keys[threadIdx.x] = 0u;
}
__global__ void k_aabb_finish(const uint32_t* __restrict__ keys, float div, float* __restrict__ aabb) {
  if (threadIdx.x < 6) aabb[threadIdx.x] = key_f32(keys[threadIdx.x]) / div;
}

__global__ void k_build_rays(const float* __restrict__ poses, int64_t n_poses, int H, int W, float half_w, float half_h,
                             float focal, int ndc, float sx, float sy, float near, float two_near,
                             float* __restrict__ ro, float* __restrict__ rd, uint32_t* __restrict__ keys) {
  const int64_t per = (int64_t)H * W, total = n_poses * per;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < total;
  const int64_t pc = live ? p : total - 1;
  const int64_t pose = pc / per, pix = pc - pose * per;
  float o[3], d[3];
  pinhole_ray(poses + 12 * pose, half_w, half_h, focal, (int)(pix / W), (int)(pix % W), o, d);
  if (ndc) ndc_ray(o, d, sx, sy, near, two_near);
  if (live) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { rd[3 * p + k] = d[k]; ro[3 * p + k] = o[k]; }
  }
  if (!keys) return;
  // (dead lanes carry a copy of the last ray: min / max unchanged)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float e = o[k] + d[k];
    float lo = fminf(o[k], e), hi = fmaxf(o[k], e);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, off, 64));
      hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(keys + k, f32_key(lo));
      atomicMax(keys + 3 + k, f32_key(hi));
    }
  }
}

// ---------------------------------------------------------------- posenc
// reference: src/core/models.py:43-50.  One thread per output element (coalesced writes).
struct Freqs { float f[16]; };

__global__ void k_posenc(const float* __restrict__ x, int64_t n, int d_in, int n_freqs, Freqs fr,
                         const float* __restrict__ mask, float* __restrict__ out) {
  const int d_out = d_in * (1 + 2 * n_freqs);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * d_out) return;
  const int64_t p = e / d_out;
  const int f = (int)(e - p * d_out);
  float v;
  if (f < d_in) {
    v = x[p * d_in + f];
  } else {
    const int q = f - d_in;
    const int band = q / (2 * d_in);
    const int r = q - band * 2 * d_in;
    const int c = (r < d_in) ? r : r - d_in;
    const float a = x[p * d_in + c] * fr.f[band];
    v = (r < d_in) ? sinf(a) : cosf(a);
  }
  if (mask) v = v * mask[f];
  out[e] = v;
}

// ---------------------------------------------------------------- sampler
__global__ void k_stratified_edges(float near, float step, int S, int64_t R, const float* __restrict__ u,
                                   int u_mode, float* __restrict__ edges) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= R * (S + 1)) return;
  const int64_t r = e / (S + 1);
  const int i = (int)(e - r * (S + 1));
  const float* ur = (u_mode == 1) ? u + r : ((u_mode == 2) ? u + r * (S + 1) : nullptr);
  edges[e] = stratified_edge(near, step, S, i, u_mode, ur);
}

__global__ void k_edges_to_packed(const float* __restrict__ edges, int64_t R, int S,
                                  int64_t* __restrict__ ri, float* __restrict__ t0, float* __restrict__ t1) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= R * S) return;
  const int64_t r = e / S;
  const int i = (int)(e - r * S);
  ri[e] = r;
  t0[e] = edges[r * (S + 1) + i];
  t1[e] = edges[r * (S + 1) + i + 1];
}

static inline int pow2ceil(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// one wave per ray, 4 rays per 256-thread block; dynamic LDS = 4 * (2*S+2+pow2ceil(n_imp)) floats
__global__ void k_sample_pdf_merge(const float* __restrict__ edges, const float* __restrict__ w, int64_t R,
                                   int S, int n_imp, const float* __restrict__ u, float* __restrict__ out) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int M = S + 1 + n_imp;
  int n2 = 1;
  while (n2 < n_imp) n2 <<= 1;
  float* cdf_s = smem + (size_t)wave * (2 * S + 2 + n2);
  float* vals_s = cdf_s + (S + 1);
  sample_pdf_merge_ray(edges + r * (S + 1), w + r * S, S, n_imp, u ? u + r * n_imp : nullptr, cdf_s, vals_s,
                       out + r * M);
}

// ---------------------------------------------------------------- compositing
struct Bkgd { int has; float c[3]; };

__global__ void k_composite_dense(const float* __restrict__ sig, const float* __restrict__ rgb,
                                  const float* __restrict__ t0, const float* __restrict__ t1, int64_t R, int S,
                                  Bkgd bk, float* __restrict__ colors, float* __restrict__ opacity,
                                  float* __restrict__ depth, float* __restrict__ weights,
                                  float* __restrict__ alphas, float* __restrict__ trans) {
  const int wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int64_t b = r * S;
  CompositeOut o{colors + 3 * r, opacity + r, depth + r, weights ? weights + b : nullptr,
                 alphas ? alphas + b : nullptr, trans ? trans + b : nullptr};
  composite_ray(sig + b, rgb + 3 * b, t0 + b, t1 + b, S, bk.has != 0, bk.c[0], bk.c[1], bk.c[2], o);
}

// packed form: the ray's sample range is found by two binary searches in the sorted
// ray_indices, then the same per-wave routine runs on that slice.
__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void k_composite_packed(const float* __restrict__ sig, const float* __restrict__ rgb,
                                   const float* __restrict__ t0, const float* __restrict__ t1,
                                   const int64_t* __restrict__ ri, int64_t N, int64_t R, Bkgd bk,
                                   float* __restrict__ colors, float* __restrict__ opacity,
                                   float* __restrict__ depth, float* __restrict__ weights,
                                   float* __restrict__ alphas, float* __restrict__ trans) {
  const int wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int64_t b = lower_bound_i64(ri, N, r);
  const int64_t e = lower_bound_i64(ri, N, r + 1);
  CompositeOut o{colors + 3 * r, opacity + r, depth + r, weights ? weights + b : nullptr,
                 alphas ? alphas + b : nullptr, trans ? trans + b : nullptr};
  composite_ray(sig + b, rgb + 3 * b, t0 + b, t1 + b, (int)(e - b), bk.has != 0, bk.c[0], bk.c[1], bk.c[2], o);
}

// ---------------------------------------------------------------- "next" rows
// OcclusionRegularizer (src/core/loss.py:26-60): one wavefront per ray sums w(t) sigma over the ray's
// samples (range found like in k_composite_packed); a second, single-block kernel averages the sums of the
// non-empty rays in a fixed order (deterministic, no float atomics).
__global__ void k_occl_ray_sums(const float* __restrict__ sig, const float* __restrict__ t,
                                const int64_t* __restrict__ ri, int64_t N, int64_t R, float a, float b, int func,
                                float* __restrict__ sums) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int64_t lo = lower_bound_i64(ri, N, r), hi = lower_bound_i64(ri, N, r + 1);
  float acc = 0.f;
  for (int64_t i = lo + lane; i < hi; i += 64) {
    const float w = func == 0 ? (-a * t[i] + b) : (a * expf(-b * t[i]));
    acc += w * sig[i];
  }
  acc = wave_sum(acc);
  if (lane == 0) sums[r] = (hi > lo) ? acc : __builtin_nanf("");  // NaN marks "no samples"
}

__global__ void k_occl_mean(const float* __restrict__ sums, int64_t R, float* __restrict__ out) {
  __shared__ float s_sum[256];
  __shared__ int s_cnt[256];
  float acc = 0.f;
  int cnt = 0;
  for (int64_t r = threadIdx.x; r < R; r += 256) {
    const float v = sums[r];
    if (v == v) { acc += v; ++cnt; }
  }
  s_sum[threadIdx.x] = acc;
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) { s_sum[threadIdx.x] += s_sum[threadIdx.x + k]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s_sum[0] / (float)s_cnt[0];  // 0/0 = NaN like torch.mean of an empty stack
}

// backward of the occlusion regulariser w.r.t. sigmas: d sigma_i = g w(t_i) / #rays with samples
__global__ void k_occl_count(const float* __restrict__ sums, int64_t R, int* __restrict__ cnt) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = r < R && sums[r] == sums[r];
  const uint64_t m = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(cnt, __popcll(m));
}
__global__ void k_occl_bwd(const float* __restrict__ t, int64_t N, float a, float b, int func,
                           const float* __restrict__ g, const int* __restrict__ cnt, float* __restrict__ d_sig) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float w = func == 0 ? (-a * t[i] + b) : (a * expf(-b * t[i]));
  d_sig[i] = g[0] * w / (float)cnt[0];
}

// to8b (src/render/rendering.py:21): (255 * clip(x,0,1)).astype(uint8) - truncation, like numpy
__global__ void k_to8b(const float* __restrict__ x, int64_t n, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = 255.0f * fminf(fmaxf(x[i], 0.0f), 1.0f);
  out[i] = (uint8_t)v;
}

static inline unsigned nblocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace fsn

using namespace fsn;

extern "C" int fsn_get_rays(const float* pose_host, int H, int W, double focal, int row0, int nrows,
                            float* rays_o, float* rays_d, fsn_stream_t stream) {
  FSN_REQUIRE(pose_host && rays_o && rays_d, FSN_E_INVALID, "fsn_get_rays: null pointer");
  FSN_REQUIRE(H > 0 && W > 0 && focal > 0 && row0 >= 0 && nrows >= 0 && row0 + nrows <= H, FSN_E_INVALID,
              "fsn_get_rays: bad geometry H=%d W=%d focal=%g rows [%d,+%d)", H, W, focal, row0, nrows);
  const int64_t npix = (int64_t)nrows * W;
  if (npix == 0) return FSN_OK;
  Pose34 P;
  for (int i = 0; i < 12; ++i) P.m[i] = pose_host[i];
  // W*0.5 and H*0.5 are formed in double by Python and then rounded to float32 (utilities.py:67)
  k_get_rays<<<nblocks(npix, 256), 256, 0, as_stream(stream)>>>(P, W, (float)(W * 0.5), (float)(H * 0.5),
                                                               (float)focal, row0, npix, rays_o, rays_d);
  FSN_LAUNCH_CHECK("k_get_rays");
  return FSN_OK;
}

extern "C" int fsn_to_ndc(const float* rays_o, const float* rays_d, int64_t n, int H, int W, double focal,
                          double near, float* ndc_o, float* ndc_d, fsn_stream_t stream) {
  FSN_REQUIRE(n >= 0 && H > 0 && W > 0 && focal > 0, FSN_E_INVALID, "fsn_to_ndc: bad arguments");
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(rays_o && rays_d && ndc_o && ndc_d, FSN_E_INVALID, "fsn_to_ndc: null pointer");
  const float sx = (float)(-1.0 / (W / (2.0 * focal)));
  const float sy = (float)(-1.0 / (H / (2.0 * focal)));
  k_to_ndc<<<nblocks(n, 256), 256, 0, as_stream(stream)>>>(rays_o, rays_d, n, sx, sy, (float)near,
                                                          (float)(2.0 * near), ndc_o, ndc_d);
  FSN_LAUNCH_CHECK("k_to_ndc");
  return FSN_OK;
}

extern "C" int fsn_build_rays(const float* poses, int64_t n_poses, int H, int W, double focal, int ndc, double near,
                              float* rays_o, float* rays_d, float* aabb, uint32_t* aabb_keys, fsn_stream_t stream) {
  FSN_REQUIRE(n_poses >= 0 && H > 0 && W > 0 && focal > 0, FSN_E_INVALID,
              "fsn_build_rays: bad geometry n=%lld H=%d W=%d focal=%g", (long long)n_poses, H, W, focal);
  if (n_poses == 0) return FSN_OK;
  FSN_REQUIRE(poses && rays_o && rays_d && (!aabb || aabb_keys), FSN_E_INVALID, "fsn_build_rays: null pointer");
  const int64_t total = n_poses * (int64_t)H * W;
  hipStream_t s = as_stream(stream);
  if (aabb) {
    k_aabb_init<<<1, 64, 0, s>>>(aabb_keys);
    FSN_LAUNCH_CHECK("k_aabb_init");
  }
  const float sx = (float)(-1.0 / (W / (2.0 * focal))), sy = (float)(-1.0 / (H / (2.0 * focal)));
  k_build_rays<<<nblocks(total, 256), 256, 0, s>>>(poses, n_poses, H, W, (float)(W * 0.5), (float)(H * 0.5), (float)focal,
                                                   ndc ? 1 : 0, sx, sy, (float)near, (float)(2.0 * near), rays_o, rays_d,
                                                   aabb ? aabb_keys : nullptr);
  FSN_LAUNCH_CHECK("k_build_rays");
  if (aabb) {
    k_aabb_finish<<<1, 64, 0, s>>>(aabb_keys, 8.0f, aabb);  // aabb / 2 ** (4 - 1) (llff.py:84)
    FSN_LAUNCH_CHECK("k_aabb_finish");
  }
  return FSN_OK;
}

extern "C" int fsn_posenc_fwd(const float* x, int64_t n, int d_in, int n_freqs, const float* freqs_host,
                              const float* mask, float* out, fsn_stream_t stream) {
  FSN_REQUIRE(n >= 0 && d_in > 0 && n_freqs >= 0, FSN_E_INVALID, "fsn_posenc_fwd: bad sizes");
  FSN_REQUIRE(n_freqs <= 16, FSN_E_UNSUPPORTED, "fsn_posenc_fwd: n_freqs=%d > 16", n_freqs);
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(x && out && (freqs_host || n_freqs == 0), FSN_E_INVALID, "fsn_posenc_fwd: null pointer");
  Freqs fr{};
  for (int i = 0; i < n_freqs; ++i) fr.f[i] = freqs_host[i];
  const int64_t tot = n * d_in * (1 + 2 * n_freqs);
  k_posenc<<<nblocks(tot, 256), 256, 0, as_stream(stream)>>>(x, n, d_in, n_freqs, fr, mask, out);
  FSN_LAUNCH_CHECK("k_posenc");
  return FSN_OK;
}

extern "C" int fsn_stratified_edges(float near, float far, int S, int64_t R, const float* u, int u_mode,
                                    float* edges, fsn_stream_t stream) {
  FSN_REQUIRE(S > 0 && R >= 0 && u_mode >= 0 && u_mode <= 2, FSN_E_INVALID, "fsn_stratified_edges: bad arguments");
  if (R == 0) return FSN_OK;
  FSN_REQUIRE(edges && (u_mode == 0 || u), FSN_E_INVALID, "fsn_stratified_edges: null pointer");
  const float step = (float)(((double)far - (double)near) / S);
  k_stratified_edges<<<nblocks(R * (S + 1), 256), 256, 0, as_stream(stream)>>>(near, step, S, R, u, u_mode, edges);
  FSN_LAUNCH_CHECK("k_stratified_edges");
  return FSN_OK;
}

extern "C" int fsn_edges_to_packed(const float* edges, int64_t R, int S, int64_t* ray_indices, float* t_starts,
                                   float* t_ends, fsn_stream_t stream) {
  FSN_REQUIRE(S > 0 && R >= 0, FSN_E_INVALID, "fsn_edges_to_packed: bad sizes");
  if (R == 0) return FSN_OK;
  FSN_REQUIRE(edges && ray_indices && t_starts && t_ends, FSN_E_INVALID, "fsn_edges_to_packed: null pointer");
  k_edges_to_packed<<<nblocks(R * S, 256), 256, 0, as_stream(stream)>>>(edges, R, S, ray_indices, t_starts, t_ends);
  FSN_LAUNCH_CHECK("k_edges_to_packed");
  return FSN_OK;
}

extern "C" int fsn_sample_pdf_merge(const float* edges, const float* weights, int64_t R, int S, int n_imp,
                                    const float* u, float* edges_out, fsn_stream_t stream) {
  FSN_REQUIRE(S > 0 && R >= 0 && n_imp >= 0, FSN_E_INVALID, "fsn_sample_pdf_merge: bad sizes");
  if (R == 0) return FSN_OK;
  FSN_REQUIRE(edges && weights && edges_out, FSN_E_INVALID, "fsn_sample_pdf_merge: null pointer");
  const size_t lds = 4 * (size_t)(2 * S + 2 + pow2ceil(n_imp)) * sizeof(float);
  FSN_REQUIRE(lds <= 64 * 1024, FSN_E_UNSUPPORTED, "fsn_sample_pdf_merge: S=%d n_imp=%d too large", S, n_imp);
  k_sample_pdf_merge<<<nblocks(R, 4), 256, lds, as_stream(stream)>>>(edges, weights, R, S, n_imp, u, edges_out);
  FSN_LAUNCH_CHECK("k_sample_pdf_merge");
  return FSN_OK;
}

static Bkgd make_bkgd(const float* b) {
  Bkgd k{};
  if (b) { k.has = 1; k.c[0] = b[0]; k.c[1] = b[1]; k.c[2] = b[2]; }
  return k;
}

extern "C" int fsn_composite_fwd(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                                 int64_t R, int S, const float* bkgd_host, float* colors, float* opacity,
                                 float* depth, float* weights, float* alphas, float* trans, fsn_stream_t stream) {
  FSN_REQUIRE(R >= 0 && S >= 0, FSN_E_INVALID, "fsn_composite_fwd: bad sizes");
  if (R == 0) return FSN_OK;
  FSN_REQUIRE(colors && opacity && depth, FSN_E_INVALID, "fsn_composite_fwd: null output");
  FSN_REQUIRE(S == 0 || (sigmas && rgbs && t_starts && t_ends), FSN_E_INVALID, "fsn_composite_fwd: null input");
  k_composite_dense<<<nblocks(R, 4), 256, 0, as_stream(stream)>>>(sigmas, rgbs, t_starts, t_ends, R, S,
                                                                 make_bkgd(bkgd_host), colors, opacity, depth,
                                                                 weights, alphas, trans);
  FSN_LAUNCH_CHECK("k_composite_dense");
  return FSN_OK;
}

extern "C" int fsn_composite_packed_fwd(const float* sigmas, const float* rgbs, const float* t_starts,
                                        const float* t_ends, const int64_t* ray_indices, int64_t N, int64_t R,
                                        const float* bkgd_host, float* colors, float* opacity, float* depth,
                                        float* weights, float* alphas, float* trans, fsn_stream_t stream) {
  FSN_REQUIRE(R >= 0 && N >= 0, FSN_E_INVALID, "fsn_composite_packed_fwd: bad sizes");
  if (R == 0) return FSN_OK;
  FSN_REQUIRE(colors && opacity && depth, FSN_E_INVALID, "fsn_composite_packed_fwd: null output");
  FSN_REQUIRE(N == 0 || (sigmas && rgbs && t_starts && t_ends && ray_indices), FSN_E_INVALID,
              "fsn_composite_packed_fwd: null input");
  k_composite_packed<<<nblocks(R, 4), 256, 0, as_stream(stream)>>>(sigmas, rgbs, t_starts, t_ends, ray_indices, N, R,
                                                                  make_bkgd(bkgd_host), colors, opacity, depth,
                                                                  weights, alphas, trans);
  FSN_LAUNCH_CHECK("k_composite_packed");
  return FSN_OK;
}

extern "C" int fsn_occlusion_reg_fwd(const float* sigmas, const float* t_vals, const int64_t* ray_idxs, int64_t N,
                                     int64_t n_rays, float a, float b, int func, float* ray_sums, float* out,
                                     fsn_stream_t stream) {
  FSN_REQUIRE(N >= 0 && n_rays >= 0 && (func == 0 || func == 1), FSN_E_INVALID, "fsn_occlusion_reg_fwd: bad arguments");
  FSN_REQUIRE(out && (n_rays == 0 || ray_sums), FSN_E_INVALID, "fsn_occlusion_reg_fwd: null output");
  FSN_REQUIRE(N == 0 || (sigmas && t_vals && ray_idxs), FSN_E_INVALID, "fsn_occlusion_reg_fwd: null input");
  if (n_rays > 0) {
    k_occl_ray_sums<<<nblocks(n_rays, 4), 256, 0, as_stream(stream)>>>(sigmas, t_vals, ray_idxs, N, n_rays, a, b, func,
                                                                     ray_sums);
    FSN_LAUNCH_CHECK("k_occl_ray_sums");
  }
  k_occl_mean<<<1, 256, 0, as_stream(stream)>>>(ray_sums, n_rays, out);
  FSN_LAUNCH_CHECK("k_occl_mean");
  return FSN_OK;
}

extern "C" int fsn_occlusion_reg_bwd(const float* t_vals, int64_t N, const float* ray_sums, int64_t n_rays, float a,
                                     float b, int func, const float* d_out, int32_t* count_ws, float* d_sigmas,
                                     fsn_stream_t stream) {
  FSN_REQUIRE(N >= 0 && n_rays >= 0 && (func == 0 || func == 1), FSN_E_INVALID, "fsn_occlusion_reg_bwd: bad arguments");
  if (N == 0) return FSN_OK;
  FSN_REQUIRE(t_vals && ray_sums && d_out && count_ws && d_sigmas, FSN_E_INVALID, "fsn_occlusion_reg_bwd: null pointer");
  FSN_HIP(hipMemsetAsync(count_ws, 0, sizeof(int32_t), as_stream(stream)));
  k_occl_count<<<nblocks(n_rays, 256), 256, 0, as_stream(stream)>>>(ray_sums, n_rays, count_ws);
  FSN_LAUNCH_CHECK("k_occl_count");
  k_occl_bwd<<<nblocks(N, 256), 256, 0, as_stream(stream)>>>(t_vals, N, a, b, func, d_out, count_ws, d_sigmas);
  FSN_LAUNCH_CHECK("k_occl_bwd");
  return FSN_OK;
}

extern "C" int fsn_to8b(const float* x, int64_t n, uint8_t* out, fsn_stream_t stream) {
  FSN_REQUIRE(n >= 0, FSN_E_INVALID, "fsn_to8b: bad size");
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(x && out, FSN_E_INVALID, "fsn_to8b: null pointer");
  k_to8b<<<nblocks(n, 256), 256, 0, as_stream(stream)>>>(x, n, out);
  FSN_LAUNCH_CHECK("k_to8b");
  return FSN_OK;
}

namespace fsn {
// ---------------------------------------------------------------- f3: video tensors (rendering.py:240-266)
// frames [N,HW,3] float -> uint8 [N,3,HW] = transpose(to8b(frames), (0,3,1,2)); one thread per output byte
// (coalesced byte stores, 12-byte-strided reads served by L2).
__global__ void k_to8b_nchw(const float* __restrict__ x, int64_t n_frames, int64_t hw, uint8_t* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_frames * 3 * hw) return;
  const int64_t f = e / (3 * hw), r = e - f * 3 * hw;
  const int64_t c = r / hw, p = r - c * hw;
  const float v = x[(f * hw + p) * 3 + c];
  out[e] = (uint8_t)(255.0f * fminf(fmaxf(v, 0.0f), 1.0f));
}

// depth [N,HW] -> colours [N,3,HW]: matplotlib's Normalize(vmin, vmax) in float32 followed by the colormap's
// lookup (index = int(x * 256), x == 1 -> 255, below / above range -> first / last entry) in a 256-entry table that
// already holds to8b() of the colormap's RGB (uint8 [256][3]).  vmm = {vmin, vmax} on the device.
__global__ void k_depth_colormap(const float* __restrict__ d, int64_t n_frames, int64_t hw, const float* __restrict__ vmm,
                                 const uint8_t* __restrict__ lut, uint8_t* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_frames * hw) return;
  const int64_t f = e / hw, p = e - f * hw;
  const float vmin = vmm[0], vmax = vmm[1];
  float x = 0.0f;
  if (vmin != vmax) x = (d[e] - vmin) / (vmax - vmin);
  float xa = x * 256.0f;
  if (xa == 256.0f) xa = 255.0f;
  int idx = xa < 0.0f ? 0 : (xa >= 256.0f ? 255 : (int)xa);
  if (!(xa == xa)) idx = 0;  // NaN depth: matplotlib's "bad" colour is transparent black; RGB of entry 0 is kept here
  uint8_t* o = out + f * 3 * hw + p;
  o[0] = lut[3 * idx + 0];
  o[hw] = lut[3 * idx + 1];
  o[2 * hw] = lut[3 * idx + 2];
}

}  // namespace fsn

using namespace fsn;

extern "C" int fsn_to8b_nchw(const float* frames, int64_t n_frames, int64_t hw, uint8_t* out, fsn_stream_t stream) {
  FSN_REQUIRE(n_frames >= 0 && hw >= 0, FSN_E_INVALID, "fsn_to8b_nchw: negative size");
  const int64_t n = n_frames * 3 * hw;
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(frames && out, FSN_E_INVALID, "fsn_to8b_nchw: null pointer");
  k_to8b_nchw<<<nblocks(n, 256), 256, 0, as_stream(stream)>>>(frames, n_frames, hw, out);
  FSN_LAUNCH_CHECK("k_to8b_nchw");
  return FSN_OK;
}

extern "C" int fsn_depth_colormap(const float* depth, int64_t n_frames, int64_t hw, const float* vmin_vmax,
                                  const uint8_t* lut_rgb8, uint8_t* out, fsn_stream_t stream) {
  FSN_REQUIRE(n_frames >= 0 && hw >= 0, FSN_E_INVALID, "fsn_depth_colormap: negative size");
  if (n_frames * hw == 0) return FSN_OK;
  FSN_REQUIRE(depth && vmin_vmax && lut_rgb8 && out, FSN_E_INVALID, "fsn_depth_colormap: null pointer");
  k_depth_colormap<<<nblocks(n_frames * hw, 256), 256, 0, as_stream(stream)>>>(depth, n_frames, hw, vmin_vmax, lut_rgb8, out);
  FSN_LAUNCH_CHECK("k_depth_colormap");
  return FSN_OK;
}

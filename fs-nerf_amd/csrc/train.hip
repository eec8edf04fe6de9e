// train.hip — C-ABI of the TRAINING step around the hot path: NeRF.forward with saved
// activations and its backward on the matrix cores (kernels in train_fused.hip), and the backward of the volume
// integration (kernel below).  The plain fp32 formulation with library GEMMs that the MFMA path was first checked
// against is NOT part of this library any more: it lives in tests/ref_fp32/ as a test-only helper.
//
// reference: src/core/models.py:111-143 (forward), src/run-nerf.py:243-285 (loss.backward()).
#include "common.hpp"
#include "mlp_layout.hpp"
#include "ray_dev.hpp"
#include "train_internal.hpp"

namespace fsn {

// ------------------------------------------------------------------ compositing backward (packed)
// One wavefront per ray.  With q_i = dL/dw_i = g.c_i - g.bkgd + g_opacity:
//   dL/dc_i = w_i g;   dL/dsigma_i = dt_i ( q_i T_i (1 - alpha_i) - sum_{j>i} q_j w_j ).
__global__ void k_composite_packed_bwd(const float* __restrict__ sig, const float* __restrict__ rgb,
                                       const float* __restrict__ t0, const float* __restrict__ t1,
                                       const int64_t* __restrict__ ri, int64_t N, int64_t R, float b0, float b1, float b2,
                                       const float* __restrict__ d_colors, const float* __restrict__ d_opacity,
                                       float* __restrict__ d_sig, float* __restrict__ d_rgb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= R) return;
  int64_t lo = 0, hi = N;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ri[mid] < r) lo = mid + 1; else hi = mid; }
  const int64_t beg = lo;
  hi = N;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ri[mid] < r + 1) lo = mid + 1; else hi = mid; }
  const int S = (int)(lo - beg);
  if (S == 0) return;
  const float* s_ = sig + beg; const float* c_ = rgb + 3 * beg; const float* a_ = t0 + beg; const float* e_ = t1 + beg;
  const float g0 = d_colors[3 * r], g1 = d_colors[3 * r + 1], g2 = d_colors[3 * r + 2];
  const float gop = d_opacity ? d_opacity[r] : 0.f;
  const float gb = g0 * b0 + g1 * b1 + g2 * b2;
  const int per = (S + 63) >> 6;
  const int i0 = lane * per, i1 = min(i0 + per, S);
  // forward quantities: exclusive prefix of sigma*dt
  float lsum = 0.f;
  for (int i = i0; i < i1; ++i) lsum += s_[i] * (e_[i] - a_[i]);
  float tot;
  float run = wave_excl_scan(lsum, tot);
  // pass 1: this lane's sum of q_j w_j; pass 2 needs the suffix sums
  float lq = 0.f;
  {
    float rr = run;
    for (int i = i0; i < i1; ++i) {
      const float sdt = s_[i] * (e_[i] - a_[i]);
      const float w = expf(-rr) * (1.0f - expf(-sdt));
      const float q = (g0 * c_[3 * i] + g1 * c_[3 * i + 1] + g2 * c_[3 * i + 2]) - gb + gop;
      lq += q * w;
      rr += sdt;
    }
  }
  float qtot;
  const float qbefore = wave_excl_scan(lq, qtot);  // sum of q_j w_j over lanes before this one
  float suffix = qtot - qbefore;                    // sum over this lane's samples and all later ones
  for (int i = i0; i < i1; ++i) {
    const float dt = e_[i] - a_[i];
    const float sdt = s_[i] * dt;
    const float T = expf(-run), ea = expf(-sdt);
    const float w = T * (1.0f - ea);
    const float q = (g0 * c_[3 * i] + g1 * c_[3 * i + 1] + g2 * c_[3 * i + 2]) - gb + gop;
    suffix -= q * w;  // now: sum over j > i
    d_sig[beg + i] = dt * (q * T * ea - suffix);
    d_rgb[3 * (beg + i) + 0] = w * g0;
    d_rgb[3 * (beg + i) + 1] = w * g1;
    d_rgb[3 * (beg + i) + 2] = w * g2;
    run += sdt;
  }
}

static inline unsigned nb(int64_t n) { return (unsigned)((n + 255) / 256); }

static int check_desc(const fsn_mlp_desc* d) {
  FSN_REQUIRE(d, FSN_E_INVALID, "null desc");
  NetGeom G;
  const char* why;
  const int rc = build_geom(*d, FSN_PREC_FP16X3, G, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "training path: %s", why);
  return FSN_OK;
}

}  // namespace fsn

using namespace fsn;

extern "C" int64_t fsn_nerf_train_workspace_floats(const fsn_mlp_desc* desc, int prec, int64_t n) {
  if (check_desc(desc) != FSN_OK) return FSN_E_INVALID;
  FSN_REQUIRE(n >= 0, FSN_E_INVALID, "fsn_nerf_train_workspace_floats: n < 0");
  FSN_REQUIRE(prec >= 0 && prec <= FSN_PREC_FP16, FSN_E_INVALID, "fsn_nerf_train_workspace_floats: unknown precision");
  return fused_train_workspace_floats(*desc, prec, n);
}

extern "C" int fsn_nerf_train_fwd(const fsn_mlp_desc* desc, int prec, const float* const* W, const float* const* b,
                                  const float* x, const float* dirs, const float* pos_mask, const float* dir_mask,
                                  int64_t n, float* ws, float* out, uint32_t* status, fsn_stream_t stream) {
  int rc = check_desc(desc);
  if (rc != FSN_OK) return rc;
  FSN_REQUIRE(prec >= 0 && prec <= FSN_PREC_FP16, FSN_E_INVALID, "fsn_nerf_train_fwd: unknown precision");
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(W && b && x && dirs && ws && out, FSN_E_INVALID, "fsn_nerf_train_fwd: null pointer");
  FSN_REQUIRE(n < (1ll << 31), FSN_E_UNSUPPORTED, "fsn_nerf_train_fwd: n too large for one call");
  return fused_train_fwd(desc, prec, W, b, x, dirs, pos_mask, dir_mask, n, ws, out, status, as_stream(stream));
}

extern "C" int fsn_nerf_train_fwd_rays(const fsn_mlp_desc* desc, int prec, const float* const* W, const float* const* b,
                                       const float* rays_o, const float* rays_d, const int64_t* ray_indices,
                                       const float* t_starts, const float* t_ends, const float* pos_mask,
                                       const float* dir_mask, int64_t n, float* ws, float* out, uint32_t* status,
                                       fsn_stream_t stream) {
  int rc = check_desc(desc);
  if (rc != FSN_OK) return rc;
  FSN_REQUIRE(prec >= 0 && prec <= FSN_PREC_FP16, FSN_E_INVALID, "fsn_nerf_train_fwd_rays: unknown precision");
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(W && b && rays_o && rays_d && ray_indices && t_starts && t_ends && ws && out, FSN_E_INVALID,
              "fsn_nerf_train_fwd_rays: null pointer");
  FSN_REQUIRE(n < (1ll << 31), FSN_E_UNSUPPORTED, "fsn_nerf_train_fwd_rays: n too large for one call");
  const TrainRays rays{rays_o, rays_d, t_starts, t_ends, ray_indices};
  return fused_train_fwd(desc, prec, W, b, nullptr, nullptr, pos_mask, dir_mask, n, ws, out, status, as_stream(stream), &rays);
}

extern "C" int fsn_nerf_train_bwd(const fsn_mlp_desc* desc, int prec, const float* const* W, int64_t n, float* ws,
                                  const float* out, const float* d_out, const float* grad_scale, float* const* dW,
                                  float* const* db, int accumulate, float* stage_scales, uint32_t* stage_amax,
                                  uint32_t* status, fsn_stream_t stream) {
  int rc = check_desc(desc);
  if (rc != FSN_OK) return rc;
  FSN_REQUIRE(prec >= 0 && prec <= FSN_PREC_FP16, FSN_E_INVALID, "fsn_nerf_train_bwd: unknown precision");
  FSN_REQUIRE(W && dW && db, FSN_E_INVALID, "fsn_nerf_train_bwd: null pointer");
  FSN_REQUIRE(n > 0 && ws && out && d_out, FSN_E_INVALID, "fsn_nerf_train_bwd: needs the forward's workspace (n > 0)");
  FSN_REQUIRE(n < (1ll << 31), FSN_E_UNSUPPORTED, "fsn_nerf_train_bwd: n too large for one call");
  FSN_REQUIRE((stage_scales == nullptr) == (stage_amax == nullptr), FSN_E_INVALID, "fsn_nerf_train_bwd: stage_scales and stage_amax go together");
  return fused_train_bwd(desc, prec, W, n, ws, out, d_out, grad_scale, dW, db, accumulate != 0, stage_scales, stage_amax, status,
                         as_stream(stream));
}

// max |d_out| as the bits of a non-negative float (unsigned order = float order; a NaN's bits lie above infinity's and
// so survive the maximum), last workgroup forms the scale
__global__ void k_grad_scale(const float* __restrict__ x, int64_t n, float* __restrict__ buf) {
  uint32_t mx = 0u;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) ? n / 4 : 0;  // 16-byte loads over the aligned bulk
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  for (int64_t i = tid; i < n4; i += nthr) {
    const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t b = v[k] & 0x7fffffffu;
      mx = b > mx ? b : mx;
    }
  }
  for (int64_t i = 4 * n4 + tid; i < n; i += nthr) {
    const uint32_t b = __float_as_uint(x[i]) & 0x7fffffffu;
    mx = b > mx ? b : mx;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)mx, m, 64);
    mx = o > mx ? o : mx;
  }
  uint32_t* w = reinterpret_cast<uint32_t*>(buf);
  __shared__ uint32_t last, wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {  // ONE atomic per workgroup (thousands of them on one address serialise: the launch took 50 us)
    const uint32_t a01 = wmax[0] > wmax[1] ? wmax[0] : wmax[1], a23 = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
    atomicMax(w + 1, a01 > a23 ? a01 : a23);
    __threadfence();
    last = atomicAdd(w + 2, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    const float amax = __uint_as_float(atomicMax(w + 1, 0u));
    float e = 0.f;
    const float q = 1024.0f / amax;  // inf for amax = 0 and for subnormal maxima, 0 for inf, NaN for NaN
    if (q > 0.f && q < __builtin_inff()) e = fminf(fmaxf(floorf(log2f(q)), -40.0f), 60.0f);  // (else 0, as nan_to_num did)
    buf[0] = ldexpf(1.0f, (int)e);
  }
}

extern "C" int fsn_grad_scale(const float* d_out, int64_t n, float* buf, fsn_stream_t stream) {
  FSN_REQUIRE(n >= 0 && buf && (n == 0 || d_out), FSN_E_INVALID, "fsn_grad_scale: bad arguments");
  int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
  blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
  k_grad_scale<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(d_out, n, buf);
  FSN_LAUNCH_CHECK("k_grad_scale");
  return FSN_OK;
}

extern "C" int fsn_composite_packed_bwd(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                                        const int64_t* ray_indices, int64_t N, int64_t R, const float* bkgd_host,
                                        const float* d_colors, const float* d_opacity, float* d_sigmas, float* d_rgbs,
                                        fsn_stream_t stream) {
  FSN_REQUIRE(N >= 0 && R >= 0, FSN_E_INVALID, "fsn_composite_packed_bwd: bad sizes");
  if (N == 0 || R == 0) return FSN_OK;
  FSN_REQUIRE(sigmas && rgbs && t_starts && t_ends && ray_indices && d_colors && d_sigmas && d_rgbs, FSN_E_INVALID,
              "fsn_composite_packed_bwd: null pointer");
  const float b0 = bkgd_host ? bkgd_host[0] : 0.f, b1 = bkgd_host ? bkgd_host[1] : 0.f, b2 = bkgd_host ? bkgd_host[2] : 0.f;
  FSN_HIP(hipMemsetAsync(d_sigmas, 0, (size_t)N * sizeof(float), as_stream(stream)));
  FSN_HIP(hipMemsetAsync(d_rgbs, 0, (size_t)N * 3 * sizeof(float), as_stream(stream)));
  k_composite_packed_bwd<<<(unsigned)((R + 3) / 4), 256, 0, as_stream(stream)>>>(sigmas, rgbs, t_starts, t_ends, ray_indices,
                                                                                N, R, b0, b1, b2, d_colors, d_opacity,
                                                                                d_sigmas, d_rgbs);
  FSN_LAUNCH_CHECK("k_composite_packed_bwd");
  return FSN_OK;
}

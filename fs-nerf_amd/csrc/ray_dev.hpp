// ray_dev.hpp — wave-level (64-lane) device routines for the per-ray parts of the path:
// volume integration (prefix-scan alpha compositing) and hierarchical resampling.
// One wavefront owns one ray; used by the standalone kernels in ray_ops.hip and by the
// fused render kernel.  Compiled with -ffp-contract=off so the float op sequence is the
// one written here (it mirrors oracle/fsnerf_oracle.py).
#pragma once
#include "common.hpp"

namespace fsn {

constexpr float kFltEps = 1.1920928955078125e-07f;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// exclusive prefix sum over the 64 lanes (Hillis-Steele on __shfl_up); total returned to all.
__device__ __forceinline__ float wave_excl_scan(float v, float& total) {
  const int lane = lane_id();
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  total = __shfl(inc, 63, 64);
  return inc - v;
}

struct CompositeOut {
  float* colors;   // [3]
  float* opacity;  // [1]
  float* depth;    // [1]
  float* weights;  // [S] or null
  float* alphas;   // [S] or null
  float* trans;    // [S] or null
};

// nerfacc volrend.rendering arithmetic (call site src/render/rendering.py:89-96) for ONE ray
// by ONE wave: dt=t1-t0, alpha=1-exp(-sigma dt), T=exp(-exclusive_sum(sigma dt)), w=T alpha,
// colors=sum w rgb (+bkgd(1-opacity)), opacity=sum w, depth=sum w (t0+t1)/2 / max(opacity,eps).
// Lane l owns the contiguous samples [l*per, (l+1)*per); pointers may be global or LDS.
__device__ __forceinline__ void composite_ray(const float* __restrict__ sig, const float* __restrict__ rgb,
                                              const float* __restrict__ t0, const float* __restrict__ t1,
                                              int S, bool has_bkgd, float b0, float b1, float b2,
                                              const CompositeOut& o) {
  const int lane = lane_id();
  const int per = (S + 63) >> 6;
  const int i0 = lane * per;
  const int i1 = min(i0 + per, S);
  float lsum = 0.f;
  for (int i = i0; i < i1; ++i) lsum += sig[i] * (t1[i] - t0[i]);
  float total;
  float run = wave_excl_scan(lsum, total);
  float ar = 0.f, ag = 0.f, ab = 0.f, ao = 0.f, ad = 0.f;
  for (int i = i0; i < i1; ++i) {
    const float a = t0[i], b = t1[i];
    const float sdt = sig[i] * (b - a);
    const float alpha = 1.0f - expf(-sdt);
    const float T = expf(-run);
    const float w = T * alpha;
    run += sdt;
    ar += w * rgb[3 * i + 0];
    ag += w * rgb[3 * i + 1];
    ab += w * rgb[3 * i + 2];
    ao += w;
    ad += w * (a + b) / 2.0f;
    if (o.weights) o.weights[i] = w;
    if (o.alphas) o.alphas[i] = alpha;
    if (o.trans) o.trans[i] = T;
  }
  ar = wave_sum(ar);
  ag = wave_sum(ag);
  ab = wave_sum(ab);
  ao = wave_sum(ao);
  ad = wave_sum(ad);
  if (lane == 0) {
    const float dep = ad / fmaxf(ao, kFltEps);
    if (has_bkgd) {
      const float k = 1.0f - ao;
      ar = ar + b0 * k;
      ag = ag + b1 * k;
      ab = ab + b2 * k;
    }
    o.colors[0] = ar;
    o.colors[1] = ag;
    o.colors[2] = ab;
    o.opacity[0] = ao;
    o.depth[0] = dep;
  }
}

// weights only (density pass of the hierarchical sampler): w[i] = T_i * alpha_i
__device__ __forceinline__ void weights_ray(const float* __restrict__ sig, const float* __restrict__ edges,
                                            int S, float* __restrict__ w_out) {
  const int lane = lane_id();
  const int per = (S + 63) >> 6;
  const int i0 = lane * per;
  const int i1 = min(i0 + per, S);
  float lsum = 0.f;
  for (int i = i0; i < i1; ++i) lsum += sig[i] * (edges[i + 1] - edges[i]);
  float total;
  float run = wave_excl_scan(lsum, total);
  for (int i = i0; i < i1; ++i) {
    const float sdt = sig[i] * (edges[i + 1] - edges[i]);
    w_out[i] = expf(-run) * (1.0f - expf(-sdt));
    run += sdt;
  }
}

// torch.linspace(0,1,n) element i, float32 (symmetric evaluation like ATen's CPU kernel)
__device__ __forceinline__ float linspace01(int i, int n) {
  if (n <= 1) return 0.f;
  const float step = 1.0f / (float)(n - 1);
  return (i < n / 2) ? (float)i * step : 1.0f - (float)(n - 1 - i) * step;
}

// Hierarchical resampling for ONE ray by ONE wave (build's definition, oracle sample_pdf +
// merge_edges): inverse-CDF samples of pdf=(max(w,0)+1e-5)/sum over `edges`, merged with the
// edges and sorted ascending into out[S+1+n_imp].  cdf_s (>= S+1 floats) and vals_s
// (>= S+1+pow2ceil(n_imp) floats) are this wave's LDS scratch; u is [n_imp] or null (deterministic).
__device__ __forceinline__ void sample_pdf_merge_ray(const float* __restrict__ edges,
                                                     const float* __restrict__ w, int S, int n_imp,
                                                     const float* __restrict__ u, float* cdf_s,
                                                     float* vals_s, float* __restrict__ out) {
  const int lane = lane_id();
  const int per = (S + 63) >> 6;
  const int i0 = lane * per;
  const int i1 = min(i0 + per, S);
  float lsum = 0.f;
  for (int i = i0; i < i1; ++i) lsum += fmaxf(w[i], 0.f) + 1e-5f;
  const float tot = wave_sum(lsum);
  float lp = 0.f;
  for (int i = i0; i < i1; ++i) lp += (fmaxf(w[i], 0.f) + 1e-5f) / tot;
  float dummy;
  float run = wave_excl_scan(lp, dummy);
  for (int i = i0; i < i1; ++i) {
    run += (fmaxf(w[i], 0.f) + 1e-5f) / tot;
    cdf_s[i + 1] = run;
  }
  if (lane == 0) cdf_s[0] = 0.f;
  for (int i = lane; i <= S; i += 64) vals_s[i] = edges[i];
  __builtin_amdgcn_wave_barrier();
  for (int k = lane; k < n_imp; k += 64) {
    const float uk = u ? u[k] : linspace01(k, n_imp);
    int lo = 0, hi = S + 1;  // searchsorted(cdf, u, right=True): #entries <= u
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf_s[mid] <= uk) lo = mid + 1; else hi = mid;
    }
    const int below = max(lo - 1, 0), above = min(lo, S);
    const float c0 = cdf_s[below], c1 = cdf_s[above];
    const float e0 = edges[below], e1 = edges[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    vals_s[S + 1 + k] = e0 + (uk - c0) / denom * (e1 - e0);
  }
  __builtin_amdgcn_wave_barrier();
  float* smp = vals_s + (S + 1);  // the n_imp importance samples
  // Deterministic u is increasing and the inverse CDF is monotone, so the samples are normally sorted
  // already; explicit (random) u leaves them unordered.  Check, and sort only when needed.
  bool unsorted = false;
  for (int k = lane; k + 1 < n_imp; k += 64) unsorted |= smp[k] > smp[k + 1];
  if (__any(unsorted)) {
    // bitonic sort in LDS, padded with +inf to a power of two
    int n2 = 1;
    while (n2 < n_imp) n2 <<= 1;
    for (int k = n_imp + lane; k < n2; k += 64) smp[k] = __builtin_huge_valf();
    __builtin_amdgcn_wave_barrier();
    for (int k = 2; k <= n2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < n2; i += 64) {
          const int l = i ^ j;
          if (l > i) {
            const float a = smp[i], b = smp[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { smp[i] = b; smp[l] = a; }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // Merge of two sorted lists by rank: an edge goes to (its index + #samples strictly below it), a
  // sample to (its index + #edges <= it): a permutation for any ties, O(log n) LDS reads per element.
  for (int i = lane; i <= S; i += 64) {
    const float v = vals_s[i];
    int lo = 0, hi = n_imp;  // lower_bound over the sorted samples
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (smp[mid] < v) lo = mid + 1; else hi = mid;
    }
    out[i + lo] = v;
  }
  for (int k = lane; k < n_imp; k += 64) {
    const float v = smp[k];
    int lo = 0, hi = S + 1;  // upper_bound over the sorted edges
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (vals_s[mid] <= v) lo = mid + 1; else hi = mid;
    }
    out[k + lo] = v;
  }
}

// Pinhole ray of pixel (h, w) (src/utils/utilities.py:57-80): d_cam = [(w - W/2)/f, -(h - H/2)/f, -1], unit length,
// rotated by the pose's 3x3 block (row-major [3][4] in m), origin = its last column.  The one definition used by
// k_get_rays and by the fused render kernel when it generates its own rays.
__device__ __forceinline__ void pinhole_ray(const float* __restrict__ m, float half_w, float half_h, float focal, int h,
                                            int w, float (&o)[3], float (&d)[3]) {
  float dx = ((float)w - half_w) / focal;
  float dy = -((float)h - half_h) / focal;
  float dz = -1.0f;
  const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
  dx = dx / nrm;
  dy = dy / nrm;
  dz = dz / nrm;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    d[k] = (dx * m[4 * k + 0] + dy * m[4 * k + 1]) + dz * m[4 * k + 2];
    o[k] = m[4 * k + 3];
  }
}

// edge i of the fixed-count stratified sampler (oracle.stratified_edges)
__device__ __forceinline__ float stratified_edge(float near, float step, int S, int i, int u_mode,
                                                 const float* __restrict__ u_ray) {
  const float fi = (float)i;
  if (u_mode == 0) return near + fi * step;
  if (u_mode == 1) return near + (fi + u_ray[0]) * step;
  const float lo = near + fmaxf(fi - 0.5f, 0.0f) * step;
  const float hi = near + fminf(fi + 0.5f, (float)S) * step;
  return lo + (hi - lo) * u_ray[i];
}

}  // namespace fsn

// optim.hip — the optimizer side of the training step on ONE flat fp32 parameter arena:
//   * Adam (src/run-nerf.py:217 torch.optim.Adam(params, lr) ... :283 optimizer.step()) as a single launch over the
//     arena (parameters, gradients and both moments are flat buffers; the gradient arena is the one the RCCL
//     all-reduce runs on, so the step needs no concatenation or copy-back);
//   * the weight-norm "frequency" regulariser (src/run-nerf.py:266-279): sum over the selected weight tensors of
//     |w|_1 (reg "l1") or |w|_2 (reg "l2"), as one reduction launch over the arena + a fixed-order finish, and its
//     gradient accumulated straight into the gradient arena.
// HBM-bound elementwise / reduction kernels: 16-byte accesses, grid-stride, no atomics (bit-reproducible).
#include "common.hpp"

#include <cmath>

namespace fsn {

typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int kMaxSeg = 40;
constexpr int kRegChunk = 8192;  // floats reduced by one 256-thread block

struct Segs {
  int32_t n;
  int64_t off[kMaxSeg], len[kMaxSeg];
  int32_t job0[kMaxSeg + 1];  // first chunk-job of each segment
};

// ---------------------------------------------------------------- Adam
// torch.optim.Adam's single-tensor update (torch/optim/adam.py, no amsgrad, maximize=False), operation for operation
// in float32:  g' = g / grad_div (+ weight_decay p);  m = m + (1-b1)(g' - m);  v = b2 v + (1-b2) g' g';
// denom = sqrt(v) / sqrt(1-b2^t) + eps;  p = p - (lr / (1-b1^t)) m / denom.   grad_div: number of ranks when the
// gradient arena holds an all-reduced SUM (folds the 1/world of the data-parallel mean into the step), else 1.
// skip_word / skip_count (device, optional): the step is a no-op - parameters and moments untouched, as when a loss
// scaler does not call optimizer.step() - when bit 0 of *skip_word is set (this rank's training launches of the step
// reported fp16 overflow) or *skip_count > 0 (the flag slot of the all-reduced gradient bucket: some rank did).
// Device-side step counter (fsn_adam_step_dev): one thread decides whether the step runs, advances the counter only
// then, and leaves the step's constants for k_adam - a skipped step does not advance the bias corrections (torch's
// GradScaler does not call optimizer.step() either).  tick: [0] run flag (1.0 / 0.0), [1] step_size, [2] bc2_sqrt.
__global__ void k_adam_tick(int32_t* __restrict__ step_count, float* __restrict__ tick, double lr, double beta1, double beta2,
                            const uint32_t* __restrict__ skip_word, const float* __restrict__ skip_count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool skip = (skip_word && (skip_word[0] & (FSN_STATUS_FP16_RANGE | FSN_STATUS_GRAD_RANGE))) || (skip_count && skip_count[0] > 0.f);
  if (skip) { tick[0] = 0.f; return; }
  const int t = step_count[0] + 1;
  step_count[0] = t;
  // bias corrections in double, as torch forms them in Python floats
  const double b1t = pow(beta1, (double)t), b2t = pow(beta2, (double)t);
  tick[0] = 1.f;
  tick[1] = (float)(lr / (1.0 - b1t));
  tick[2] = (float)sqrt(1.0 - b2t);
}

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       int64_t n, float step_size, float omb1, float b2, float omb2, float bc2_sqrt, float eps,
                       float wd, float grad_div, const uint32_t* __restrict__ skip_word,
                       const float* __restrict__ skip_count, const float* __restrict__ tick) {
  if (tick) {  // (fsn_adam_step_dev: k_adam_tick has looked at the skip words and formed this step's constants)
    if (tick[0] == 0.f) return;
    step_size = tick[1];
    bc2_sqrt = tick[2];
  } else if ((skip_word && (skip_word[0] & (FSN_STATUS_FP16_RANGE | FSN_STATUS_GRAD_RANGE))) || (skip_count && skip_count[0] > 0.f)) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      f32x4 P = *reinterpret_cast<const f32x4*>(p + i), G = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 M = *reinterpret_cast<const f32x4*>(m + i), V = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gk = G[k] / grad_div;
        if (wd != 0.f) gk = gk + wd * P[k];
        M[k] = M[k] + omb1 * (gk - M[k]);
        V[k] = V[k] * b2 + omb2 * gk * gk;
        const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
        P[k] = P[k] - step_size * (M[k] / denom);
      }
      *reinterpret_cast<f32x4*>(p + i) = P;
      *reinterpret_cast<f32x4*>(m + i) = M;
      *reinterpret_cast<f32x4*>(v + i) = V;
    } else {
      for (int64_t j = i; j < n; ++j) {
        float gk = g[j] / grad_div;
        if (wd != 0.f) gk = gk + wd * p[j];
        const float mk = m[j] + omb1 * (gk - m[j]);
        const float vk = v[j] * b2 + omb2 * gk * gk;
        m[j] = mk;
        v[j] = vk;
        p[j] = p[j] - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
      }
    }
  }
}

// ---------------------------------------------------------------- weight-norm regulariser
// job j of segment s covers arena[off[s] + kRegChunk (j - job0[s]) ...): partial[j] = sum |w| (l1) or sum w^2 (l2)
__global__ void k_wnorm_partial(const float* __restrict__ arena, Segs S, int l2, float* __restrict__ partial) {
  __shared__ float red[4];
  const int job = blockIdx.x;
  int s = 0;
  while (s + 1 < S.n && S.job0[s + 1] <= job) ++s;
  const int64_t beg = S.off[s] + (int64_t)(job - S.job0[s]) * kRegChunk;
  const int64_t end = min(beg + kRegChunk, S.off[s] + S.len[s]);
  float acc = 0.f;
  for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    const float w = arena[i];
    acc += l2 ? w * w : fabsf(w);
  }
#pragma unroll
  for (int msk = 32; msk >= 1; msk >>= 1) acc += __shfl_xor(acc, msk, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[job] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one block: per-segment sums in job order, norm per tensor (l2: sqrt), total -> out; seg_norm[s] kept for backward
__global__ void k_wnorm_finish(const float* __restrict__ partial, Segs S, int l2, float* __restrict__ seg_norm,
                               float* __restrict__ out) {
  __shared__ float sn[kMaxSeg];
  const int s = threadIdx.x;
  if (s < S.n) {
    float t = 0.f;
    for (int j = S.job0[s]; j < S.job0[s + 1]; ++j) t += partial[j];
    t = l2 ? sqrtf(t) : t;
    sn[s] = t;
    seg_norm[s] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < S.n; ++i) tot += sn[i];
    out[0] = tot;
  }
}

// grad[i] += d_out * (l1: sign(w), l2: w / |w_s|_2) over the selected segments
__global__ void k_wnorm_bwd(const float* __restrict__ arena, Segs S, int l2, const float* __restrict__ seg_norm,
                            const float* __restrict__ d_out, float* __restrict__ grad) {
  const int job = blockIdx.x;
  int s = 0;
  while (s + 1 < S.n && S.job0[s + 1] <= job) ++s;
  const int64_t beg = S.off[s] + (int64_t)(job - S.job0[s]) * kRegChunk;
  const int64_t end = min(beg + kRegChunk, S.off[s] + S.len[s]);
  const float d = d_out[0];
  const float inv = l2 ? (seg_norm[s] > 0.f ? 1.0f / seg_norm[s] : 0.f) : 0.f;
  for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    const float w = arena[i];
    const float gw = l2 ? w * inv : (w > 0.f ? 1.0f : (w < 0.f ? -1.0f : 0.f));
    grad[i] = grad[i] + d * gw;
  }
}

static int make_segs(int n_seg, const int64_t* off, const int64_t* len, Segs& S, int& n_jobs, const char* who) {
  FSN_REQUIRE(n_seg >= 0 && n_seg <= kMaxSeg && (n_seg == 0 || (off && len)), FSN_E_INVALID, "%s: bad segment table", who);
  S.n = n_seg;
  int j = 0;
  for (int i = 0; i < n_seg; ++i) {
    FSN_REQUIRE(off[i] >= 0 && len[i] > 0, FSN_E_INVALID, "%s: segment %d has offset %lld, length %lld", who, i,
                (long long)off[i], (long long)len[i]);
    S.off[i] = off[i];
    S.len[i] = len[i];
    S.job0[i] = j;
    j += (int)((len[i] + kRegChunk - 1) / kRegChunk);
  }
  S.job0[n_seg] = j;
  n_jobs = j;
  return FSN_OK;
}

}  // namespace fsn

using namespace fsn;

extern "C" int fsn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                             double lr, double beta1, double beta2, double eps, double weight_decay, double grad_div,
                             const uint32_t* skip_word, const float* skip_count, fsn_stream_t stream) {
  FSN_REQUIRE(n >= 0 && step >= 1 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && grad_div > 0, FSN_E_INVALID,
              "fsn_adam_step: bad arguments (n=%lld step=%d)", (long long)n, step);
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(params && grads && exp_avg && exp_avg_sq, FSN_E_INVALID, "fsn_adam_step: null pointer");
  // bias corrections in double, as torch forms them in Python floats
  const double b1t = std::pow(beta1, (double)step), b2t = std::pow(beta2, (double)step);
  const double step_size = lr / (1.0 - b1t);
  const double bc2_sqrt = sqrt(1.0 - b2t);
  int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  const int64_t want = (n / 4 + 255) / 256;
  const unsigned grid = (unsigned)(want < 8 * cus ? (want > 0 ? want : 1) : 8 * cus);
  k_adam<<<grid, 256, 0, as_stream(stream)>>>(params, grads, exp_avg, exp_avg_sq, n, (float)step_size,
                                              (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)bc2_sqrt, (float)eps, (float)weight_decay,
                                              (float)grad_div, skip_word, skip_count, nullptr);
  FSN_LAUNCH_CHECK("k_adam");
  return FSN_OK;
}

extern "C" int fsn_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 int32_t* step_count, float* tick, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, double grad_div, const uint32_t* skip_word, const float* skip_count,
                                 fsn_stream_t stream) {
  FSN_REQUIRE(n >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && grad_div > 0, FSN_E_INVALID,
              "fsn_adam_step_dev: bad arguments (n=%lld)", (long long)n);
  FSN_REQUIRE(step_count && tick, FSN_E_INVALID, "fsn_adam_step_dev: null step_count / tick");
  if (n > 0) FSN_REQUIRE(params && grads && exp_avg && exp_avg_sq, FSN_E_INVALID, "fsn_adam_step_dev: null pointer");
  int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  hipStream_t s = as_stream(stream);
  k_adam_tick<<<1, 64, 0, s>>>(step_count, tick, lr, beta1, beta2, skip_word, skip_count);
  FSN_LAUNCH_CHECK("k_adam_tick");
  if (n == 0) return FSN_OK;
  const int64_t want = (n / 4 + 255) / 256;
  const unsigned grid = (unsigned)(want < 8 * cus ? (want > 0 ? want : 1) : 8 * cus);
  k_adam<<<grid, 256, 0, s>>>(params, grads, exp_avg, exp_avg_sq, n, 0.f, (float)(1.0 - beta1), (float)beta2,
                              (float)(1.0 - beta2), 1.f, (float)eps, (float)weight_decay, (float)grad_div, nullptr, nullptr, tick);
  FSN_LAUNCH_CHECK("k_adam");
  return FSN_OK;
}

extern "C" int64_t fsn_weight_norm_workspace_floats(int n_seg, const int64_t* seg_len_host) {
  FSN_REQUIRE(n_seg >= 0 && n_seg <= kMaxSeg && (n_seg == 0 || seg_len_host), FSN_E_INVALID,
              "fsn_weight_norm_workspace_floats: bad segment table");
  int64_t j = 0;
  for (int i = 0; i < n_seg; ++i) j += (seg_len_host[i] + kRegChunk - 1) / kRegChunk;
  return j + kMaxSeg;  // chunk partials + per-tensor norms
}

extern "C" int fsn_weight_norm_fwd(const float* arena, int n_seg, const int64_t* seg_off_host, const int64_t* seg_len_host,
                                   int l2, float* workspace, float* out, fsn_stream_t stream) {
  Segs S;
  int nj;
  const int rc = make_segs(n_seg, seg_off_host, seg_len_host, S, nj, "fsn_weight_norm_fwd");
  if (rc != FSN_OK) return rc;
  FSN_REQUIRE(out && (n_seg == 0 || (arena && workspace)), FSN_E_INVALID, "fsn_weight_norm_fwd: null pointer");
  hipStream_t s = as_stream(stream);
  if (nj > 0) {
    k_wnorm_partial<<<(unsigned)nj, 256, 0, s>>>(arena, S, l2, workspace);
    FSN_LAUNCH_CHECK("k_wnorm_partial");
  }
  k_wnorm_finish<<<1, 64, 0, s>>>(workspace, S, l2, workspace + nj, out);
  FSN_LAUNCH_CHECK("k_wnorm_finish");
  return FSN_OK;
}

extern "C" int fsn_weight_norm_bwd(const float* arena, int n_seg, const int64_t* seg_off_host, const int64_t* seg_len_host,
                                   int l2, const float* workspace, const float* d_out, float* grad_arena,
                                   fsn_stream_t stream) {
  Segs S;
  int nj;
  const int rc = make_segs(n_seg, seg_off_host, seg_len_host, S, nj, "fsn_weight_norm_bwd");
  if (rc != FSN_OK) return rc;
  if (nj == 0) return FSN_OK;
  FSN_REQUIRE(arena && workspace && d_out && grad_arena, FSN_E_INVALID, "fsn_weight_norm_bwd: null pointer");
  k_wnorm_bwd<<<(unsigned)nj, 256, 0, as_stream(stream)>>>(arena, S, l2, workspace + nj, d_out, grad_arena);
  FSN_LAUNCH_CHECK("k_wnorm_bwd");
  return FSN_OK;
}

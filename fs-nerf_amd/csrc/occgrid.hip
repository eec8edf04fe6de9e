// occgrid.hip — occupancy-grid sampler for the `estimator` slot of render_rays:
// what the reference actually renders with (nerfacc OccGridEstimator; call sites src/render/rendering.py:66-74,
// src/run-nerf.py:96-98, 288-295).  nerfacc's source is not part of the reference, so the arithmetic below is
// THIS build's definition of the same contract ("occupancy sampler"; oracle/fsnerf_oracle.py mirrors it
// operation for operation):
//
//  * grid: `levels` nested boxes around the region of interest (level l = roi scaled by 2^l about its centre),
//    res^3 cells each, cell index (ix*res + iy)*res + iz; `occs` fp32 per cell, `bits` one bit per cell.
//  * march: per ray a lattice t_k = near_r + k*step, near_r = near_plane (+ u_r*step when stratified: the whole
//    march shifts); the interval [t_k, t_k + step) is a sample iff it starts inside [max(t_enter, near_r),
//    min(t_exit, far_plane)) of the outermost box and the cell of its MIDPOINT, at the finest level containing
//    the midpoint, is occupied.  One wavefront per ray, 64 lattice points per iteration, ballot + popcount
//    ranks; two passes (count, fill) around an exclusive scan of the counts.
//  * visibility: T_i = exp(-sum_{j<i} sigma_j dt_j) per ray (prefix scan), keep iff T_i >= early_stop_eps and
//    alpha_i >= alpha_thre.
//  * update: occs[c] = max(occs[c]*decay, occ_c) for the evaluated cells; bit = occs > threshold.
#include "common.hpp"
#include "occ_dev.hpp"
#include "ray_dev.hpp"

namespace fsn {

// pass 0: counts[r]; pass 1: fill ray_indices / t_starts / t_ends at offsets[r]
template <bool FILL>
__global__ void k_occ_march(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t R, GridDev G,
                            const uint32_t* __restrict__ bits, float near_plane, float far_plane, float step,
                            const float* __restrict__ u, int32_t max_steps, int64_t* __restrict__ counts,
                            const int64_t* __restrict__ offsets, int64_t* __restrict__ ray_indices,
                            float* __restrict__ t_starts, float* __restrict__ t_ends) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
  const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
  const RayLattice L = ray_lattice(G, o, d, near_plane, far_plane, step, u != nullptr, u ? u[r] : 0.f);
  const int64_t base_out = FILL ? offsets[r] : 0;
  const int total = march_ray(G, bits, o, d, L, step, max_steps, [&](float ts, float te, bool keep, uint64_t m, int before) {
    if (FILL && keep) {
      const int64_t pos = base_out + before + __popcll(m & ((1ull << lane) - 1ull));
      ray_indices[pos] = r;
      t_starts[pos] = ts;
      t_ends[pos] = te;
    }
  });
  if (!FILL && lane == 0) counts[r] = total;
}

// keep[i] = T_i >= eps && alpha_i >= alpha_thre, packed samples sorted by ray
__global__ void k_visibility(const float* __restrict__ sig, const float* __restrict__ t0, const float* __restrict__ t1,
                             const int64_t* __restrict__ ri, int64_t N, int64_t R, float eps, float alpha_thre,
                             uint8_t* __restrict__ keep) {
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
  const int per = (S + 63) >> 6;
  const int i0 = lane * per, i1 = min(i0 + per, S);
  float lsum = 0.f;
  for (int i = i0; i < i1; ++i) lsum += sig[beg + i] * (t1[beg + i] - t0[beg + i]);
  float tot;
  float run = wave_excl_scan(lsum, tot);
  for (int i = i0; i < i1; ++i) {
    const float sdt = sig[beg + i] * (t1[beg + i] - t0[beg + i]);
    const float T = expf(-run), alpha = 1.0f - expf(-sdt);
    keep[beg + i] = (T >= eps && alpha >= alpha_thre) ? 1 : 0;
    run += sdt;
  }
}

__global__ void k_occ_ema(float* __restrict__ occs, const int64_t* __restrict__ cells, const float* __restrict__ vals,
                          int64_t n, float decay) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cells[i];
  occs[c] = fmaxf(occs[c] * decay, vals[i]);
}

// ---------------------------------------------------------------- update_every_n_steps: cell selection on the device
// (run-nerf.py:288-295 -> OccGridEstimator.update_every_n_steps; round 3 chose the cells with randint / nonzero / unique /
// stack torch ops - two host syncs - and expanded the bit field to bools on every call.)  Past the warm-up an update
// re-evaluates n_uniform cells drawn uniformly and n_occupied cells drawn uniformly from the OCCUPIED ones (both with
// replacement), each at a random point inside the cell.  Counter-based randomness (a 32-bit mixing hash of (seed,
// draw index, stream): no generator state on the device, the oracle restates it): draw i of stream k is
// r(i, k) = mix(mix(i + seed_lo) ^ (seed_hi + 0x9e3779b9 (k + 1))).
__device__ __host__ __forceinline__ uint32_t occ_mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __host__ __forceinline__ uint32_t occ_rand(uint32_t i, uint32_t k, uint32_t seed_lo, uint32_t seed_hi) {
  return occ_mix(occ_mix(i + seed_lo) ^ (seed_hi + 0x9e3779b9u * (k + 1u)));
}

// exclusive prefix of the per-word popcounts of one level's bit field (n_words words): prefix[w], prefix[n_words] = total.
// One block of 1024 threads: contiguous chunks per thread, block scan of the chunk sums.
__global__ __launch_bounds__(1024) void k_occ_word_prefix(const uint32_t* __restrict__ bits, int n_words, int32_t* __restrict__ prefix) {
  __shared__ int32_t part[1024];
  const int per = (n_words + 1023) / 1024;
  const int w0 = threadIdx.x * per, w1 = min(w0 + per, n_words);
  int32_t sum = 0;
  for (int w = w0; w < w1; ++w) sum += __popc(bits[w]);
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    const int32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int32_t run = part[threadIdx.x] - sum;
  for (int w = w0; w < w1; ++w) { prefix[w] = run; run += __popc(bits[w]); }
  if (threadIdx.x == 1023) prefix[n_words] = part[1023];
}

// draw i -> cell of level `lvl` (index inside the level) and a point inside it.  all_cells: draw i IS cell i (warm-up).
__global__ void k_occ_select(const uint32_t* __restrict__ bits, const int32_t* __restrict__ prefix, int res, int64_t lvl_cell0,
                             int64_t n_draws, int64_t n_uniform, int all_cells, uint32_t seed_lo, uint32_t seed_hi,
                             float lox, float loy, float loz, float hix, float hiy, float hiz,
                             int64_t* __restrict__ cells, float* __restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_draws) return;
  const uint32_t res3 = (uint32_t)res * res * res;
  const int n_words = (int)(res3 >> 5);
  uint32_t cell;
  if (all_cells) {
    cell = (uint32_t)i;
  } else {
    const uint32_t r = occ_rand((uint32_t)i, 0u, seed_lo, seed_hi);
    const int32_t total = i < n_uniform ? 0 : prefix[n_words];  // (prefix may be null when nothing is drawn from it)
    if (i < n_uniform || total == 0) {
      cell = r % res3;
    } else {
      const int32_t j = (int32_t)(r % (uint32_t)total);  // the j-th occupied cell of the level
      int lo = 0, hi = n_words;                          // last word w with prefix[w] <= j
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (prefix[mid] <= j) lo = mid; else hi = mid; }
      uint32_t m = bits[lo];
      for (int q = j - prefix[lo]; q > 0; --q) m &= m - 1u;  // drop the lowest set bits below the wanted one
      cell = ((uint32_t)lo << 5) + (uint32_t)(__ffs((int)m) - 1);
    }
  }
  const uint32_t ix = cell / ((uint32_t)res * res), iy = (cell / (uint32_t)res) % (uint32_t)res, iz = cell % (uint32_t)res;
  const float inv24 = 1.0f / 16777216.0f;
  const float u0 = (float)(occ_rand((uint32_t)i, 1u, seed_lo, seed_hi) >> 8) * inv24;
  const float u1 = (float)(occ_rand((uint32_t)i, 2u, seed_lo, seed_hi) >> 8) * inv24;
  const float u2 = (float)(occ_rand((uint32_t)i, 3u, seed_lo, seed_hi) >> 8) * inv24;
  const float fr = (float)res;
  cells[i] = lvl_cell0 + (int64_t)cell;
  x[3 * i + 0] = lox + (((float)ix + u0) / fr) * (hix - lox);
  x[3 * i + 1] = loy + (((float)iy + u1) / fr) * (hiy - loy);
  x[3 * i + 2] = loz + (((float)iz + u2) / fr) * (hiz - loz);
}

// EMA with duplicate cells (draws with replacement): pending[c] = max over the draws that hit c (order-preserving
// integer keys, 0 = untouched), then ONE pass over all cells applies occs = max(occs * decay, pending) and clears it.
__device__ __forceinline__ uint32_t occ_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ void k_occ_scatter_max(uint32_t* __restrict__ pending, const int64_t* __restrict__ cells,
                                  const float* __restrict__ vals, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = vals[i];
  if (v == v) atomicMax(pending + cells[i], occ_key(v));  // (NaN never enters the grid)
}
__global__ void k_occ_ema_pending(float* __restrict__ occs, uint32_t* __restrict__ pending, int64_t n_cells, float decay) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  const uint32_t k = pending[c];
  if (k == 0u) return;
  const float v = __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
  occs[c] = fmaxf(occs[c] * decay, v);
  pending[c] = 0u;
}

// bits word w = cells 32w .. 32w+31; threshold read from device memory (it is a mean computed on the device)
__global__ void k_occ_binarize(const float* __restrict__ occs, int64_t n_cells, const float* __restrict__ thre,
                               uint32_t* __restrict__ bits) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = c < n_cells && occs[c] > thre[0];
  const uint64_t m = __ballot(on);
  const int lane = threadIdx.x & 63;
  if (lane == 0 && c < n_cells) bits[c >> 5] = (uint32_t)m;
  if (lane == 32 && c < n_cells) bits[c >> 5] = (uint32_t)(m >> 32);
}

}  // namespace fsn

using namespace fsn;

extern "C" int fsn_occgrid_march(const float* rays_o, const float* rays_d, int64_t R, const float* aabb_host, int res,
                                 int levels, const uint32_t* bits, float near_plane, float far_plane, float step,
                                 const float* u, int max_steps, int64_t* counts, const int64_t* offsets,
                                 int64_t* ray_indices, float* t_starts, float* t_ends, fsn_stream_t stream) {
  GridDev G;
  const int rc = make_grid(aabb_host, res, levels, G);
  if (rc != FSN_OK) return rc;
  FSN_REQUIRE(R >= 0 && step > 0.f && max_steps > 0, FSN_E_INVALID, "fsn_occgrid_march: bad arguments");
  if (R == 0) return FSN_OK;
  FSN_REQUIRE(rays_o && rays_d && bits, FSN_E_INVALID, "fsn_occgrid_march: null pointer");
  const unsigned grid = (unsigned)((R + 3) / 4);
  if (offsets) {
    FSN_REQUIRE(ray_indices && t_starts && t_ends, FSN_E_INVALID, "fsn_occgrid_march: fill pass needs the outputs");
    k_occ_march<true><<<grid, 256, 0, as_stream(stream)>>>(rays_o, rays_d, R, G, bits, near_plane, far_plane, step, u,
                                                           max_steps, nullptr, offsets, ray_indices, t_starts, t_ends);
  } else {
    FSN_REQUIRE(counts, FSN_E_INVALID, "fsn_occgrid_march: count pass needs `counts`");
    k_occ_march<false><<<grid, 256, 0, as_stream(stream)>>>(rays_o, rays_d, R, G, bits, near_plane, far_plane, step, u,
                                                            max_steps, counts, nullptr, nullptr, nullptr, nullptr);
  }
  FSN_LAUNCH_CHECK("k_occ_march");
  return FSN_OK;
}

extern "C" int fsn_packed_visibility(const float* sigmas, const float* t_starts, const float* t_ends,
                                     const int64_t* ray_indices, int64_t N, int64_t R, float early_stop_eps,
                                     float alpha_thre, uint8_t* keep, fsn_stream_t stream) {
  FSN_REQUIRE(N >= 0 && R >= 0, FSN_E_INVALID, "fsn_packed_visibility: bad sizes");
  if (N == 0 || R == 0) return FSN_OK;
  FSN_REQUIRE(sigmas && t_starts && t_ends && ray_indices && keep, FSN_E_INVALID, "fsn_packed_visibility: null pointer");
  k_visibility<<<(unsigned)((R + 3) / 4), 256, 0, as_stream(stream)>>>(sigmas, t_starts, t_ends, ray_indices, N, R,
                                                                       early_stop_eps, alpha_thre, keep);
  FSN_LAUNCH_CHECK("k_visibility");
  return FSN_OK;
}

extern "C" int fsn_occgrid_update(float* occs, int64_t n_cells, const int64_t* cells, const float* vals, int64_t n,
                                  float decay, const float* threshold_dev, uint32_t* bits, fsn_stream_t stream) {
  FSN_REQUIRE(occs && bits && n_cells > 0 && n_cells % 64 == 0 && n >= 0, FSN_E_INVALID, "fsn_occgrid_update: bad arguments");
  if (n > 0) {
    FSN_REQUIRE(cells && vals, FSN_E_INVALID, "fsn_occgrid_update: null pointer");
    k_occ_ema<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(occs, cells, vals, n, decay);
    FSN_LAUNCH_CHECK("k_occ_ema");
  }
  if (threshold_dev) {
    k_occ_binarize<<<(unsigned)((n_cells + 255) / 256), 256, 0, as_stream(stream)>>>(occs, n_cells, threshold_dev, bits);
    FSN_LAUNCH_CHECK("k_occ_binarize");
  }
  return FSN_OK;
}

extern "C" int fsn_occgrid_select(const uint32_t* bits, int res, int levels, int lvl, const float* aabb_host, int all_cells,
                                  int64_t n_uniform, int64_t n_occupied, uint64_t seed, int32_t* prefix_scratch,
                                  int64_t* cells, float* x, fsn_stream_t stream) {
  GridDev G;
  const int rc = make_grid(aabb_host, res, levels, G);
  if (rc != FSN_OK) return rc;
  FSN_REQUIRE(lvl >= 0 && lvl < levels && n_uniform >= 0 && n_occupied >= 0, FSN_E_INVALID, "fsn_occgrid_select: bad arguments");
  const int64_t res3 = (int64_t)res * res * res;
  FSN_REQUIRE(res3 % 32 == 0 && res3 < (1ll << 31), FSN_E_UNSUPPORTED, "fsn_occgrid_select: res^3 must be a multiple of 32 below 2^31");
  const int64_t n = all_cells ? res3 : n_uniform + n_occupied;
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(bits && cells && x && (all_cells || n_occupied == 0 || prefix_scratch), FSN_E_INVALID, "fsn_occgrid_select: null pointer");
  hipStream_t s = as_stream(stream);
  const uint32_t* lb = bits + lvl * (res3 >> 5);
  if (!all_cells && n_occupied > 0) {
    k_occ_word_prefix<<<1, 1024, 0, s>>>(lb, (int)(res3 >> 5), prefix_scratch);
    FSN_LAUNCH_CHECK("k_occ_word_prefix");
  }
  // level box: the roi scaled by 2^lvl about its centre (OccGridEstimator.level_aabb)
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    const double c = ((double)aabb_host[a] + (double)aabb_host[3 + a]) / 2.0, h = ((double)aabb_host[3 + a] - (double)aabb_host[a]) / 2.0 * (double)(1 << lvl);
    lo[a] = (float)(c - h);
    hi[a] = (float)(c + h);
  }
  k_occ_select<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(lb, prefix_scratch, res, (int64_t)lvl * res3, n,
                                                           (!all_cells && n_occupied == 0) ? n : n_uniform, all_cells ? 1 : 0,
                                                           (uint32_t)seed, (uint32_t)(seed >> 32), lo[0], lo[1], lo[2], hi[0],
                                                           hi[1], hi[2], cells, x);
  FSN_LAUNCH_CHECK("k_occ_select");
  return FSN_OK;
}

extern "C" int fsn_occgrid_update_multi(float* occs, int64_t n_cells, uint32_t* pending, const int64_t* cells,
                                        const float* vals, int64_t n, float decay, fsn_stream_t stream) {
  FSN_REQUIRE(occs && pending && n_cells > 0 && n >= 0, FSN_E_INVALID, "fsn_occgrid_update_multi: bad arguments");
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(cells && vals, FSN_E_INVALID, "fsn_occgrid_update_multi: null pointer");
  hipStream_t s = as_stream(stream);
  k_occ_scatter_max<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(pending, cells, vals, n);
  FSN_LAUNCH_CHECK("k_occ_scatter_max");
  k_occ_ema_pending<<<(unsigned)((n_cells + 255) / 256), 256, 0, s>>>(occs, pending, n_cells, decay);
  FSN_LAUNCH_CHECK("k_occ_ema_pending");
  return FSN_OK;
}

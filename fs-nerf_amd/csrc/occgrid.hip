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

// occ_dev.hpp — device code of the occupancy-grid sampler shared by the standalone kernels (occgrid.hip) and the fused
// occupancy render kernel (render_occ.hip): the grid lookup and the per-ray lattice march, ONE definition so that both
// produce the same samples bit for bit.  The rule itself is this build's definition of the estimator contract
// (occgrid.hip's header; oracle.occgrid_march mirrors it).
#pragma once
#include "common.hpp"

namespace fsn {

struct GridDev {
  float amin[3], amax[3];  // region of interest (level 0)
  int32_t res, levels;
};

inline int make_grid(const float* aabb_host, int res, int levels, GridDev& G) {
  FSN_REQUIRE(aabb_host, FSN_E_INVALID, "occupancy grid: null aabb");
  FSN_REQUIRE(res >= 1 && res <= 1024 && levels >= 1 && levels <= 8, FSN_E_INVALID, "occupancy grid: bad resolution / levels");
  for (int a = 0; a < 3; ++a) {
    G.amin[a] = aabb_host[a];
    G.amax[a] = aabb_host[3 + a];
    FSN_REQUIRE(G.amax[a] > G.amin[a], FSN_E_INVALID, "occupancy grid: empty aabb");
  }
  G.res = res; G.levels = levels;
  return FSN_OK;
}

// occupancy of the cell holding point p: finest level whose box contains p; false outside all boxes
__device__ __forceinline__ bool grid_occupied(const GridDev& G, const uint32_t* __restrict__ bits, float px, float py,
                                              float pz) {
  const float p[3] = {px, py, pz};
  float c[3], h[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    c[a] = (G.amin[a] + G.amax[a]) / 2.0f;
    h[a] = (G.amax[a] - G.amin[a]) / 2.0f;
  }
  float s = 1.0f;
  for (int l = 0; l < G.levels; ++l, s *= 2.0f) {
    bool in = true;
    int ci[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float lo = c[a] - h[a] * s, hi = c[a] + h[a] * s;
      in = in && p[a] >= lo && p[a] <= hi;
      int q = (int)floorf((p[a] - lo) / (hi - lo) * (float)G.res);
      ci[a] = min(max(q, 0), G.res - 1);
    }
    if (in) {
      const int64_t cell = (int64_t)l * G.res * G.res * G.res + ((int64_t)ci[0] * G.res + ci[1]) * G.res + ci[2];
      return (bits[cell >> 5] >> (cell & 31)) & 1u;
    }
  }
  return false;
}

// The lattice of one ray: t_k = near_r + k step, k >= k0, while t_k < t_hi.
struct RayLattice {
  float near_r, t_lo, t_hi;
  int32_t k0;
  bool any;  // the ray meets the outermost box inside [near, far)
};

__device__ __forceinline__ RayLattice ray_lattice(const GridDev& G, const float (&o)[3], const float (&d)[3],
                                                  float near_plane, float far_plane, float step, bool has_u, float u_r) {
  const float sc = (float)(1 << (G.levels - 1));
  float tmin = -__builtin_huge_valf(), tmax = __builtin_huge_valf();
  bool miss = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float c = (G.amin[a] + G.amax[a]) / 2.0f, h = (G.amax[a] - G.amin[a]) / 2.0f * sc;
    const float lo = c - h, hi = c + h;
    if (d[a] == 0.0f) {
      miss = miss || o[a] < lo || o[a] > hi;
    } else {
      const float ta = (lo - o[a]) / d[a], tb = (hi - o[a]) / d[a];
      tmin = fmaxf(tmin, fminf(ta, tb));
      tmax = fminf(tmax, fmaxf(ta, tb));
    }
  }
  RayLattice L;
  L.near_r = has_u ? near_plane + u_r * step : near_plane;
  L.t_lo = fmaxf(tmin, L.near_r);
  L.t_hi = fminf(tmax, far_plane);
  L.any = !miss && L.t_hi > L.t_lo;
  int k0 = L.any ? (int)ceilf((L.t_lo - L.near_r) / step) : 0;
  L.k0 = k0 < 0 ? 0 : k0;
  return L;
}

// One wavefront marches one ray, 64 lattice points per iteration; `sink(ts, te, keep, mask, total)` sees every
// iteration (keep: this lane's point is a sample; mask: ballot of keep; total: samples before this iteration).
// Returns the ray's sample count.
template <class Sink>
__device__ __forceinline__ int march_ray(const GridDev& G, const uint32_t* __restrict__ bits, const float (&o)[3],
                                         const float (&d)[3], const RayLattice& L, float step, int32_t max_steps,
                                         Sink&& sink) {
  const int lane = (int)(threadIdx.x & 63);
  int total = 0;
  if (!L.any) return 0;
  for (int it = 0; it < max_steps; it += 64) {
    const int k = L.k0 + it + lane;
    const float ts = L.near_r + (float)k * step;
    const float te = ts + step;
    const bool in_range = (it + lane) < max_steps && ts >= L.t_lo && ts < L.t_hi;
    bool keep = false;
    if (in_range) {
      const float tm = (ts + te) / 2.0f;
      keep = grid_occupied(G, bits, o[0] + d[0] * tm, o[1] + d[1] * tm, o[2] + d[2] * tm);
    }
    const uint64_t m = __ballot(keep);
    sink(ts, te, keep, m, total);
    total += __popcll(m);
    // wave-uniform exit: the first lane's lattice point of the NEXT iteration is already past the box
    const float ts_next = L.near_r + (float)(L.k0 + it + 64) * step;
    if (!(ts_next < L.t_hi)) break;
  }
  return total;
}

}  // namespace fsn

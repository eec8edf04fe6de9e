// render_occ.hip — the path the reference itself renders with (src/render/rendering.py:58-107 with nerfacc's
// OccGridEstimator in the estimator slot, src/run-nerf.py:96-98, 288-295) in ONE launch, without a host sync for the
// data-dependent sample count:
//   occupancy-grid march -> density pass (sigma_fn) -> visibility cull -> full pass (rgb_sigma_fn) -> packed volume
//   integration,
// per BATCH of rays inside a persistent workgroup.  A batch is as many whole rays as fit the LDS sample list
// (kCap samples, kMaxRays rays), pulled in chunks of up to 8 (one wave marches one ray) from a global work counter, so
// that dense and empty rays balance across the chip; a ray that no longer fits is carried into the next batch.  The
// chunk size is GUIDED (round 4): a pull takes remaining / (2 x workgroups) rays, at most 8 and at least 1, so a
// training-sized launch (4096 rays = 16 per workgroup, one dense ray = up to 8 MLP tiles) ends with single-ray pulls
// instead of 8-ray commitments - the tail of the launch is one ray's work, not one chunk's (~3.5 ms of a 3.5-ms launch
// before).  Rays are independent and integrated by one wave each: the results do not depend on the chunking.  The sample
// list, densities, keep flags, colours live in LDS; the two MLP passes run on the matrix cores over 128-sample tiles
// of the list (mlp_dev.hpp) with the weight stream running on from tile to tile and batch to batch - both kinds of
// tile stream the same blob from its first phase, a tile announces its length when it starts (WStream::begin_tile).
// HBM traffic: the rays in (or none: generated from the pose), 20 B per ray out, the bit grid through L2.
// Every sample is evaluated independently of its tile position and every ray is integrated by one wave with the
// arithmetic of the standalone kernels (occ_dev.hpp, ray_dev.hpp): the image does not depend on how batches form and is
// the unfused path's (march -> k_mlp_fwd -> k_visibility -> k_mlp_fwd -> k_composite_packed).
#include "common.hpp"
#define FSN_KLOOP_ASM
#define FSN_BF16X3_ONEACC 1  // inference: bf16x3 accumulates its three products in one tile (mlp_dev.hpp)
#include "mlp_dev.hpp"
#include "occ_dev.hpp"
#include "ray_dev.hpp"

namespace fsn {

FSN_DEBUG_DEFINE_RECORD(g_dbg_occ)
#define FSN_DEBUG_RECORD g_dbg_occ

constexpr int kCap = 2048;      // samples of a batch (candidates; kept samples are a subset)
constexpr int kMaxRays = 128;   // rays of a batch

struct OccKArgs {
  NetParams net;
  fsn_occ_render_args a;
  GridDev G;
  float cam_hw, cam_hh, cam_f;
  int32_t use_vis;
};

struct OccLds {
  fsn_occ_render_args a;
  GridDev G;
  float cam_hw, cam_hh, cam_f;
  int32_t use_vis;
  // batch state (written by thread 0 between barriers)
  int64_t chunk_base, carry_base;
  int32_t chunk_from, carry_from, chunk_take, carry_take, stop, n_cand, n_kept, n_rays;
  int32_t cnt[kWaves];
  // rays of the batch
  float rays[kMaxRays * 6];
  int64_t ray_id[kMaxRays];
  int32_t cand_off[kMaxRays], cand_cnt[kMaxRays], kept_off[kMaxRays + 1], kept_cnt[kMaxRays];
  // candidate samples (march order: sorted by ray slot, then t)
  float t0c[kCap], sigc[kCap];
  uint16_t slotc[kCap];
  uint8_t keepf[kCap];
  // kept samples
  float t0k[kCap], t1k[kCap], sigk[kCap], rgbk[3 * kCap];
  uint16_t slotk[kCap];
};

// sample groups per wave of a tile (mlp.hip / render.hip): two in the single-pass modes of 256-wide networks
template <int NT, int PREC>
constexpr int groups_per_wave() { return ((PREC & 1) == 1 && NT == 8) ? 2 : 1; }

constexpr int kOccLdsBytes = kRingBytes + (kAuxCapFloats + 96) * 4 + (int)sizeof(OccLds);
static_assert(kOccLdsBytes <= 160 * 1024, "LDS budget");

// sample source: interval [t0, t0 + step) of ray `ray`: x = o + d (t0 + t1) / 2 (rendering.py:59-61, 77-79)
struct OccSrc {
  const float* ray;
  float t0, step;
  __device__ __forceinline__ void pos(float& x, float& y, float& z) const {
    const float tm = t0 + (t0 + step);
    x = ray[0] + ray[3] * tm / 2.0f;
    y = ray[1] + ray[4] * tm / 2.0f;
    z = ray[2] + ray[5] * tm / 2.0f;
  }
  __device__ __forceinline__ void dir(float& x, float& y, float& z) const { x = ray[3]; y = ray[4]; z = ray[5]; }
};

__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// exclusive prefix sum of an int over the 64 lanes; total to all
__device__ __forceinline__ int wave_excl_scan_i(int v, int& total) {
  const int lane = lane_id();
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  total = __shfl(inc, 63, 64);
  return inc - v;
}

template <int NT, int PREC>
__global__ __launch_bounds__(kThreads) void k_render_occ(OccKArgs k) {
  __shared__ __attribute__((aligned(1024))) char smem[kOccLdsBytes];
  float* aux = reinterpret_cast<float*>(smem + kRingBytes);
  OccLds& S = *reinterpret_cast<OccLds*>(smem + kRingBytes + (kAuxCapFloats + 96) * 4);
  int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;  // (laundered per batch below: render.hip)
  if (tid == 0) {
    S.a = k.a; S.G = k.G; S.cam_hw = k.cam_hw; S.cam_hh = k.cam_hh; S.cam_f = k.cam_f; S.use_vis = k.use_vis;
    S.carry_from = kWaves; S.carry_base = 0; S.carry_take = 0;
  }
  __syncthreads();
  const fsn_occ_render_args& a = S.a;
  NetDev net;
  load_net(k.net, a.pos_mask, a.dir_mask, aux, net);
  __syncthreads();
  WStream st;
  st.init(smem, nullptr, 0, 0, k.net.blob + k.net.stream_off, WStream::kDynamicPhases, 1);
  const uint32_t nph_density = (uint32_t)k.net.nph_density, nph_full = (uint32_t)k.net.nph_full;
  ARing ring;
  prime_ring<PREC, NT>(st, ring);

  // this wave's ray of a chunk -> registers (every lane holds the same values)
  auto fetch_ray = [&](int64_t ray, float (&o)[3], float (&d)[3]) {
    if (a.rays_o) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[3 * ray + c]; d[c] = a.rays_d[3 * ray + c]; }
    } else {
      const int h = a.cam_row0 + (int)(ray / a.cam_W), w = (int)(ray % a.cam_W);
      pinhole_ray(a.cam_pose, S.cam_hw, S.cam_hh, S.cam_f, h, w, o, d);
    }
  };

  for (;;) {
#ifndef FSN_NO_LAUNDER_TID
    {
      int t_ = threadIdx.x;
      asm volatile("" : "+v"(t_));
      tid = t_; wave = __builtin_amdgcn_readfirstlane(t_ >> 6); lane = t_ & 63;
    }
#endif
    // ------------------------------------------------------------ build a batch
    if (tid == 0) { S.n_cand = 0; S.n_rays = 0; S.stop = 0; }
    __syncthreads();
    for (;;) {
      if (tid == 0) {
        if (S.carry_from < kWaves) {
          S.chunk_base = S.carry_base; S.chunk_from = S.carry_from; S.chunk_take = S.carry_take; S.carry_from = kWaves;
        } else {
          // guided chunk size from a (possibly stale) look at the queue: remaining / (2 x workgroups), 1 .. 8 rays
          unsigned long long* wc = reinterpret_cast<unsigned long long*>(a.work_counter);
          const long long seen = (long long)__hip_atomic_load(wc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          long long take = (a.R - seen) / (2ll * (long long)gridDim.x);
          take = take < 1 ? 1 : (take > kWaves ? kWaves : take);
          S.chunk_base = (int64_t)atomicAdd(wc, (unsigned long long)take);
          S.chunk_from = 0;
          S.chunk_take = (int32_t)take;
        }
      }
      __syncthreads();
      // (values read back from LDS arrive in VGPRs; these are workgroup-uniform and live across the barriers below:
      // kept scalar, they are not spilled per lane - 8-byte spill stores per lane and batch were 2 GB of scratch
      // writes per frame)
      const int64_t base = uniform_i64(S.chunk_base);
      const int from = __builtin_amdgcn_readfirstlane(S.chunk_from), take = __builtin_amdgcn_readfirstlane(S.chunk_take);
      if (base >= a.R) break;  // (workgroup-uniform: read from LDS)
      const int64_t ray = base + wave;
      const bool active = wave >= from && wave < take && ray < a.R;
      float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f};
      RayLattice L;
      L.any = false; L.near_r = L.t_lo = L.t_hi = 0.f; L.k0 = 0;
      int c = 0;
      if (active) {
        fetch_ray(ray, o, d);
        L = ray_lattice(S.G, o, d, a.near_plane, a.far_plane, a.step, a.u != nullptr, a.u ? a.u[ray] : 0.f);
        c = march_ray(S.G, a.bits, o, d, L, a.step, a.max_steps, [](float, float, bool, uint64_t, int) {});
      }
      if (lane == 0) FSN_AT(S.cnt, wave) = active ? c : -1;
      __syncthreads();
      // accept the chunk's rays in order while they fit (every thread evaluates the same recurrence)
      int nc = S.n_cand, nr = S.n_rays, my_off = -1, my_slot = 0, first_rej = kWaves;
      bool stop = false;
      for (int w = from; w < kWaves; ++w) {
        const int cw = FSN_AT(S.cnt, w);
        if (cw < 0) continue;
        if (!stop && nr < kMaxRays && nc + cw <= kCap) {
          if (w == wave) { my_off = nc; my_slot = nr; }
          nc += cw;
          nr += 1;
        } else if (!stop) {
          stop = true;
          first_rej = w;
        }
      }
      __syncthreads();
      if (tid == 0) {
        S.n_cand = nc; S.n_rays = nr;
        if (stop) { S.carry_base = base; S.carry_from = first_rej; S.carry_take = take; S.stop = 1; }
        else if (take < kWaves) S.stop = 1;  // the queue is running out: this (small) chunk is the whole batch
      }
      if (my_off >= 0) {
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 3; ++q) { FSN_AT(S.rays, 6 * my_slot + q) = o[q]; FSN_AT(S.rays, 6 * my_slot + 3 + q) = d[q]; }
          FSN_AT(S.ray_id, my_slot) = ray;
          FSN_AT(S.cand_off, my_slot) = my_off;
          FSN_AT(S.cand_cnt, my_slot) = c;
        }
        march_ray(S.G, a.bits, o, d, L, a.step, a.max_steps, [&](float ts, float, bool keep, uint64_t m, int before) {
          if (keep) {
            const int pos = my_off + before + __popcll(m & ((1ull << lane) - 1ull));
            FSN_AT(S.t0c, pos) = ts;
            FSN_AT(S.slotc, pos) = (uint16_t)my_slot;
          }
        });
      }
      __syncthreads();
      if (S.stop) break;
    }
    const int n_rays = __builtin_amdgcn_readfirstlane(S.n_rays), n_cand = __builtin_amdgcn_readfirstlane(S.n_cand);
    if (n_rays == 0) break;  // no ray left for this workgroup (uniform)

    // ------------------------------------------------------------ density pass + visibility (estimator.sampling)
    // (single-pass modes of 256-wide networks: 256-sample tiles, two sample groups per wave sharing every A operand -
    // mlp_tile2, the arithmetic per sample is mlp_tile's; this is what the opt-in bf16 visibility cull runs)
    constexpr int NG = groups_per_wave<NT, PREC>(), TILE = 128 * NG;
    if (S.use_vis && n_cand > 0) {
      for (int sub = 0; sub * TILE < n_cand; ++sub) {
        if constexpr (NG == 1) {
          const int idx = sub * 128 + wave * 16 + (lane & 15);
          const int ic = min(idx, n_cand - 1);
          const OccSrc src{FSN_SPAN(S.rays, 6 * FSN_AT(S.slotc, ic), 6), FSN_AT(S.t0c, ic), a.step};
          float sigma, rgb[3];
          st.begin_tile(nph_density);
          mlp_tile<NT, PREC, false>(st, net, src, ring, sigma, rgb);
          if (lane < 16 && idx < n_cand) FSN_AT(S.sigc, idx) = sigma;
        } else {
          const int idx = sub * TILE + wave * 32 + (lane & 15);
          const int i0 = min(idx, n_cand - 1), i1 = min(idx + 16, n_cand - 1);
          const OccSrc src0{FSN_SPAN(S.rays, 6 * FSN_AT(S.slotc, i0), 6), FSN_AT(S.t0c, i0), a.step};
          const OccSrc src1{FSN_SPAN(S.rays, 6 * FSN_AT(S.slotc, i1), 6), FSN_AT(S.t0c, i1), a.step};
          float sigma[2], rgb[2][3];
          st.begin_tile(nph_density);
          mlp_tile2<NT, PREC, false>(st, net, src0, src1, ring, sigma, rgb);
          if (lane < 32) {  // lanes 0-15 write group 0, lanes 16-31 group 1 (every lane holds both results)
            const int q = lane >> 4, iq = idx + 16 * q;
            if (iq < n_cand) FSN_AT(S.sigc, iq) = sigma[q];
          }
        }
      }
      lds_barrier();
      // keep flags per ray: the arithmetic of k_visibility (occgrid.hip), one wave per ray
      for (int r = wave; r < n_rays; r += kWaves) {
        const int Sn = FSN_AT(S.cand_cnt, r), beg = FSN_AT(S.cand_off, r);
        const int per = (Sn + 63) >> 6;
        const int i0 = lane * per, i1 = min(i0 + per, Sn);
        float lsum = 0.f;
        for (int i = i0; i < i1; ++i) lsum += FSN_AT(S.sigc, beg + i) * ((FSN_AT(S.t0c, beg + i) + a.step) - FSN_AT(S.t0c, beg + i));
        float tot;
        float run = wave_excl_scan(lsum, tot);
        int nk = 0;
        for (int i = i0; i < i1; ++i) {
          const float sdt = FSN_AT(S.sigc, beg + i) * ((FSN_AT(S.t0c, beg + i) + a.step) - FSN_AT(S.t0c, beg + i));
          const float T = expf(-run), alpha = 1.0f - expf(-sdt);
          const bool kp = T >= a.early_stop_eps && alpha >= a.alpha_thre;
          FSN_AT(S.keepf, beg + i) = kp ? 1 : 0;
          nk += kp ? 1 : 0;
          run += sdt;
        }
        int tk;
        wave_excl_scan_i(nk, tk);
        if (lane == 0) FSN_AT(S.kept_cnt, r) = tk;
      }
    } else {
      for (int i = tid; i < n_cand; i += kThreads) FSN_AT(S.keepf, i) = 1;
      for (int r = tid; r < n_rays; r += kThreads) FSN_AT(S.kept_cnt, r) = FSN_AT(S.cand_cnt, r);
    }
    lds_barrier();
    if (wave == 0) {  // exclusive scan of the kept counts over the batch's rays (<= 128: two per lane)
      const int c0 = 2 * lane < n_rays ? FSN_AT(S.kept_cnt, 2 * lane) : 0, c1 = 2 * lane + 1 < n_rays ? FSN_AT(S.kept_cnt, 2 * lane + 1) : 0;
      int tot;
      const int ex = wave_excl_scan_i(c0 + c1, tot);
      if (2 * lane < n_rays) FSN_AT(S.kept_off, 2 * lane) = ex;
      if (2 * lane + 1 < n_rays) FSN_AT(S.kept_off, 2 * lane + 1) = ex + c0;
      if (lane == 0) S.n_kept = tot;
    }
    lds_barrier();
    for (int r = wave; r < n_rays; r += kWaves) {  // compaction, order preserved
      const int Sn = FSN_AT(S.cand_cnt, r), beg = FSN_AT(S.cand_off, r), ko = FSN_AT(S.kept_off, r);
      const int per = (Sn + 63) >> 6;
      const int i0 = lane * per, i1 = min(i0 + per, Sn);
      int nk = 0;
      for (int i = i0; i < i1; ++i) nk += FSN_AT(S.keepf, beg + i);
      int tk;
      int pos = ko + wave_excl_scan_i(nk, tk);
      for (int i = i0; i < i1; ++i) {
        if (FSN_AT(S.keepf, beg + i)) {
          const float t0 = FSN_AT(S.t0c, beg + i);
          FSN_AT(S.t0k, pos) = t0;
          FSN_AT(S.t1k, pos) = t0 + a.step;
          FSN_AT(S.slotk, pos) = (uint16_t)r;
          ++pos;
        }
      }
    }
    lds_barrier();
    if (a.sample_t0 && !a.ex_weights) {  // sampler mode: the kept samples are the result
      for (int r = wave; r < n_rays; r += kWaves) {
        const int64_t ray = FSN_AT(S.ray_id, r);
        const int ko = FSN_AT(S.kept_off, r), Sk = FSN_AT(S.kept_cnt, r);
        for (int i = lane; i < Sk; i += 64) a.sample_t0[ray * a.sample_cap + i] = FSN_AT(S.t0k, ko + i);
        if (lane == 0) {
          if (a.n_cand) a.n_cand[ray] = FSN_AT(S.cand_cnt, r);
          a.n_kept[ray] = Sk;
        }
      }
      lds_barrier();
      continue;
    }
    // ------------------------------------------------------------ full pass (rgb_sigma_fn) + packed integration
    const int n_kept = __builtin_amdgcn_readfirstlane(S.n_kept);
    for (int sub = 0; sub * TILE < n_kept; ++sub) {
      if constexpr (NG == 1) {
        const int idx = sub * 128 + wave * 16 + (lane & 15);
        const int ic = min(idx, n_kept - 1);
        const OccSrc src{FSN_SPAN(S.rays, 6 * FSN_AT(S.slotk, ic), 6), FSN_AT(S.t0k, ic), a.step};
        float sigma, rgb[3];
        st.begin_tile(nph_full);
        mlp_tile<NT, PREC, true>(st, net, src, ring, sigma, rgb);
        if (lane < 16 && idx < n_kept) {
          FSN_AT(S.sigk, idx) = sigma;
          FSN_AT(S.rgbk, 3 * idx + 0) = rgb[0];
          FSN_AT(S.rgbk, 3 * idx + 1) = rgb[1];
          FSN_AT(S.rgbk, 3 * idx + 2) = rgb[2];
        }
      } else {
        const int idx = sub * TILE + wave * 32 + (lane & 15);
        const int i0 = min(idx, n_kept - 1), i1 = min(idx + 16, n_kept - 1);
        const OccSrc src0{FSN_SPAN(S.rays, 6 * FSN_AT(S.slotk, i0), 6), FSN_AT(S.t0k, i0), a.step};
        const OccSrc src1{FSN_SPAN(S.rays, 6 * FSN_AT(S.slotk, i1), 6), FSN_AT(S.t0k, i1), a.step};
        float sigma[2], rgb[2][3];
        st.begin_tile(nph_full);
        mlp_tile2<NT, PREC, true>(st, net, src0, src1, ring, sigma, rgb);
        if (lane < 32) {
          const int q = lane >> 4, iq = idx + 16 * q;
          if (iq < n_kept) {
            FSN_AT(S.sigk, iq) = sigma[q];
            FSN_AT(S.rgbk, 3 * iq + 0) = rgb[q][0];
            FSN_AT(S.rgbk, 3 * iq + 1) = rgb[q][1];
            FSN_AT(S.rgbk, 3 * iq + 2) = rgb[q][2];
          }
        }
      }
    }
    lds_barrier();
    for (int r = wave; r < n_rays; r += kWaves) {
      const int64_t ray = FSN_AT(S.ray_id, r);
      const int ko = FSN_AT(S.kept_off, r), Sk = FSN_AT(S.kept_cnt, r);
      // (EXTRAS mode: the per-sample weights / alphas / trans go straight into the ray's slot row)
      const int64_t row = a.ex_weights ? ray * a.sample_cap : 0;
      CompositeOut o{a.colors + 3 * ray, a.opacity + ray, a.depth + ray, a.ex_weights ? a.ex_weights + row : nullptr,
                     a.ex_weights ? a.ex_alphas + row : nullptr, a.ex_weights ? a.ex_trans + row : nullptr};
      composite_ray(FSN_SPAN(S.sigk, ko, Sk), FSN_SPAN(S.rgbk, 3 * ko, 3 * Sk), FSN_SPAN(S.t0k, ko, Sk), FSN_SPAN(S.t1k, ko, Sk), Sk,
                    true, a.bkgd[0], a.bkgd[1], a.bkgd[2], o);
      if (a.ex_weights) {
        for (int i = lane; i < Sk; i += 64) {
          a.sample_t0[row + i] = FSN_AT(S.t0k, ko + i);
          a.ex_sigmas[row + i] = FSN_AT(S.sigk, ko + i);
        }
        for (int i = lane; i < 3 * Sk; i += 64) a.ex_rgbs[3 * row + i] = FSN_AT(S.rgbk, 3 * ko + i);
      }
      if (lane == 0) {
        if (a.n_cand) a.n_cand[ray] = FSN_AT(S.cand_cnt, r);
        if (a.n_kept) a.n_kept[ray] = Sk;
      }
    }
    lds_barrier();
  }
  st.drain();
}

template <int NT, int PREC>
static int launch_occ(const OccKArgs& k, int cus, hipStream_t s) {
  const int64_t chunks = (k.a.R + kWaves - 1) / kWaves;
  const unsigned grid = (unsigned)(chunks < cus ? chunks : cus);
  k_render_occ<NT, PREC><<<grid, kThreads, 0, s>>>(k);
  FSN_LAUNCH_CHECK("k_render_occ");
  return FSN_OK;
}

}  // namespace fsn

namespace fsn {
int debug_report_occ(unsigned* host4) {  // (fsn_debug_report, render.hip)
#ifdef FSN_DEBUG
  unsigned zero[4] = {0u, 0u, 0u, 0u};
  FSN_HIP(hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_dbg_occ), sizeof(unsigned) * 4));
  FSN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_occ), zero, sizeof(zero)));
#else
  for (int i = 0; i < 4; ++i) host4[i] = 0u;
#endif
  return FSN_OK;
}
}  // namespace fsn

using namespace fsn;

extern "C" int fsn_render_rays_occgrid(const fsn_mlp_desc* desc, int prec, const void* blob,
                                       const fsn_occ_render_args* args, fsn_stream_t stream) {
  FSN_REQUIRE(desc && args, FSN_E_INVALID, "fsn_render_rays_occgrid: null pointer");
  const fsn_occ_render_args& a = *args;
  FSN_REQUIRE(a.R >= 0 && a.step > 0.f && a.max_steps > 0, FSN_E_INVALID, "fsn_render_rays_occgrid: bad sizes");
  FSN_REQUIRE(prec >= 0 && prec <= FSN_PREC_FP16X3U, FSN_E_UNSUPPORTED, "fsn_render_rays_occgrid: precision mode %d", prec);
  NetGeom G;
  const char* why;
  int rc = build_geom(*desc, prec, G, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_render_rays_occgrid: %s", why);
  OccKArgs k;
  rc = make_grid(a.aabb, a.res, a.levels, k.G);
  if (rc != FSN_OK) return rc;
  if (a.R == 0) return FSN_OK;
  FSN_REQUIRE(blob && a.bits && a.work_counter, FSN_E_INVALID, "fsn_render_rays_occgrid: null pointer");
  if (a.ex_weights)
    FSN_REQUIRE(a.sample_t0 && a.ex_alphas && a.ex_trans && a.ex_sigmas && a.ex_rgbs && a.colors && a.opacity && a.depth,
                FSN_E_INVALID, "fsn_render_rays_occgrid: the extras mode needs sample_t0, all five ex_* arrays and the per-ray outputs");
  if (a.sample_t0) {
    FSN_REQUIRE(a.n_kept && a.sample_cap >= a.max_steps, FSN_E_INVALID,
                "fsn_render_rays_occgrid: the sampler / extras mode needs n_kept and sample_cap >= max_steps");
  } else {
    FSN_REQUIRE(a.colors && a.opacity && a.depth, FSN_E_INVALID, "fsn_render_rays_occgrid: null output pointer");
  }
  FSN_REQUIRE(a.max_steps <= kCap, FSN_E_UNSUPPORTED,
              "fsn_render_rays_occgrid: max_steps %d > %d samples of one ray group (use the unfused path)", a.max_steps, kCap);
  if (a.rays_o) {
    FSN_REQUIRE(a.rays_d, FSN_E_INVALID, "fsn_render_rays_occgrid: rays_o without rays_d");
  } else {
    FSN_REQUIRE(a.cam_H > 0 && a.cam_W > 0 && a.cam_focal > 0 && a.cam_row0 >= 0 &&
                    a.R <= (int64_t)(a.cam_H - a.cam_row0) * a.cam_W,
                FSN_E_INVALID, "fsn_render_rays_occgrid: no rays and no valid camera");
  }
  FSN_REQUIRE(G.aux_floats <= kAuxCapFloats, FSN_E_UNSUPPORTED, "fsn_render_rays_occgrid: network too deep for LDS");
  NetParams p;
  p.blob = static_cast<const char*>(blob);
  p.aux_off = (int32_t)G.aux_off; p.aux_floats = G.aux_floats; p.stream_off = (int32_t)G.stream_off;
  p.nph_density = G.nph_density; p.nph_full = G.nph_full;
  p.n_layers = desc->n_layers; p.skip_mask = desc->skip_mask;
  p.n_freqs_pos = desc->n_freqs_pos; p.n_freqs_dir = desc->n_freqs_dir;
  p.status = a.status;
  k.net = p;
  k.a = a;
  k.cam_hw = (float)(a.cam_W * 0.5);
  k.cam_hh = (float)(a.cam_H * 0.5);
  k.cam_f = (float)a.cam_focal;
  k.use_vis = (a.early_stop_eps > 0.f || a.alpha_thre > 0.f) ? 1 : 0;
  const int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  hipStream_t s = as_stream(stream);
  FSN_HIP(hipMemsetAsync(a.work_counter, 0, sizeof(unsigned long long), s));
  if (prec == FSN_PREC_FP16X3U) return desc->d_hidden == 256 ? launch_occ<8, 4>(k, cus, s) : launch_occ<4, 4>(k, cus, s);
  const int key = (desc->d_hidden == 256 ? 4 : 0) + prec;
  switch (key) {
    case 0: return launch_occ<4, 0>(k, cus, s);
    case 1: return launch_occ<4, 1>(k, cus, s);
    case 2: return launch_occ<4, 2>(k, cus, s);
    case 3: return launch_occ<4, 3>(k, cus, s);
    case 4: return launch_occ<8, 0>(k, cus, s);
    case 5: return launch_occ<8, 1>(k, cus, s);
    case 6: return launch_occ<8, 2>(k, cus, s);
    default: return launch_occ<8, 3>(k, cus, s);
  }
}

namespace fsn {
struct GatherEx {  // EXTRAS mode: slot rows -> packed arrays (weights, alphas, trans, sigmas; rgbs x3), or all null
  const float* slot[5];
  float* out[5];
};
__global__ void k_occ_gather(const int32_t* __restrict__ n_kept, const int64_t* __restrict__ offsets,
                             const float* __restrict__ t0s, int cap, int64_t R, float step, int64_t* __restrict__ ri,
                             float* __restrict__ ts, float* __restrict__ te, GatherEx ex) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int n = n_kept[r];
  const int64_t off = offsets[r];
  for (int i = lane; i < n; i += 64) {
    const float t = t0s[r * cap + i];
    ri[off + i] = r;
    ts[off + i] = t;
    te[off + i] = t + step;
  }
  if (ex.slot[0]) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int i = lane; i < n; i += 64) ex.out[k][off + i] = ex.slot[k][r * cap + i];
    for (int i = lane; i < 3 * n; i += 64) ex.out[4][3 * off + i] = ex.slot[4][3 * r * cap + i];
  }
}
}  // namespace fsn

extern "C" int fsn_occ_gather_samples(const int32_t* n_kept, const int64_t* offsets, const float* sample_t0, int sample_cap,
                                      int64_t R, float step, int64_t* ray_indices, float* t_starts, float* t_ends,
                                      fsn_stream_t stream) {
  FSN_REQUIRE(R >= 0 && sample_cap > 0, FSN_E_INVALID, "fsn_occ_gather_samples: bad sizes");
  if (R == 0) return FSN_OK;
  FSN_REQUIRE(n_kept && offsets && sample_t0 && ray_indices && t_starts && t_ends, FSN_E_INVALID,
              "fsn_occ_gather_samples: null pointer");
  k_occ_gather<<<(unsigned)((R + 3) / 4), 256, 0, as_stream(stream)>>>(n_kept, offsets, sample_t0, sample_cap, R, step,
                                                                       ray_indices, t_starts, t_ends, GatherEx{});
  FSN_LAUNCH_CHECK("k_occ_gather");
  return FSN_OK;
}

extern "C" int fsn_occ_gather_extras(const int32_t* n_kept, const int64_t* offsets, const float* sample_t0, int sample_cap,
                                     int64_t R, float step, int64_t* ray_indices, float* t_starts, float* t_ends,
                                     const float* const* slots, float* const* out, fsn_stream_t stream) {
  FSN_REQUIRE(R >= 0 && sample_cap > 0, FSN_E_INVALID, "fsn_occ_gather_extras: bad sizes");
  if (R == 0) return FSN_OK;
  FSN_REQUIRE(n_kept && offsets && sample_t0 && ray_indices && t_starts && t_ends && slots && out, FSN_E_INVALID,
              "fsn_occ_gather_extras: null pointer");
  GatherEx ex;
  for (int k = 0; k < 5; ++k) {
    FSN_REQUIRE(slots[k] && out[k], FSN_E_INVALID, "fsn_occ_gather_extras: null slot / output array %d", k);
    ex.slot[k] = slots[k];
    ex.out[k] = out[k];
  }
  k_occ_gather<<<(unsigned)((R + 3) / 4), 256, 0, as_stream(stream)>>>(n_kept, offsets, sample_t0, sample_cap, R, step,
                                                                       ray_indices, t_starts, t_ends, ex);
  FSN_LAUNCH_CHECK("k_occ_gather");
  return FSN_OK;
}

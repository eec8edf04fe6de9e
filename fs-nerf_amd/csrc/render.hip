// render.hip — the whole hot path of render_rays (src/render/rendering.py:25-107) in ONE launch:
//   stratified interval edges -> [density pass of the coarse net -> per-ray weights (wave scan)
//   -> inverse-CDF resampling + sorted union] -> full pass of the fine net -> alpha compositing.
// A workgroup (8 waves, two per SIMD) owns a group of G rays; sample positions, densities, colours, weights
// and the resampled edges live in LDS; the two MLP passes run on the matrix cores
// (mlp_dev.hpp) with the weight stream continuing seamlessly from pass to pass and from ray
// group to ray group; per-ray scans/reductions are done by one wavefront per ray (ray_dev.hpp).
// HBM traffic is the rays in and the requested outputs out.
#include "common.hpp"
// Wave priorities (s_setprio around the k-loop / the pair epilogue, FSN_PRIO) are off: round 1's "+1.8 %" was measured
// with a lane-divergent condition, which compiles to both s_setprio instructions executed by every wave; with the
// wave-uniform form the asymmetric priorities cost 2.3 % (A/B on MI355X, tools/ab_fused.py).
// hand-scheduled k-loop of the x3 modes (kloop_gen.hpp): A operands two units ahead, counted waits
#define FSN_KLOOP_ASM
// waves 4..7 one weight phase behind waves 0..3 (mlp_dev.hpp, kLag): six ring slots
// (measured: 510 ms against 432 ms per frame without it - the workgroup barrier of every phase opening re-joins the
// two groups, so the wave that is not in its epilogue only waits at the next barrier; kept as an experiment switch)
#ifndef FSN_LAG
#define FSN_LAG 0
#endif
#if FSN_LAG
#define FSN_NSLOT 6
#endif
#define FSN_BF16X3_ONEACC 1  // inference: bf16x3 accumulates its three products in one tile (mlp_dev.hpp)
#include "mlp_dev.hpp"
#include "ray_dev.hpp"

namespace fsn {

FSN_DEBUG_DEFINE_RECORD(g_dbg_render)
#define FSN_DEBUG_RECORD g_dbg_render

#ifdef FSN_RENDER_BIG  // experiment: groups of up to 8 rays (158.8 KB of LDS)
constexpr int kMaxGroupSamples = 1536;  // G * (S + n_imp) <= this
constexpr int kMaxGroupCoarse = 768;    // G * S <= this
constexpr int kMaxG = 8;
#else
constexpr int kMaxGroupSamples = 768;  // G * (S + n_imp) <= this
constexpr int kMaxGroupCoarse = 768;   // G * S <= this
constexpr int kMaxG = 4;
#endif
constexpr int kMaxRaySamples = 384;    // S + n_imp <= this
// sample groups of 16 per wave: 2 in the single-pass modes of 256-wide networks (mlp_dev.hpp gemm_layer2), so that a
// workgroup tile is 256 samples and every weight phase - one barrier, one round of LDS-DMA, 16 KiB from L2 - is used
// by twice as many samples
template <int NT, int PREC>
constexpr int groups_per_wave() { return ((PREC & 1) == 1 && NT == 8) ? 2 : 1; }

struct RenderKArgs {
  NetParams netC, netF;
  fsn_render_args a;
  int32_t G, nsubC, nsubF;
  float step;
  int32_t two_phase;
  float cam_hw, cam_hh, cam_f;  // W/2, H/2, focal as float32 (formed in double, utilities.py:67)
};

struct RenderLds {
  // The launch arguments live in LDS, not in SGPRs: they are read where they are used (mostly between the
  // MLP passes) instead of being held - and spilled - across the register-starved MFMA passes.
  fsn_render_args args;
  int32_t Gc, nsubC, nsubF;  // RenderKArgs::G / nsubC / nsubF / step
  float step;
  int32_t two_phase;
  float cam_hw, cam_hh, cam_f;
  float rays[kMaxG * 6];
  float edgesC[kMaxGroupCoarse + kMaxG];
  float sigC[kMaxGroupCoarse];
  float wC[kMaxGroupCoarse];
  float edgesF[kMaxGroupSamples + kMaxG];
  float sigF[kMaxGroupSamples];
  float rgbF[3 * kMaxGroupSamples];
  float cdf[kMaxG][kMaxRaySamples + 1];
  float vals[kMaxG][kMaxRaySamples + 1];
};

constexpr int kNetLdsBytes = (kAuxCapFloats + 96) * 4;
constexpr int kRenderLdsBytes = kRingBytes + 2 * kNetLdsBytes + (int)sizeof(RenderLds);
static_assert(kRenderLdsBytes <= 160 * 1024, "LDS budget");

// sample source of the fused kernel: interval midpoint on the ray, x = o + d*(t0+t1)/2
// (rendering.py:61,79), rebuilt from LDS whenever the MLP asks for it
struct RaySrc {
  const float* ray;  // [ox,oy,oz,dx,dy,dz] in LDS
  const float* e;    // &edges[i] of this sample's interval in LDS
  __device__ __forceinline__ void pos(float& x, float& y, float& z) const {
    const float tm = e[0] + e[1];
    x = ray[0] + ray[3] * tm / 2.0f;
    y = ray[1] + ray[4] * tm / 2.0f;
    z = ray[2] + ray[5] * tm / 2.0f;
  }
  __device__ __forceinline__ void dir(float& x, float& y, float& z) const { x = ray[3]; y = ray[4]; z = ray[5]; }
};

// per-launch parameters of k_render_fused, read from LDS at the point of use
#define GRP_S (a.S)
#define GRP_NI (a.n_imp)
#define GRP_SO (a.S + a.n_imp)
#define GRP_HIER (a.n_imp > 0)
#define GRP_G (S_.Gc)
#define GRP_R (a.R)
#ifdef FSN_STAMP
__device__ unsigned long long g_stamp[256 * 8 * 8];
#endif
template <int NT, int PREC, bool TWO_PHASE>
__global__ __launch_bounds__(kThreads) void k_render_fused(RenderKArgs k) {
#ifdef FSN_STAMP
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
#endif
  // measurement aid (fsn_render_args.clock_out): the clock the chip holds while this launch runs
  uint64_t clk_t0 = 0, clk_r0 = 0;
  const bool clk_on = k.a.clock_out != nullptr && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) < 64;
  if (clk_on) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  auto clk_end = [&]() __attribute__((always_inline)) {
    if (clk_on && (threadIdx.x & 63) == 0) {
      k.a.clock_out[0] += __builtin_amdgcn_s_memtime() - clk_t0;
      k.a.clock_out[1] += __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
  };
  __shared__ __attribute__((aligned(1024))) char smem[kRenderLdsBytes];
  float* auxC = reinterpret_cast<float*>(smem + kRingBytes);
  float* auxF = reinterpret_cast<float*>(smem + kRingBytes + kNetLdsBytes);
  RenderLds& S_ = *reinterpret_cast<RenderLds*>(smem + kRingBytes + 2 * kNetLdsBytes);
  if (threadIdx.x == 0) S_.args = k.a;
  __syncthreads();
  const fsn_render_args& a = S_.args;
  // Loop parameters are read from LDS where they are used (GRP_* below) rather than cached in registers for the
  // whole kernel: 73 fewer spilled SGPRs and 24 fewer spilled VGPRs around the MFMA passes, +2 % (A/B on MI355X).
  if (threadIdx.x == 0) {
    S_.Gc = k.G; S_.nsubC = k.nsubC; S_.nsubF = k.nsubF; S_.step = k.step; S_.two_phase = k.two_phase;
    S_.cam_hw = k.cam_hw; S_.cam_hh = k.cam_hh; S_.cam_f = k.cam_f;
  }
  __syncthreads();
  NetDev netC, netF;
  load_net(k.netF, a.pos_mask, a.dir_mask, auxF, netF);
  if (GRP_HIER) load_net(k.netC, a.pos_mask, a.dir_mask, auxC, netC);
  else netC = netF;
  __syncthreads();
  WStream st;
  {
    // weight-stream schedule: per group "nsubC coarse tiles, nsubF fine tiles", or, in two-phase mode, all of this
    // workgroup's coarse tiles followed by all of its fine tiles
    const int64_t ngr = (k.a.R + k.G - 1) / k.G;
    const int64_t mine = ngr > (int64_t)blockIdx.x ? (ngr - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t repC = k.two_phase ? (uint32_t)(mine * k.nsubC) : (uint32_t)k.nsubC;
    const uint32_t repF = k.two_phase == 2 ? 0u : (k.two_phase ? (uint32_t)(mine * k.nsubF) : (uint32_t)k.nsubF);
    st.init(smem, k.netC.blob + k.netC.stream_off, GRP_HIER ? (uint32_t)k.netC.nph_density : 0u, repC,
            k.netF.blob + k.netF.stream_off, (uint32_t)k.netF.nph_full, repF);
  }
  // The thread id is laundered once per group (an empty asm the compiler cannot see through): lane-derived LDS addresses
  // are then RECOMPUTED per group (a few VALU instructions) instead of being hoisted to the kernel's entry, spilled
  // around the MFMA passes and reloaded per tile - each reload a VMEM load whose compiler-inserted vmcnt(0) also drains
  // the hand-counted LDS-DMA prefetch.  57 -> 20 spilled VGPRs, 144 -> 68 bytes of scratch per lane; fp16x3 frame
  // unchanged (427.2 / 427.5 ms), bf16 frame 137.5 -> 134.8 ms (FSN_NO_LAUNDER_TID: the plain form).
#ifndef FSN_NO_LAUNDER_TID
  int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#define FSN_RELAUNDER()                                              \
  do {                                                               \
    int t_ = threadIdx.x;                                            \
    asm volatile("" : "+v"(t_));                                     \
    tid = t_; wave = __builtin_amdgcn_readfirstlane(t_ >> 6); lane = t_ & 63; \
  } while (0)
#else
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#define FSN_RELAUNDER() do {} while (0)
#endif
  constexpr int NG = groups_per_wave<NT, PREC>(), TILE = 128 * NG;
  ARing ring;
  prime_ring<PREC, NT>(st, ring);
  // ---- the steps of one ray group (G rays), as the kernel strings them together below
  // rays of the group into LDS: from the caller's tensors, or generated from the camera (pose + pixel index)
  // q = n / d for the small non-negative integers of the group bookkeeping (n < 8192, 0 < d <= 512): one float
  // multiply and a fix-up instead of hipcc's ~30-instruction integer division by a run-time divisor ((n + 0.5) / d is at
  // least 1 / (2 d) >= 2^-10 away from every integer, the float product's error is below 8192 * 2^-23 * 2 = 2^-9 .. so
  // the truncation is checked and corrected by one step either way)
  auto small_div = [](int n, int d) __attribute__((always_inline)) {
    int q = (int)(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d));
    q += (q + 1) * d <= n ? 1 : 0;
    q -= q * d > n ? 1 : 0;
    return q;
  };
  auto load_rays = [&](int64_t r0) __attribute__((always_inline)) {
    if (a.rays_o) {
      if (tid < GRP_G * 6) {
        const int g = tid / 6, c = tid - 6 * g;
        const int64_t ray = min(r0 + g, GRP_R - 1);
        FSN_AT(S_.rays, tid) = c < 3 ? a.rays_o[3 * ray + c] : a.rays_d[3 * ray + c - 3];
      }
    } else if (tid < GRP_G) {
      const int64_t ray = min(r0 + tid, GRP_R - 1);
      // camera mode: R <= H W < 2^31 (checked at the entry), so the pixel index divides as 32-bit unsigned by a scalar -
      // the 64-bit division by a divisor read back from LDS was ~100 instructions and the source of in-loop spills
      const uint32_t Wc = (uint32_t)__builtin_amdgcn_readfirstlane(a.cam_W), rr = (uint32_t)ray;
      const int h = a.cam_row0 + (int)(rr / Wc), w = (int)(rr % Wc);
      float o[3], d[3];
      pinhole_ray(a.cam_pose, S_.cam_hw, S_.cam_hh, S_.cam_f, h, w, o, d);
#pragma unroll
      for (int c = 0; c < 3; ++c) { FSN_AT(S_.rays, 6 * tid + c) = o[c]; FSN_AT(S_.rays, 6 * tid + 3 + c) = d[c]; }
    }
  };
  // stratified interval edges -> density pass of the coarse net (sigma_fn, rendering.py:58-64) -> per-ray weights,
  // inverse-CDF resampling, sorted union (one wave per ray) -> S_.edgesF
  auto coarse_stage = [&](int64_t r0) __attribute__((always_inline)) {
    for (int e = tid; e < GRP_G * (GRP_S + 1); e += kThreads) {
      const int g = small_div(e, GRP_S + 1), i = e - g * (GRP_S + 1);
      const int64_t ray = min(r0 + g, GRP_R - 1);
      const float* ur = a.u_mode == 1 ? a.u + ray : (a.u_mode == 2 ? a.u + ray * (GRP_S + 1) : nullptr);
      FSN_AT(S_.edgesC, e) = stratified_edge(a.near, S_.step, GRP_S, i, a.u_mode, ur);
    }
    lds_barrier();
    if (!GRP_HIER) return;
    st.pass_begin();
    for (int sub = 0; sub < S_.nsubC; ++sub) {
      // sample slot of this lane in group q of its wave
      auto slot = [&](int q) { return sub * TILE + (wave * NG + q) * 16 + (lane & 15); };
      auto source = [&](int idx) {
        const int idc = min(idx, GRP_G * GRP_S - 1);
        const int g = small_div(idc, GRP_S), i = idc - g * GRP_S;
        return RaySrc{FSN_SPAN(S_.rays, 6 * g, 6), FSN_SPAN(S_.edgesC, g * (GRP_S + 1) + i, 2)};
      };
      if constexpr (NG == 1) {
        const int idx = slot(0);
        float sigma, rgb[3];
        mlp_tile<NT, PREC, false>(st, netC, source(idx), ring, sigma, rgb);
        if (lane < 16 && idx < GRP_G * GRP_S) FSN_AT(S_.sigC, idx) = sigma;
      } else {
        float sigma[2], rgb[2][3];
        mlp_tile2<NT, PREC, false>(st, netC, source(slot(0)), source(slot(1)), ring, sigma, rgb);
        const int q = (lane >> 4) & 1, idx = slot(q);  // lanes 0-15 store group 0, lanes 16-31 group 1
        if (lane < 32 && idx < GRP_G * GRP_S) FSN_AT(S_.sigC, idx) = sigma[q];
      }
    }
    st.pass_end();
    lds_barrier();
    for (int g = wave; g < GRP_G; g += kWaves) {
      const int64_t ray = min(r0 + g, GRP_R - 1);
      float* wc = FSN_SPAN(S_.wC, g * GRP_S, GRP_S);
      weights_ray(FSN_SPAN(S_.sigC, g * GRP_S, GRP_S), FSN_SPAN(S_.edgesC, g * (GRP_S + 1), GRP_S + 1), GRP_S, wc);
      __builtin_amdgcn_wave_barrier();
      if (a.weights_coarse && r0 + g < GRP_R)
        for (int i = lane; i < GRP_S; i += 64) a.weights_coarse[ray * GRP_S + i] = wc[i];
      sample_pdf_merge_ray(FSN_SPAN(S_.edgesC, g * (GRP_S + 1), GRP_S + 1), wc, GRP_S, GRP_NI,
                           a.u_fine ? a.u_fine + ray * GRP_NI : nullptr, FSN_AT(S_.cdf, g), FSN_AT(S_.vals, g),
                           FSN_SPAN(S_.edgesF, g * (GRP_SO + 1), GRP_SO + 1));
    }
    lds_barrier();
  };
  // full pass of the fine net (rgb_sigma_fn, rendering.py:76-84) on `edges` (LDS) -> volume integration (nerfacc
  // rendering arithmetic, rendering.py:89-96), one wave per ray
  auto fine_stage = [&](int64_t r0, const float* edges, bool write_edges) __attribute__((always_inline)) {
    st.pass_begin();
    for (int sub = 0; sub < S_.nsubF; ++sub) {
      auto slot = [&](int q) { return sub * TILE + (wave * NG + q) * 16 + (lane & 15); };
      auto source = [&](int idx) {
        const int idc = min(idx, GRP_G * GRP_SO - 1);
        const int g = small_div(idc, GRP_SO), i = idc - g * GRP_SO;
        return RaySrc{FSN_SPAN(S_.rays, 6 * g, 6), edges + g * (GRP_SO + 1) + i};
      };
      if constexpr (NG == 1) {
        const int idx = slot(0);
        float sigma, rgb[3];
        mlp_tile<NT, PREC, true>(st, netF, source(idx), ring, sigma, rgb);
        if (lane < 16 && idx < GRP_G * GRP_SO) {
          FSN_AT(S_.sigF, idx) = sigma;
          FSN_AT(S_.rgbF, 3 * idx + 0) = rgb[0];
          FSN_AT(S_.rgbF, 3 * idx + 1) = rgb[1];
          FSN_AT(S_.rgbF, 3 * idx + 2) = rgb[2];
        }
      } else {
        float sigma[2], rgb[2][3];
        mlp_tile2<NT, PREC, true>(st, netF, source(slot(0)), source(slot(1)), ring, sigma, rgb);
        const int q = (lane >> 4) & 1, idx = slot(q);
        if (lane < 32 && idx < GRP_G * GRP_SO) {
          FSN_AT(S_.sigF, idx) = sigma[q];
          FSN_AT(S_.rgbF, 3 * idx + 0) = rgb[q][0];
          FSN_AT(S_.rgbF, 3 * idx + 1) = rgb[q][1];
          FSN_AT(S_.rgbF, 3 * idx + 2) = rgb[q][2];
        }
      }
    }
    st.pass_end();
    lds_barrier();
    for (int g = wave; g < GRP_G; g += kWaves) {
      if (r0 + g >= GRP_R) continue;
      const int64_t ray = r0 + g;
      const float* eg = edges + g * (GRP_SO + 1);
      CompositeOut o{a.colors + 3 * ray, a.opacity + ray, a.depth + ray,
                     a.weights ? a.weights + ray * GRP_SO : nullptr, a.alphas ? a.alphas + ray * GRP_SO : nullptr,
                     a.trans ? a.trans + ray * GRP_SO : nullptr};
      composite_ray(FSN_SPAN(S_.sigF, g * GRP_SO, GRP_SO), FSN_SPAN(S_.rgbF, 3 * g * GRP_SO, 3 * GRP_SO), eg, eg + 1, GRP_SO,
                    true, a.bkgd[0], a.bkgd[1], a.bkgd[2], o);
      if (a.sigmas)
        for (int i = lane; i < GRP_SO; i += 64) a.sigmas[ray * GRP_SO + i] = FSN_AT(S_.sigF, g * GRP_SO + i);
      if (a.rgbs)
        for (int i = lane; i < 3 * GRP_SO; i += 64) a.rgbs[ray * GRP_SO * 3 + i] = FSN_AT(S_.rgbF, 3 * g * GRP_SO + i);
      if (write_edges && a.edges_out)
        for (int i = lane; i <= GRP_SO; i += 64) a.edges_out[ray * (GRP_SO + 1) + i] = eg[i];
    }
    lds_barrier();
  };
  const int64_t ngroups = (GRP_R + GRP_G - 1) / GRP_G;
  if constexpr (!TWO_PHASE) {
    // one group at a time: coarse pass, resampling, fine pass, integration (the weight stream alternates between
    // the two networks; small launches, or no hand-over buffer)
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
      FSN_RELAUNDER();
      const int64_t r0 = grp * (int64_t)__builtin_amdgcn_readfirstlane(GRP_G);  // (G comes from LDS: keep r0 scalar)
      load_rays(r0);
      coarse_stage(r0);
      fine_stage(r0, GRP_HIER ? FSN_SPAN(S_.edgesF, 0, GRP_G * (GRP_SO + 1)) : FSN_SPAN(S_.edgesC, 0, GRP_G * (GRP_SO + 1)), true);
    }
  } else {
    // Two phases (frame-sized hierarchical launches): coarse pass + resampling of ALL groups of this workgroup, the
    // resampled edges handed over through a.edges_out in HBM (772 B per ray at 64+128, written and read once);
    // then the fine pass + integration of all of them.  Every CU of an XCD now streams the same network at the
    // same time, so the XCD's 4 MiB L2 holds one weight stream (2.0 MB coarse, then 2.3 MB fine) instead of
    // thrashing on both; the weight stream's schedule is "coarse x all tiles, then fine x all tiles" (init below).
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
      FSN_RELAUNDER();
      const int64_t r0 = grp * (int64_t)__builtin_amdgcn_readfirstlane(GRP_G);  // (G comes from LDS: keep r0 scalar)
      load_rays(r0);
      coarse_stage(r0);
      for (int e = tid; e < GRP_G * (GRP_SO + 1); e += kThreads) {
        const int g = small_div(e, GRP_SO + 1), i = e - g * (GRP_SO + 1);
        // (streaming: written once here, read once below, 494 MB per 800x800 frame - kept out of the way of the
        // weight streams the XCD's L2 is there for)
#ifdef FSN_EDGES_PLAIN  // experiment: plain (L2 write-back) hand-over stores / loads
        if (r0 + g < GRP_R) a.edges_out[(r0 + g) * (GRP_SO + 1) + i] = FSN_AT(S_.edgesF, e);
#else
        if (r0 + g < GRP_R) __builtin_nontemporal_store(FSN_AT(S_.edgesF, e), a.edges_out + (r0 + g) * (GRP_SO + 1) + i);
#endif
      }
      lds_barrier();
    }
    // The edges are read back by this workgroup only.  Release / acquire at workgroup scope around the barrier: every
    // store of this workgroup is performed before any wave of it reads the buffer back, and no read-back is served
    // from a line fetched before the barrier - stated to the compiler and the hardware instead of relying on the
    // dispatch-start L1 invalidate and on phase 1 never touching the buffer.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (S_.two_phase == 2) {  // sampler only (fsn_render_args.two_phase == 2): the resampled edges are the result
      st.drain();
      clk_end();
      return;
    }
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
      FSN_RELAUNDER();
      const int64_t r0 = grp * (int64_t)__builtin_amdgcn_readfirstlane(GRP_G);  // (G comes from LDS: keep r0 scalar)
      load_rays(r0);
      for (int e = tid; e < GRP_G * (GRP_SO + 1); e += kThreads) {
        const int g = small_div(e, GRP_SO + 1), i = e - g * (GRP_SO + 1);
#ifdef FSN_EDGES_PLAIN
        FSN_AT(S_.edgesF, e) = a.edges_out[min(r0 + g, GRP_R - 1) * (GRP_SO + 1) + i];
#else
        FSN_AT(S_.edgesF, e) = __builtin_nontemporal_load(a.edges_out + min(r0 + g, GRP_R - 1) * (GRP_SO + 1) + i);
#endif
      }
      lds_barrier();
      fine_stage(r0, FSN_SPAN(S_.edgesF, 0, GRP_G * (GRP_SO + 1)), false);
    }
  }
  st.drain();
  clk_end();
#ifdef FSN_STAMP
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) {
    unsigned long long* o = g_stamp + (blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;
    o[0] = __builtin_amdgcn_s_memtime() - t_begin; o[1] = st.t_k; o[2] = st.t_e; o[3] = st.t_n;
    o[4] = st.t_ko; o[5] = st.t_eo;
  }
#endif
}

// ------------------------------------------------------------------ the kernel's own MFMA stream, bare (bench.py)
// fsn_bench_bare_stream: 256 -> 256 hidden layers back to back through gemm_layer - the generated GEMM-pair blocks, the
// pair epilogues, the weight stream with its LDS-DMA ring and barriers, walking the hidden phases of a real blob - on
// fixed activations.  What the render kernels' structure reaches with nothing else in the kernel.
template <int PREC>
__global__ __launch_bounds__(kThreads) void k_bare_stream(const char* stream, uint32_t nph, int layers, uint64_t* clk) {
  __shared__ __attribute__((aligned(1024))) char smem[kRingBytes + 256 * 4];
  float* zeros = reinterpret_cast<float*>(smem + kRingBytes);
  for (int i = threadIdx.x; i < 256; i += blockDim.x) zeros[i] = 0.f;
  __syncthreads();
  NetDev net{};
  net.aux = zeros;  // (biases: zero)
  net.n_layers = 8;
  WStream st;
  st.init(smem, nullptr, 0, 0, stream, nph, 1);
  ARing ring;
  prime_ring<PREC, 8>(st, ring);
  const int g = (threadIdx.x >> 4) & 3;
  // activations as a ReLU layer of the scaled network leaves them: half of them zero, the rest spread over 2^0 .. 2^9,
  // low parts ~2^-11 of the high parts (pseudo-random per lane: the clock the chip holds depends on the data)
  Frag A[8], B[8];
  Frag none[1];
  uint32_t h = 0x9e3779b9u * (threadIdx.x + 1u) + 0x85ebca6bu * (blockIdx.x + 1u);
  auto next = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return h; };
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t r = next();
      const uint16_t hi = (r & 1u) ? (uint16_t)(0x3c00u + ((r >> 1) % 9u) * 0x0400u + ((r >> 8) & 0x3ffu)) : (uint16_t)0;
      const uint16_t lo = (r & 1u) ? (uint16_t)((hi - 0x2c00u) ^ ((r >> 20) & 0x8000u)) : (uint16_t)0;
      A[k].hi[j] = (short)hi;
      A[k].lo[j] = (short)lo;
    }
  Heads heads{0.f, {0.f, 0.f, 0.f}, {0u, 0u, 0u}};
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int l = 0; l < layers; ++l) {
    gemm_layer<PREC, 8, 8, 0, EPI_RELU_CVT>(st, net, 0, A, none, B, heads, ring, g);
    asm volatile("" ::"v"(B[0].hi), "v"(B[7].lo));
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  st.drain();
  if (clk && (threadIdx.x & 63) == 0) {
    uint64_t* o = clk + 2 * (blockIdx.x * kWaves + (threadIdx.x >> 6));
    o[0] = t1 - t0;
    o[1] = r1 - r0;
  }
  if (heads.rs.fall == 0xffffffffu && clk) clk[0] = 0;  // (keeps the range tracking of the epilogues alive)
}

#undef GRP_S
#undef GRP_NI
#undef GRP_SO
#undef GRP_HIER
#undef GRP_G
#undef GRP_R

template <int NT, int PREC>
static int launch_render(RenderKArgs k, int cus, hipStream_t s) {
  // rays per workgroup group: as many as fill one tile of the coarse pass, within the LDS arrays
  // Rays per workgroup group: the per-ray stages (weights, resampling, integration) run one wave per ray, so a group of
  // two rays (one 128-sample tile of a 64-sample coarse pass) leaves six waves idle in them; the x3 modes take two
  // tiles' worth (four rays at 64+128: 424.3 -> 421.7 ms per frame), the two-group single-pass modes already do.
#ifndef FSN_RENDER_G1
#define FSN_RENDER_G1 2
#endif
  constexpr int NG = groups_per_wave<NT, PREC>(), TILE = 128 * NG, CAP = NG == 1 ? 384 * FSN_RENDER_G1 : kMaxGroupSamples;
  const int So = k.a.S + k.a.n_imp;
  int g = (NG == 1 ? FSN_RENDER_G1 : 1) * TILE / k.a.S;
  if (g < 1) g = 1;
  if (g > CAP / So) g = CAP / So;
  if (g > kMaxG) g = kMaxG;
  if (g > kMaxGroupCoarse / k.a.S) g = kMaxGroupCoarse / k.a.S;
  if (g < 1) g = 1;
  k.G = g;
  k.nsubC = (g * k.a.S + TILE - 1) / TILE;
  k.nsubF = (g * So + TILE - 1) / TILE;
  const int64_t ngroups = (k.a.R + k.G - 1) / k.G;
  const unsigned grid = (unsigned)(ngroups < cus ? ngroups : cus);
  // (two instantiations: the two-phase variant's second copy of both MLP passes costs registers and code the usual
  // launch should not pay for)
  if (k.two_phase) k_render_fused<NT, PREC, true><<<grid, kThreads, 0, s>>>(k);
  else k_render_fused<NT, PREC, false><<<grid, kThreads, 0, s>>>(k);
  FSN_LAUNCH_CHECK("k_render_fused");
  return FSN_OK;
}

static NetParams net_params(const fsn_mlp_desc& d, const NetGeom& G, const void* blob, uint32_t* status) {
  NetParams p;
  p.blob = static_cast<const char*>(blob);
  p.aux_off = (int32_t)G.aux_off; p.aux_floats = G.aux_floats; p.stream_off = (int32_t)G.stream_off;
  p.nph_density = G.nph_density; p.nph_full = G.nph_full;
  p.n_layers = d.n_layers; p.skip_mask = d.skip_mask;
  p.n_freqs_pos = d.n_freqs_pos; p.n_freqs_dir = d.n_freqs_dir;
  p.status = status;
  return p;
}

}  // namespace fsn

using namespace fsn;

#ifdef FSN_STAMP
extern "C" int fsn_dbg_stamps(unsigned long long* host_out) {
  FSN_HIP(hipDeviceSynchronize());
  FSN_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 256 * 8 * 8));
  return FSN_OK;
}
#endif

extern "C" int fsn_render_rays_fused(const fsn_mlp_desc* desc, int prec, const void* blob_coarse,
                                     const void* blob_fine, const fsn_render_args* args, fsn_stream_t stream) {
  FSN_REQUIRE(desc && args, FSN_E_INVALID, "fsn_render_rays_fused: null pointer");
  const fsn_render_args& a = *args;
  FSN_REQUIRE(a.R >= 0 && a.S > 0 && a.n_imp >= 0 && a.u_mode >= 0 && a.u_mode <= 2, FSN_E_INVALID,
              "fsn_render_rays_fused: bad sizes R=%lld S=%d n_imp=%d u_mode=%d", (long long)a.R, a.S, a.n_imp, a.u_mode);
  NetGeom G;
  const char* why;
  const int rc = build_geom(*desc, prec, G, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_render_rays_fused: %s", why);
  if (a.R == 0) return FSN_OK;
  const bool sampler_only = a.two_phase == 2;
  if (sampler_only) {
    FSN_REQUIRE(a.n_imp > 0 && blob_coarse && a.edges_out, FSN_E_INVALID,
                "fsn_render_rays_fused: two_phase == 2 (sampler only) needs n_imp > 0, blob_coarse and edges_out");
    if (!blob_fine) blob_fine = blob_coarse;
  } else {
    FSN_REQUIRE(blob_fine && a.colors && a.opacity && a.depth, FSN_E_INVALID, "fsn_render_rays_fused: null pointer");
  }
  if (a.rays_o) {
    FSN_REQUIRE(a.rays_d, FSN_E_INVALID, "fsn_render_rays_fused: rays_o without rays_d");
  } else {
    FSN_REQUIRE(a.cam_H > 0 && a.cam_W > 0 && a.cam_focal > 0 && a.cam_row0 >= 0 &&
                    (int64_t)a.cam_H * a.cam_W < (1ll << 31) && a.R <= (int64_t)(a.cam_H - a.cam_row0) * a.cam_W,
                FSN_E_INVALID, "fsn_render_rays_fused: no rays and no valid camera (H=%d W=%d focal=%g row0=%d R=%lld)",
                a.cam_H, a.cam_W, a.cam_focal, a.cam_row0, (long long)a.R);
  }
  FSN_REQUIRE(a.n_imp == 0 || blob_coarse, FSN_E_INVALID, "fsn_render_rays_fused: hierarchical sampling needs blob_coarse");
  FSN_REQUIRE(a.u_mode == 0 || a.u, FSN_E_INVALID, "fsn_render_rays_fused: u_mode %d needs u", a.u_mode);
  const int So = a.S + a.n_imp;
  FSN_REQUIRE(So <= kMaxRaySamples, FSN_E_UNSUPPORTED, "fsn_render_rays_fused: S+n_imp=%d > %d", So, kMaxRaySamples);
  {
    int n2 = 1;
    while (n2 < a.n_imp) n2 <<= 1;  // the resampler's sort scratch is padded to a power of two
    FSN_REQUIRE(a.S + n2 <= kMaxRaySamples, FSN_E_UNSUPPORTED, "fsn_render_rays_fused: S + pow2ceil(n_imp) = %d > %d",
                a.S + n2, kMaxRaySamples);
  }
  FSN_REQUIRE(G.aux_floats <= kAuxCapFloats, FSN_E_UNSUPPORTED, "fsn_render_rays_fused: network too deep for LDS");
  RenderKArgs k;
  k.netF = net_params(*desc, G, blob_fine, a.status);
  k.netC = net_params(*desc, G, a.n_imp > 0 ? blob_coarse : blob_fine, a.status);
  k.a = a;
  k.G = k.nsubC = k.nsubF = 0;  // (set per instantiation in launch_render: the tile is 128 or 256 samples)
  k.step = (float)(((double)a.far - (double)a.near) / a.S);
  k.two_phase = sampler_only ? 2 : ((a.two_phase && a.n_imp > 0 && a.edges_out) ? 1 : 0);
  k.cam_hw = (float)(a.cam_W * 0.5);
  k.cam_hh = (float)(a.cam_H * 0.5);
  k.cam_f = (float)a.cam_focal;
  const int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  hipStream_t s = as_stream(stream);
  if (prec == FSN_PREC_FP16X2) return desc->d_hidden == 256 ? launch_render<8, 6>(k, cus, s) : launch_render<4, 6>(k, cus, s);
  if (prec == FSN_PREC_FP16X3U) return desc->d_hidden == 256 ? launch_render<8, 4>(k, cus, s) : launch_render<4, 4>(k, cus, s);
  const int key = (desc->d_hidden == 256 ? 4 : 0) + prec;
  switch (key) {
    case 0: return launch_render<4, 0>(k, cus, s);
    case 1: return launch_render<4, 1>(k, cus, s);
    case 2: return launch_render<4, 2>(k, cus, s);
    case 3: return launch_render<4, 3>(k, cus, s);
    case 4: return launch_render<8, 0>(k, cus, s);
    case 5: return launch_render<8, 1>(k, cus, s);
    case 6: return launch_render<8, 2>(k, cus, s);
    default: return launch_render<8, 3>(k, cus, s);
  }
}

extern "C" int fsn_bench_bare_stream(const fsn_mlp_desc* desc, int prec, const void* blob, int layers, uint64_t* clock_out,
                                     fsn_stream_t stream) {
  FSN_REQUIRE(desc && blob && layers > 0, FSN_E_INVALID, "fsn_bench_bare_stream: bad arguments");
  FSN_REQUIRE(desc->d_hidden == 256 && (prec == FSN_PREC_BF16X3 || prec == FSN_PREC_FP16X3 || prec == FSN_PREC_FP16X3U),
              FSN_E_UNSUPPORTED, "fsn_bench_bare_stream: d_hidden 256 in an x3 mode only");
  NetGeom G;
  const char* why;
  const int rc = build_geom(*desc, prec, G, &why);
  FSN_REQUIRE(rc == FSN_OK, rc, "fsn_bench_bare_stream: %s", why);
  const int cus = fsn_device_cus();
  if (cus <= 0) return FSN_E_HIP;
  const char* sp = static_cast<const char*>(blob) + G.stream_off;
  hipStream_t s = as_stream(stream);
  if (prec == FSN_PREC_BF16X3) k_bare_stream<0><<<cus, kThreads, 0, s>>>(sp, (uint32_t)G.nph_density, layers, clock_out);
  else if (prec == FSN_PREC_FP16X3) k_bare_stream<2><<<cus, kThreads, 0, s>>>(sp, (uint32_t)G.nph_density, layers, clock_out);
  else k_bare_stream<4><<<cus, kThreads, 0, s>>>(sp, (uint32_t)G.nph_density, layers, clock_out);
  FSN_LAUNCH_CHECK("k_bare_stream");
  return cus;
}

namespace fsn { int debug_report_occ(unsigned* host4); }  // render_occ.hip

#ifdef FSN_DEBUG
namespace fsn {
// negative control of the debug build's checks: one deliberate out-of-range index and one out-of-range span
__global__ void k_debug_selftest(int n) {
  __shared__ float arr[32];
  arr[threadIdx.x & 31] = 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    FSN_AT(arr, n) = 1.f;              // n = 32: one past the end
    float* p = FSN_SPAN(arr, 30, 4);   // [30, 34) of 32
    p[0] = 2.f;
  }
}
}  // namespace fsn
#endif

extern "C" int fsn_debug_selftest(void) {
#ifdef FSN_DEBUG
  fsn::k_debug_selftest<<<1, 64>>>(32);
  FSN_LAUNCH_CHECK("k_debug_selftest");
  return FSN_OK;
#else
  FSN_REQUIRE(false, FSN_E_UNSUPPORTED, "fsn_debug_selftest: not a debug build");
#endif
}

extern "C" int fsn_debug_report(uint32_t* out_host) {
  FSN_REQUIRE(out_host, FSN_E_INVALID, "fsn_debug_report: null pointer");
#ifdef FSN_DEBUG
  FSN_HIP(hipDeviceSynchronize());
  unsigned zero[4] = {0u, 0u, 0u, 0u};
  FSN_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_dbg_render), sizeof(unsigned) * 4));
  FSN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_render), zero, sizeof(zero)));
  return fsn::debug_report_occ(out_host + 4);
#else
  FSN_REQUIRE(false, FSN_E_UNSUPPORTED, "fsn_debug_report: not a debug build (make -C fs-nerf_amd/csrc debug)");
#endif
}

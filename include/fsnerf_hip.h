/*
 * fsnerf_hip.h — C-ABI of libfsnerf_hip.so: the NeRF ray-rendering hot path of
 * a-lemus96/fs-nerf as hand-written HIP kernels for MI355X (gfx950 / CDNA4).
 *
 * The reference has no FFI of its own (it is 100 % Python): the path sits
 * behind Python callables.  Each entry point below therefore names the reference callable
 * (file:line under /root/reference) whose arithmetic it replaces; the Python host layer in
 * fs-nerf_amd/ keeps those callables' names and signatures and binds this library with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - tensors are dense row-major float32 unless stated; ray_indices is int64
 *     (reference contract, src/render/rendering.py:66,92);
 *   - the caller owns all memory; nothing is retained past the call's stream work;
 *   - all calls are asynchronous on `stream` (a hipStream_t; NULL = default stream);
 *   - return 0 on success, negative on error (FSN_E_*); the message is available from
 *     fsn_last_error() (thread-local).  Zero-sample / zero-ray inputs are not errors.
 */
#ifndef FSNERF_HIP_H
#define FSNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fsn_stream_t; /* hipStream_t */

#define FSN_OK 0
#define FSN_E_INVALID (-1)     /* bad argument (null pointer, negative size, ...) */
#define FSN_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define FSN_E_HIP (-3)         /* HIP runtime error (launch, no device, ...) */

/* arithmetic of the MLP contractions */
#define FSN_PREC_BF16X3 0 /* split-bf16 (hi+lo) x 3 MFMA passes, fp32 accumulate: ~1e-5 relative per product */
#define FSN_PREC_BF16 1   /* single bf16 MFMA pass, fp32 accumulate (BASELINE config 5).  The single-pass modes round
                             every operand to 16 bits and evaluate the encodings' sines with the hardware
                             instruction (|err| ~1e-4 rad, below that rounding) */
#define FSN_PREC_FP16X3 2 /* split-fp16 (hi + 2^-11 lo') x 3 MFMA passes, the two correction products in their own
                             accumulator: fp32-grade accuracy for layer scales from 2^-14 (6.1e-5) up to 65504 (both ends
                             reported through `status`, below) - the default "parity" mode */
#define FSN_PREC_FP16 3   /* single fp16 MFMA pass */
#define FSN_PREC_FP16X3U 4 /* round 4, inference entry points only: split-fp16 x 3 MFMA passes with UNSCALED low parts,
                             the three products of a unit in one accumulator, no merge in the epilogue (the arithmetic of
                             rounds 1-2: 6 % faster than FSN_PREC_FP16X3).  float32-grade because every layer's
                             activations are kept at 2^4 .. 2^10 by a power-of-two scale per layer that is folded into
                             the packed weights (fsn_mlp_pack_scaled; an exact transformation of the network, the outputs
                             are unscaled by the head weights) and calibrated from the layers' measured maxima
                             (fsn_mlp_layer_maxima).  Both ends of the calibration are reported through `status` */
#define FSN_PREC_FP16X2 6 /* two fp16 MFMA passes: activations high+low parts, weights' high part only (the x3 blob
                             layout; inference only).  Measured accuracy in DESIGN.md - not the parity mode */

/* Range guard of the fp16 modes.  Entry points that evaluate the MLP take `status`: a DEVICE pointer to one
 * uint32_t owned by the caller (or NULL = no report).  A kernel sets bit 0 (atomic OR, never clears) when a hidden
 * activation - or, in the training backward, a scaled gradient - reached fp16 infinity, i.e. left the range in
 * which FSN_PREC_FP16X3 / _FP16 / _FP16X2 are valid; the results of that launch are then not to be used (re-run in
 * FSN_PREC_BF16X3, which has the range of float32).  Bit 1 is the other end of the envelope of the split modes
 * (FSN_PREC_FP16X3 / _FP16X2, forward passes): in some layer the largest |activation| over a wavefront's 16 samples x
 * all features was non-zero but below 2^-14 (an fp16 subnormal); high + scaled low part then no longer carry 22 bits
 * relative to the layer's scale - results are finite and still ~bf16x3-grade, the caller decides (the Python host
 * re-runs in FSN_PREC_BF16X3).  FSN_PREC_FP16X3U keeps unscaled low parts (an fp16 subnormal, fixed 2^-24 resolution,
 * for |activation| < 2^-3): bit 1 there means that a wavefront's layer maximum was below 2^-4, i.e. the layer's scale
 * (fsn_mlp_pack_scaled) no longer fits the data - the caller re-calibrates and re-packs.  The bf16 modes set neither bit. */
#define FSN_STATUS_FP16_RANGE 1u
#define FSN_STATUS_FP16_SMALL 2u
#define FSN_STATUS_GRAD_RANGE 4u /* fsn_nerf_train_bwd with stage_scales: a stored gradient reached fp16 infinity - the
                                   call's gradients are zeroed, fsn_adam_step skips the update, the stage's scale drops
                                   (an event of the delayed gradient scaling like a loss scaler's, not a range fallback) */

int fsn_version(void);
/* Debug build only (make -C fs-nerf_amd/csrc debug -> libfsnerf_hip_dbg.so, -DFSN_DEBUG): every index into the
 * LDS arrays of k_render_fused / k_render_occ is range-checked; a violation is recorded and clamped, not trapped (GPU
 * sanitizers are not available on this pool).  out_host: 8 uint32 = {violations, source line of the first, its index,
 * the array's extent} for render.hip and for render_occ.hip; the records are cleared.  The release library returns
 * FSN_E_UNSUPPORTED. */
int fsn_debug_report(uint32_t* out_host);
/* ... its negative control: a one-block launch with one deliberate out-of-range index and one out-of-range span into a
 * 32-element LDS array (the next fsn_debug_report must say 2 violations, extent 32, for render.hip). */
int fsn_debug_selftest(void);
const char* fsn_last_error(void);
/* number of compute units of the current device (grid sizing), or negative on error */
int fsn_device_cus(void);

/* ---- a1: get_rays(pose, hwf, device)                     src/utils/utilities.py:36-82
 * pose_host: 12 floats = rows 0..2 of the camera-to-world matrix, row-major [3][4].
 * Rows [row0, row0+nrows) of the H x W image are produced (row0=0,nrows=H for a frame; a
 * rank renders its own row block).  rays_o, rays_d: [nrows*W, 3].  focal / near are doubles
 * because the reference forms W*0.5, 2*near and 1/(W/(2 focal)) in Python double precision
 * before rounding to float32 (utilities.py:67, 104-114). */
int fsn_get_rays(const float* pose_host, int H, int W, double focal, int row0, int nrows,
                 float* rays_o, float* rays_d, fsn_stream_t stream);

/* ---- a2: to_ndc(rays_o, rays_d, hwf, near)               src/utils/utilities.py:84-120 */
int fsn_to_ndc(const float* rays_o, const float* rays_d, int64_t n, int H, int W, double focal,
               double near, float* ndc_o, float* ndc_d, fsn_stream_t stream);

/* ---- f4: the datasets' ray tables       src/nerfdata/datasets/blender.py:174-191, llff.py:59-90
 * get_rays of ALL poses (DEVICE [n_poses, 12]: rows 0..2 of each camera-to-world matrix) in one launch, straight into
 * rays_o / rays_d [n_poses*H*W, 3] (pose-major, then row-major pixels: torch.stack(...).reshape(-1, 6) of the reference),
 * optionally mapped to NDC (utilities.py:84-120 with `near`), and - when aabb != NULL - the region of interest the
 * reference derives for its estimator (llff.py:77-84): aabb[0..2] = min, aabb[3..5] = max over all rays of {o, o + d},
 * divided by 2^(4-1); reduced inside the same launch (aabb_keys: 6 uint32 of DEVICE scratch).  Bit for bit the per-pose
 * fsn_get_rays / fsn_to_ndc / torch min-max sequence. */
int fsn_build_rays(const float* poses, int64_t n_poses, int H, int W, double focal, int ndc, double near,
                   float* rays_o, float* rays_d, float* aabb, uint32_t* aabb_keys, fsn_stream_t stream);

/* ---- a4: PositionalEncoder.forward(x)                    src/core/models.py:43-50
 * x [n, d_in] -> out [n, d_in*(1+2*n_freqs)], block order x, sin f0, cos f0, sin f1, ...
 * freqs_host: n_freqs float32 values (models.py:31-34).  mask (device, [d_out]) may be NULL:
 * the build's frequency mask, multiplied onto the encoded features (mask==1 = reference). */
int fsn_posenc_fwd(const float* x, int64_t n, int d_in, int n_freqs, const float* freqs_host,
                   const float* mask, float* out, fsn_stream_t stream);

/* ---- a8: the `estimator.sampling` slot                   src/render/rendering.py:66-74
 * Fixed-count stratified sampler (build's definition): S+1 sorted interval
 * edges per ray, edges [R, S+1].  u_mode 0: u ignored, e_i = near + i*step;
 * 1: u [R], one shift per ray, e_i = near + (i+u_r)*step; 2: u [R, S+1], per-edge jitter. */
int fsn_stratified_edges(float near, float far, int S, int64_t R, const float* u, int u_mode,
                         float* edges, fsn_stream_t stream);
/* edges [R, S+1] -> packed (ray_indices int64 [R*S], t_starts [R*S], t_ends [R*S]) */
int fsn_edges_to_packed(const float* edges, int64_t R, int S, int64_t* ray_indices,
                        float* t_starts, float* t_ends, fsn_stream_t stream);
/* hierarchical "+n_imp" step: inverse-CDF samples of the piecewise-constant pdf given by the
 * coarse weights [R,S] over edges [R,S+1], merged with the coarse edges and sorted:
 * edges_out [R, S+1+n_imp].  u [R, n_imp] or NULL (deterministic linspace(0,1,n_imp)). */
int fsn_sample_pdf_merge(const float* edges, const float* weights, int64_t R, int S, int n_imp,
                         const float* u, float* edges_out, fsn_stream_t stream);

/* ---- a7: nerfacc.volrend.rendering arithmetic            call site src/render/rendering.py:89-96
 * Dense form: every ray has S samples; sigmas [R,S], rgbs [R,S,3], t_starts/t_ends [R,S].
 * colors [R,3], opacity [R], depth [R]; weights/alphas/trans [R,S] may be NULL.
 * bkgd_host: 3 floats (render_bkgd) or NULL (no background term). */
int fsn_composite_fwd(const float* sigmas, const float* rgbs, const float* t_starts,
                      const float* t_ends, int64_t R, int S, const float* bkgd_host,
                      float* colors, float* opacity, float* depth, float* weights,
                      float* alphas, float* trans, fsn_stream_t stream);
/* Packed (variable samples per ray) form with the reference's own contract: samples of a ray
 * are contiguous; ray_indices int64 [N] non-decreasing.  Rays with no sample get the
 * background colour, zero opacity and zero depth (reference fallback, rendering.py:97-103). */
int fsn_composite_packed_fwd(const float* sigmas, const float* rgbs, const float* t_starts,
                             const float* t_ends, const int64_t* ray_indices, int64_t N,
                             int64_t R, const float* bkgd_host, float* colors, float* opacity,
                             float* depth, float* weights, float* alphas, float* trans,
                             fsn_stream_t stream);

/* ---- a5: NeRF(d_pos,d_dir,n_layers,d_hidden,skip,...).forward(x, dirs)   src/core/models.py:53-143 */
typedef struct fsn_mlp_desc {
  int32_t n_layers;    /* hidden layers before the bottleneck (8)                 */
  int32_t d_hidden;    /* 256 or 128                                              */
  uint32_t skip_mask;  /* bit i set: x_in is concatenated after layer i (i < n_layers-1) */
  int32_t n_freqs_pos; /* <= 10 */
  int32_t n_freqs_dir; /* <= 4  */
  float freqs_pos[16]; /* models.py:31-34 values */
  float freqs_dir[16];
} fsn_mlp_desc;

/* Size in bytes of the packed-weights blob for (desc, prec); negative on error. */
int64_t fsn_mlp_blob_bytes(const fsn_mlp_desc* desc, int prec);
/* Pack a reference-format state_dict into the MFMA streaming layout (DESIGN.md "weight blob").
 * weights / biases: HOST arrays of n_layers+4 DEVICE pointers in the order
 *   layers.0 .. layers.{n_layers-1}, sigma, connection, branch, rgb
 * each weight row-major [out, in] exactly as in the state_dict (src/core/models.py:96-108). */
int fsn_mlp_pack(const fsn_mlp_desc* desc, int prec, const float* const* weights_host_of_dev,
                 const float* const* biases_host_of_dev, void* blob, fsn_stream_t stream);
/* fsn_mlp_pack of the SCALED network (round 4; any prec, meant for FSN_PREC_FP16X3U).  layer_exps_host: n_layers + 2
 * int32 exponents e_g, one per GEMM in kernel order (hidden layers 0 .. n_layers-1, connection, branch), or NULL = all
 * zero = fsn_mlp_pack.  With s_g = 2^e_g the blob holds the network whose GEMM g produces s_g times the reference's
 * activations: rows of W_g that multiply activations are scaled by s_g / s_(g-1) (the connection and the sigma head read
 * layer n_layers-1, the branch reads the connection), rows that multiply an encoding by s_g, biases by s_g, sigma.weight
 * by 1 / s_(n_layers-1), rgb.weight by 1 / s_branch.  Every factor is a power of two and ReLU commutes with it: in exact
 * arithmetic the outputs are the reference's (src/core/models.py:111-143); the point is that the 16-bit parts the
 * kernels form sit where fp16 is accurate whatever the network's own scale.  |e_g| <= 60. */
int fsn_mlp_pack_scaled(const fsn_mlp_desc* desc, int prec, const float* const* weights_host_of_dev,
                        const float* const* biases_host_of_dev, const int32_t* layer_exps_host, void* blob,
                        fsn_stream_t stream);
/* Calibration of those scales: NeRF.forward(x, dirs) on n probe samples (any blob / prec; FSN_PREC_BF16X3 on an
 * unscaled blob has float32's range) reporting, per GEMM in kernel order, the largest |output after its activation|
 * (ReLU layers, branch: max relu; connection: max |feat|) seen by any sample.  maxima: DEVICE float32 [n_layers + 2],
 * zeroed by the caller (the launch takes the maximum with what is there, so several probes may accumulate).  Values are
 * those of the blob's own scale (a blob packed with exponents e reports 2^e_g x the reference's).  No other output. */
int fsn_mlp_layer_maxima(const fsn_mlp_desc* desc, int prec, const void* blob, const float* x, const float* dirs,
                         const float* pos_mask, const float* dir_mask, int64_t n, float* maxima, fsn_stream_t stream);
/* Same packing on the CPU with HOST pointers and a HOST blob (format tests, no GPU needed). */
int fsn_mlp_pack_host(const fsn_mlp_desc* desc, int prec, const float* const* weights_host,
                      const float* const* biases_host, void* blob_host);
int fsn_mlp_pack_scaled_host(const fsn_mlp_desc* desc, int prec, const float* const* weights_host,
                             const float* const* biases_host, const int32_t* layer_exps_host, void* blob_host);
/* x [n,3]; dirs [n,3] or NULL.  out [n,4]=[r,g,b,sigma] when dirs != NULL, else [n,1]=sigma.
 * pos_mask [3*(1+2*n_freqs_pos)] / dir_mask [3*(1+2*n_freqs_dir)] device pointers or NULL. */
int fsn_mlp_fwd(const fsn_mlp_desc* desc, int prec, const void* blob, const float* x,
                const float* dirs, const float* pos_mask, const float* dir_mask, int64_t n,
                float* out, uint32_t* status, fsn_stream_t stream);
/* fsn_mlp_fwd_rays: the same forward with the samples given as the reference's closures form them (sigma_fn /
 * rgb_sigma_fn, src/render/rendering.py:58-64, 76-84): sample s is the midpoint of [t_starts[s], t_ends[s]) on ray
 * ray_indices[s], x = o + d (t0 + t1) / 2 evaluated in the launch in the reference's operation order, dirs = d
 * (full != 0: [n,4] output, else [n,1]) - no [n,3] position / direction tensors in HBM. */
int fsn_mlp_fwd_rays(const fsn_mlp_desc* desc, int prec, const void* blob, const float* rays_o, const float* rays_d,
                     const int64_t* ray_indices, const float* t_starts, const float* t_ends, int full,
                     const float* pos_mask, const float* dir_mask, int64_t n, float* out, uint32_t* status,
                     fsn_stream_t stream);

/* ---- a6: render_rays(rays_o, rays_d, estimator, model, ...)   src/render/rendering.py:25-107
 * The whole path fused in one launch for the fixed-count sampler: stratified edges ->
 * [density pass of the coarse net -> weights -> sample_pdf -> sorted union] -> full pass of
 * the fine net -> volume integration.  n_imp == 0 renders the S coarse intervals with
 * `blob_fine` only (blob_coarse unused).  S_out = S + n_imp intervals per ray.
 *   u_mode/u as fsn_stratified_edges; u_fine [R,n_imp] or NULL (deterministic).
 *   rays_o / rays_d [R,3], or rays_o == NULL and the cam_* fields (rays generated in the launch);
 *   outputs: colors [R,3], opacity [R], depth [R]; optional (NULL to skip):
 *   weights, alphas, trans, sigmas [R,S_out], rgbs [R,S_out,3], edges_out [R,S_out+1],
 *   weights_coarse [R,S] (only written when n_imp > 0). */
typedef struct fsn_render_args {
  const float* rays_o; /* [R,3] */
  const float* rays_d; /* [R,3] */
  int64_t R;
  float near, far;
  int32_t S, n_imp;
  int32_t u_mode;
  const float* u;
  const float* u_fine;
  const float* pos_mask;
  const float* dir_mask;
  float bkgd[3];
  float* colors;
  float* opacity;
  float* depth;
  float* weights;
  float* alphas;
  float* trans;
  float* sigmas;
  float* rgbs;
  float* edges_out;
  float* weights_coarse;
  uint32_t* status; /* range guard of the fp16 modes (above), or NULL */
  /* f3: rays generated inside the launch.  When rays_o == NULL (rays_d ignored), ray r is pixel
   * (cam_row0 + r / cam_W, r % cam_W) of the cam_H x cam_W pinhole image of `cam_pose` (12 floats, rows 0..2 of the
   * camera-to-world matrix) with focal length cam_focal: the arithmetic of fsn_get_rays (utilities.py:57-80),
   * so a frame is ONE launch with no ray tensors in HBM.  R = number of pixels rendered. */
  float cam_pose[12];
  int32_t cam_H, cam_W, cam_row0;
  double cam_focal;
  /* Hierarchical launches over many rays run in two phases when `edges_out` is given (it doubles as the hand-over
   * buffer): every workgroup first runs the coarse pass + resampling of ALL its ray groups, then the fine pass of
   * all of them, so that an XCD's L2 holds ONE network's weight stream at a time (2.0 / 2.3 MB of the 4 MiB)
   * instead of both.  Results are identical.  two_phase: 0 = never, 1 = whenever edges_out != NULL and n_imp > 0,
   * 2 = SAMPLER ONLY: the launch stops after the coarse pass + resampling; edges_out [R,S+n_imp+1] (and
   * weights_coarse when given) are its results, the per-ray outputs are not written and may be NULL, blob_fine may be
   * NULL.  This is estimator.sampling of the hierarchical sampler (stratified edges -> density pass -> weights ->
   * inverse-CDF resampling -> sorted union) as ONE launch in front of the training forward. */
  int32_t two_phase;
  /* Measurement aid (bench.py, NULL = off): DEVICE uint64_t[2].  Wave 0 of workgroup 0 reads the shader clock counter
   * (s_memtime) and the constant 100 MHz counter (s_memrealtime) when it enters and when it leaves the kernel and ADDS
   * the two differences to clock_out[0] / clock_out[1]: their ratio x 0.1 is the clock in GHz the chip held during the
   * launch(es) - workgroup 0 lives as long as the persistent launch does. */
  uint64_t* clock_out;
} fsn_render_args;

int fsn_render_rays_fused(const fsn_mlp_desc* desc, int prec, const void* blob_coarse,
                          const void* blob_fine, const fsn_render_args* args_host,
                          fsn_stream_t stream);

/* Measurement aid (bench.py's `roofline.bare_stream`; not part of the path): the fused kernel's own MFMA stream with
 * nothing else in the kernel.  One persistent 512-thread workgroup per compute unit runs `layers` 256 -> 256 hidden
 * layers back to back through the SAME code as the render kernels (mlp_dev.hpp gemm_layer: the hand-scheduled GEMM-pair
 * blocks, the pair epilogues with ReLU / 16-bit split / range tracking, the weight stream's LDS-DMA ring with its
 * barriers) on fixed pseudo-random activations, the weight stream WALKING the hidden phases of a real packed blob
 * (d_hidden 256, x3 / x2 modes) like a density pass does - no encodings, heads, samplers, compositing or tile tails.
 * clock_out: DEVICE uint64_t[2 * 8 * n_workgroups] or NULL: per wave the s_memtime and s_memrealtime differences
 * around the loop.  Returns the number of workgroups launched (> 0) or an FSN_E_* code; per workgroup and layer the
 * stream issues 8 waves x 384 MFMAs (v_mfma_f32_16x16x32, 16,384 FLOP each) and fills 256 KiB from L2. */
int fsn_bench_bare_stream(const fsn_mlp_desc* desc, int prec, const void* blob, int layers, uint64_t* clock_out,
                          fsn_stream_t stream);

/* ---- a6 with the occupancy estimator in the `estimator` slot: the path the reference itself renders with
 * (render_rays, src/render/rendering.py:58-107; OccGridEstimator, src/run-nerf.py:96-98, 288-295) in ONE launch:
 * grid march -> density pass (sigma_fn) -> visibility cull -> full pass (rgb_sigma_fn) -> packed volume integration,
 * per batch of rays inside persistent workgroups, no host sync for the data-dependent sample count.  Same sampling
 * rule and arithmetic as fsn_occgrid_march + fsn_mlp_fwd + fsn_packed_visibility + fsn_mlp_fwd +
 * fsn_composite_packed_fwd (the results are the unfused sequence's).  Outputs per ray only: colors [R,3], opacity [R],
 * depth [R] (what render_frame consumes, rendering.py:169-171), optional sample counts per ray n_cand / n_kept [R]
 * (int32; NULL to skip); the per-sample extras of render_rays' full return contract in EXTRAS mode (below).
 *   aabb / res / levels / bits / near_plane / far_plane / step / u (one jitter value per ray or NULL) / max_steps as
 *   fsn_occgrid_march (max_steps <= 2048 here, else FSN_E_UNSUPPORTED); early_stop_eps / alpha_thre as
 *   fsn_packed_visibility (both <= 0: no density pass, every marched sample is kept);
 *   rays_o / rays_d [R,3], or rays_o == NULL and the cam_* fields as in fsn_render_args;
 *   work_counter: 8 bytes of DEVICE scratch (the launch zeroes it: the global ray-chunk queue);
 *   prec: FSN_PREC_BF16X3 .. FSN_PREC_FP16. */
typedef struct fsn_occ_render_args {
  const float* rays_o;
  const float* rays_d;
  int64_t R;
  float aabb[6];
  int32_t res, levels;
  const uint32_t* bits;
  float near_plane, far_plane, step;
  const float* u;
  int32_t max_steps;
  float early_stop_eps, alpha_thre;
  const float* pos_mask;
  const float* dir_mask;
  float bkgd[3];
  float* colors;
  float* opacity;
  float* depth;
  int32_t* n_cand;
  int32_t* n_kept;
  uint32_t* status;
  void* work_counter;
  float cam_pose[12];
  int32_t cam_H, cam_W, cam_row0;
  double cam_focal;
  /* SAMPLER mode (sample_t0 != NULL; estimator.sampling(..., sigma_fn) of a training step, rendering.py:58-74): every
   * batch stops after the visibility cull; per ray the kept count goes to n_kept (required) and the kept samples'
   * interval starts to sample_t0[ray * sample_cap + i] (sample_cap >= max_steps); colors / opacity / depth are not
   * written and may be NULL.  fsn_occ_gather_samples packs the slots behind an exclusive scan of n_kept. */
  float* sample_t0;
  int32_t sample_cap;
  /* EXTRAS mode (round 4; ex_weights != NULL, with sample_t0 / sample_cap / n_kept as above): render_rays' FULL return
   * contract (rendering.py:88-107: (rgb, opacity, depth, extras), ray_indices, t_vals) in the one launch, without
   * gradients.  The launch runs the full pass and the integration as usual (colors / opacity / depth are written) and
   * additionally leaves, in per-ray slot rows of sample_cap entries, the kept samples' weights / alphas / trans / sigmas
   * [R, sample_cap] and rgbs [R, sample_cap, 3] next to their interval starts in sample_t0; fsn_occ_gather_extras packs
   * them behind an exclusive scan of n_kept into nerfacc's packed [N] arrays. */
  float* ex_weights;
  float* ex_alphas;
  float* ex_trans;
  float* ex_sigmas;
  float* ex_rgbs;
} fsn_occ_render_args;

int fsn_render_rays_occgrid(const fsn_mlp_desc* desc, int prec, const void* blob, const fsn_occ_render_args* args_host,
                            fsn_stream_t stream);
/* Packed (ray_indices, t_starts, t_ends = t_start + step) from the sampler mode's per-ray slots: offsets = exclusive
 * scan of n_kept (int64 [R]); the outputs hold sum(n_kept) entries, sorted by ray then t - what fsn_occgrid_march +
 * fsn_packed_visibility + a compaction produce, bit for bit. */
int fsn_occ_gather_samples(const int32_t* n_kept, const int64_t* offsets, const float* sample_t0, int sample_cap,
                           int64_t R, float step, int64_t* ray_indices, float* t_starts, float* t_ends,
                           fsn_stream_t stream);
/* fsn_occ_gather_samples plus the EXTRAS mode's slot rows -> packed weights / alphas / trans / sigmas [N], rgbs [N,3]
 * (slots / outputs in that order: HOST arrays of 5 DEVICE pointers). */
int fsn_occ_gather_extras(const int32_t* n_kept, const int64_t* offsets, const float* sample_t0, int sample_cap, int64_t R,
                          float step, int64_t* ray_indices, float* t_starts, float* t_ends,
                          const float* const* slots_host_of_dev, float* const* out_host_of_dev, fsn_stream_t stream);

/* ---- "next" rows ------------------------------------------------------------------ */

/* f4: OcclusionRegularizer.__call__(sigmas, t_vals, ray_idxs)          src/core/loss.py:26-60
 * mean over the rays that own at least one sample of  sum_i w(t_i) sigma_i,  w = -a t + b (func 0,
 * 'linear') or a exp(-b t) (func 1, 'exp').  ray_idxs int64 [N] non-decreasing, n_rays > max index.
 * ray_sums [n_rays] is caller-provided workspace; out is one float. */
int fsn_occlusion_reg_fwd(const float* sigmas, const float* t_vals, const int64_t* ray_idxs, int64_t N,
                          int64_t n_rays, float a, float b, int func, float* ray_sums, float* out,
                          fsn_stream_t stream);

/* backward of fsn_occlusion_reg_fwd w.r.t. sigmas: d_sigmas[i] = *d_out * w(t_i) / #rays with samples.  ray_sums is
 * the forward's workspace (NaN marks rays without samples); count_ws: one int32 of device scratch. */
int fsn_occlusion_reg_bwd(const float* t_vals, int64_t N, const float* ray_sums, int64_t n_rays, float a, float b,
                          int func, const float* d_out, int32_t* count_ws, float* d_sigmas, fsn_stream_t stream);

/* f1: the training step around the path.                         src/run-nerf.py:243-285, models.py:111-143
 * NeRF.forward keeping what its backward needs in a caller-provided workspace, and that backward (gradients of
 * every parameter; sample positions / directions receive none on this path).
 *   prec 0..3 (FSN_PREC_*).  Forward = the inference kernel with fp32 activations saved in tiles of
 *     128 samples; backward = register-resident dgrad chain on transposed weights + split-K wgrad GEMMs over all
 *     samples (csrc/train_fused.hip).  grad_scale: DEVICE pointer to one float, a power of two that d_out is
 *     multiplied by on entry (results are divided by it again) so that fp16 parts keep small gradients; null = 1.
 *   weights / biases / d_weights / d_biases: HOST arrays of n_layers+4 DEVICE pointers in state_dict order
 *   (layers.0.., sigma, connection, branch, rgb); gradients are overwritten, not accumulated.
 *   workspace: fsn_nerf_train_workspace_floats(desc, prec, n) floats, written by _fwd, consumed (and scribbled on)
 *   by _bwd with the same desc / prec / n; out / d_out [n,4] = [rgb, sigma].
 *   status: the range-guard word OF THIS STEP AND THIS NETWORK (device, or NULL): the caller zeroes it before _fwd
 *   and hands the same word to _bwd.  In the fp16 modes _bwd writes this call's gradients as ZEROS when bit 0
 *   (FSN_STATUS_FP16_RANGE) was raised by either launch (the sums are inf / NaN then); the bf16 modes never consult
 *   it.  It is NOT a sticky shared word: a flag raised by another network or by an inference launch must not zero
 *   this network's gradients.  Hand it to fsn_adam_step(skip_word) to skip the update of a flagged step. */
int64_t fsn_nerf_train_workspace_floats(const fsn_mlp_desc* desc, int prec, int64_t n);
int fsn_nerf_train_fwd(const fsn_mlp_desc* desc, int prec, const float* const* weights, const float* const* biases,
                       const float* x, const float* dirs, const float* pos_mask, const float* dir_mask,
                       int64_t n, float* workspace, float* out, uint32_t* status, fsn_stream_t stream);
/* fsn_nerf_train_fwd with the samples in ray form (as fsn_mlp_fwd_rays): the training step's forward reads the rays and
 * the packed intervals, never a gathered [n,3] tensor (run-nerf.py:243-252 -> rendering.py:76-84). */
int fsn_nerf_train_fwd_rays(const fsn_mlp_desc* desc, int prec, const float* const* weights, const float* const* biases,
                            const float* rays_o, const float* rays_d, const int64_t* ray_indices, const float* t_starts,
                            const float* t_ends, const float* pos_mask, const float* dir_mask, int64_t n,
                            float* workspace, float* out, uint32_t* status, fsn_stream_t stream);
/* accumulate != 0: the gradients are ADDED to d_weights / d_biases (the caller's .grad buffers: what autograd's
 * AccumulateGrad would do with a returned tensor, without the temporaries and the 24 add launches per step);
 * 0: they are overwritten.  A flagged call (bit 0 of *status) contributes zeros either way.
 * stage_scales / stage_amax (both or neither; fp16 modes): n_layers + 2 device floats, initially 1.0, and as many zeroed
 * device words - the backward chain's per-stage power-of-two factors on top of grad_scale and the maxima they are
 * steered by.  The call uses the factors it finds and leaves the ones for the NEXT call (delayed scaling): a network
 * whose layer gradients span more than fp16's range (weight-norm regularised ones do) keeps float32-grade weight
 * gradients from the second call on; run one throw-away call to calibrate a fresh pair of arrays. */
int fsn_nerf_train_bwd(const fsn_mlp_desc* desc, int prec, const float* const* weights, int64_t n, float* workspace,
                       const float* out, const float* d_out, const float* grad_scale, float* const* d_weights,
                       float* const* d_biases, int accumulate, float* stage_scales, uint32_t* stage_amax,
                       uint32_t* status, fsn_stream_t stream);
/* The power-of-two `grad_scale` of the fp16 modes' backward in one launch: 2^floor(log2(1024 / max|d_out|)), exponent
 * clamped to [-40, 60], 1 when the maximum is 0 / inf / NaN (the arithmetic of ops.grad_scale_for, which took nine
 * elementwise / reduction launches).  buf: 4 device floats, ZEROED by the caller; buf[0] receives the scale (buf[1..2]
 * are the launch's reduction words). */
int fsn_grad_scale(const float* d_out, int64_t n, float* buf, fsn_stream_t stream);
/* backward of fsn_composite_packed_fwd with respect to sigmas and rgbs, given dL/dcolors [R,3] and
 * (optional) dL/dopacity [R]; dL/ddepth is not propagated. */
int fsn_composite_packed_bwd(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                             const int64_t* ray_indices, int64_t N, int64_t R, const float* bkgd_host,
                             const float* d_colors, const float* d_opacity, float* d_sigmas, float* d_rgbs,
                             fsn_stream_t stream);

/* f1: optimizer side of the training step on ONE flat float32 parameter arena (run-nerf.py:217, 266-285).
 * fsn_adam_step: torch.optim.Adam's update (no amsgrad), operation for operation in float32, one launch over the
 *   arena: params / grads / exp_avg / exp_avg_sq [n]; `step` = 1, 2, ... (bias corrections are formed in double on
 *   the host); grad_div divides the gradient first (pass the number of ranks when `grads` holds an all-reduced SUM).
 *   skip_word / skip_count (DEVICE pointers or NULL): the launch leaves parameters and moments untouched when
 *   FSN_STATUS_FP16_RANGE or FSN_STATUS_GRAD_RANGE is set in *skip_word or *skip_count > 0 - the per-step status word of fsn_nerf_train_fwd/_bwd (fp16 overflow in
 *   this step) and the flag slot of an all-reduced gradient bucket; decided on the device, no host sync.
 * fsn_weight_norm_*: the weight-norm "frequency" regulariser of run-nerf.py:266-279: out = sum over the selected
 *   tensors (segments [off, off+len) of the arena, HOST tables, <= 40) of |w|_1 (l2 = 0) or |w|_2 (l2 = 1).
 *   workspace: fsn_weight_norm_workspace_floats(...) floats, written by _fwd and read by _bwd (per-tensor norms);
 *   _bwd ACCUMULATES d_out[0] * d(out)/dw into grad_arena (same layout as the arena).  No atomics: sums are taken
 *   in a fixed order. */
int fsn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                  double lr, double beta1, double beta2, double eps, double weight_decay, double grad_div,
                  const uint32_t* skip_word, const float* skip_count, fsn_stream_t stream);
/* fsn_adam_step with the step counter ON THE DEVICE (round 4): step_count (DEVICE int32[1]) holds the number of updates
 * applied so far; a first one-thread launch looks at skip_word / skip_count, and only when the step runs advances the
 * counter and forms the bias-correction constants (double precision, torch's expressions) into tick (DEVICE float[4],
 * scratch); the update launch reads them.  A skipped step therefore does not advance the bias corrections - as with
 * torch's GradScaler, which does not call optimizer.step() on an overflowing step - and the host never needs the count. */
int fsn_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t* step_count,
                      float* tick, double lr, double beta1, double beta2, double eps, double weight_decay, double grad_div,
                      const uint32_t* skip_word, const float* skip_count, fsn_stream_t stream);
int64_t fsn_weight_norm_workspace_floats(int n_seg, const int64_t* seg_len_host);
int fsn_weight_norm_fwd(const float* arena, int n_seg, const int64_t* seg_off_host, const int64_t* seg_len_host, int l2,
                        float* workspace, float* out, fsn_stream_t stream);
int fsn_weight_norm_bwd(const float* arena, int n_seg, const int64_t* seg_off_host, const int64_t* seg_len_host, int l2,
                        const float* workspace, const float* d_out, float* grad_arena, fsn_stream_t stream);

/* f2: occupancy-grid sampler for the `estimator` slot (nerfacc OccGridEstimator; call sites
 * src/render/rendering.py:66-74, src/run-nerf.py:96-98, 288-295).  nerfacc is not part of the reference: the
 * arithmetic is this build's definition of the contract, mirrored by oracle/fsnerf_oracle.py.
 *   grid: `levels` nested boxes (level l = roi aabb scaled by 2^l about its centre), res^3 cells each, cell index
 *   (ix*res + iy)*res + iz (+ l*res^3); `bits` = one bit per cell (word c>>5, bit c&31), `occs` fp32 per cell.
 * fsn_occgrid_march: per ray the lattice t_k = near_r + k*step, near_r = near_plane + (u ? u[r]*step : 0); interval
 *   [t_k, t_k+step) is a sample iff t_k lies in [max(t_enter, near_r), min(t_exit, far_plane)) of the outermost box
 *   and the cell of its midpoint (finest level containing it) is occupied; at most max_steps lattice points per ray.
 *   Two passes: offsets == NULL -> counts[R] (int64); else fill ray_indices / t_starts / t_ends at offsets[r]
 *   (exclusive scan of the counts, by the caller).
 * fsn_packed_visibility: keep[i] = (T_i >= early_stop_eps && alpha_i >= alpha_thre), T = exp(-exclusive sum of
 *   sigma*dt) per ray, samples packed and sorted by ray.
 * fsn_occgrid_update: occs[cells[i]] = max(occs[cells[i]]*decay, vals[i]) (cells must be unique), then, when
 *   threshold_dev != NULL (device pointer to one float), bits = (occs > *threshold_dev). n_cells % 64 == 0. */
int fsn_occgrid_march(const float* rays_o, const float* rays_d, int64_t R, const float* aabb_host, int res, int levels,
                      const uint32_t* bits, float near_plane, float far_plane, float step, const float* u,
                      int max_steps, int64_t* counts, const int64_t* offsets, int64_t* ray_indices, float* t_starts,
                      float* t_ends, fsn_stream_t stream);
int fsn_packed_visibility(const float* sigmas, const float* t_starts, const float* t_ends, const int64_t* ray_indices,
                          int64_t N, int64_t R, float early_stop_eps, float alpha_thre, uint8_t* keep,
                          fsn_stream_t stream);
int fsn_occgrid_update(float* occs, int64_t n_cells, const int64_t* cells, const float* vals, int64_t n, float decay,
                       const float* threshold_dev, uint32_t* bits, fsn_stream_t stream);
/* update_every_n_steps on the device (round 4; run-nerf.py:288-295): which cells of level `lvl` an update re-evaluates,
 * and where.  all_cells != 0 (warm-up): draw i is cell i, n = res^3.  Else n = n_uniform + n_occupied draws WITH
 * replacement: the first n_uniform uniform over the level's cells, the rest uniform over its OCCUPIED cells read straight
 * from the bit field (popcount prefix over its words in prefix_scratch, DEVICE int32 [res^3/32 + 1]; uniform when the
 * level is empty).  Each draw gets a point uniform inside its cell of the level's box.  Randomness: the counter-based
 * hash of csrc/occgrid.hip (occ_rand) of (seed, draw index) - no generator state, no host sync, restated by the oracle.
 * cells: int64 [n] (global cell index, + lvl res^3), x: [n,3].
 * fsn_occgrid_update_multi: fsn_occgrid_update's EMA for draws that may repeat a cell: occs[c] = max(occs[c] * decay,
 * max of the vals drawn for c) - the maximum is taken first (pending: DEVICE uint32 [n_cells], all zero between calls),
 * so the result does not depend on the order of the draws. */
int fsn_occgrid_select(const uint32_t* bits, int res, int levels, int lvl, const float* aabb_host, int all_cells,
                       int64_t n_uniform, int64_t n_occupied, uint64_t seed, int32_t* prefix_scratch, int64_t* cells,
                       float* x, fsn_stream_t stream);
int fsn_occgrid_update_multi(float* occs, int64_t n_cells, uint32_t* pending, const int64_t* cells, const float* vals,
                             int64_t n, float decay, fsn_stream_t stream);

/* f3: to8b(x) = (255 * clip(x, 0, 1)).astype(uint8)                    src/render/rendering.py:21 */
int fsn_to8b(const float* x, int64_t n, uint8_t* out, fsn_stream_t stream);
/* f3: render_video(frames, d_frames, cmap)                             src/render/rendering.py:240-266
 * fsn_to8b_nchw: frames [N,HW,3] float -> uint8 [N,3,HW] = transpose(to8b(frames), (0,3,1,2)).
 * fsn_depth_colormap: depth [N,HW] -> uint8 [N,3,HW]: matplotlib's Normalize(vmin, vmax) (float32; vmin == vmax
 *   maps to 0) and ScalarMappable lookup, index = int(x * 256) (x == 1 -> 255, out of range -> first / last entry),
 *   in lut_rgb8 = to8b of the colormap's 256 RGB entries (uint8 [256][3], device).  vmin_vmax: 2 floats (device). */
int fsn_to8b_nchw(const float* frames, int64_t n_frames, int64_t hw, uint8_t* out, fsn_stream_t stream);
int fsn_depth_colormap(const float* depth, int64_t n_frames, int64_t hw, const float* vmin_vmax,
                       const uint8_t* lut_rgb8, uint8_t* out, fsn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FSNERF_HIP_H */

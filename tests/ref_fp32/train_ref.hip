// train_ref.hip — TEST-ONLY reference formulation of the NeRF training step (not part of libfsnerf_hip.so):
// fp32 activations round-trip HBM layer by layer and every Linear layer is a plain library GEMM (rocBLAS sgemm,
// bound with dlopen at first use); elementwise pieces are the small HIP kernels below.  tests/test_train_step.py
// checks the hand-written MFMA training kernels of the product against it (next to float64 autograd on the oracle).
// Built into tests/ref_fp32/libfsnerf_ref_fp32.so by tests/ref_fp32/Makefile; loaded only by the tests
// (fs_nerf_amd._lib.register_reference_library).
//
// reference: src/core/models.py:111-143 (forward), src/run-nerf.py:243-285 (loss.backward()).
#include "common.hpp"
#include "mlp_layout.hpp"

#include <dlfcn.h>

#include <mutex>

namespace fsn {


// ------------------------------------------------------------------ rocBLAS (dynamic)
typedef void* rb_handle;
typedef int (*rb_create_t)(rb_handle*);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_atomics_t)(rb_handle, int);
typedef int (*rb_sgemm_t)(rb_handle, int, int, int, int, int, const float*, const float*, int, const float*, int,
                          const float*, float*, int);
constexpr int RB_N = 111, RB_T = 112;  // rocblas_operation_none / _transpose

struct RocBlas {
  rb_handle h = nullptr;
  rb_set_stream_t set_stream = nullptr;
  rb_sgemm_t sgemm = nullptr;
};
static RocBlas g_rb;
static std::mutex g_rb_mutex;

static int rb_get(hipStream_t s, RocBlas** out) {
  std::lock_guard<std::mutex> lk(g_rb_mutex);
  if (!g_rb.h) {
    void* lib = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    FSN_REQUIRE(lib, FSN_E_HIP, "training path: cannot load rocBLAS (%s)", dlerror());
    rb_create_t create = (rb_create_t)dlsym(lib, "rocblas_create_handle");
    g_rb.set_stream = (rb_set_stream_t)dlsym(lib, "rocblas_set_stream");
    g_rb.sgemm = (rb_sgemm_t)dlsym(lib, "rocblas_sgemm");
    rb_atomics_t atomics = (rb_atomics_t)dlsym(lib, "rocblas_set_atomics_mode");
    FSN_REQUIRE(create && g_rb.set_stream && g_rb.sgemm, FSN_E_HIP, "training path: rocBLAS symbols missing");
    FSN_REQUIRE(create(&g_rb.h) == 0 && g_rb.h, FSN_E_HIP, "rocblas_create_handle failed");
    if (atomics) atomics(g_rb.h, 0);  // rocblas_atomics_not_allowed: reproducible sums
  }
  FSN_REQUIRE(g_rb.set_stream(g_rb.h, s) == 0, FSN_E_HIP, "rocblas_set_stream failed");
  *out = &g_rb;
  return FSN_OK;
}

// Row-major helpers (A[N,K] lda, W[M,K] ldw, Y[N,M] ldy ...), all fp32, beta = 0.
static int gemm_xwT(RocBlas* rb, int64_t N, int M, int K, const float* X, int ldx, const float* W, int ldw, float* Y,
                    int ldy) {  // Y = X . W^T
  const float one = 1.f, zero = 0.f;
  FSN_REQUIRE(rb->sgemm(rb->h, RB_T, RB_N, M, (int)N, K, &one, W, ldw, X, ldx, &zero, Y, ldy) == 0, FSN_E_HIP,
              "rocblas_sgemm (forward) failed");
  return FSN_OK;
}
static int gemm_dyw(RocBlas* rb, int64_t N, int M, int K, const float* dY, int ldy, const float* W, int ldw, float* dX,
                    int ldx) {  // dX[N,K] = dY[N,M] . W[M,K]
  const float one = 1.f, zero = 0.f;
  FSN_REQUIRE(rb->sgemm(rb->h, RB_N, RB_N, K, (int)N, M, &one, W, ldw, dY, ldy, &zero, dX, ldx) == 0, FSN_E_HIP,
              "rocblas_sgemm (dgrad) failed");
  return FSN_OK;
}
static int gemm_dyTx(RocBlas* rb, int64_t N, int M, int K, const float* dY, int ldy, const float* X, int ldx, float* dW,
                     int ldw) {  // dW[M,K] = dY[N,M]^T . X[N,K]
  const float one = 1.f, zero = 0.f;
  FSN_REQUIRE(rb->sgemm(rb->h, RB_N, RB_T, K, M, (int)N, &one, X, ldx, dY, ldy, &zero, dW, ldw) == 0, FSN_E_HIP,
              "rocblas_sgemm (wgrad) failed");
  return FSN_OK;
}

// ------------------------------------------------------------------ elementwise kernels
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

__global__ void k_bias_act(float* __restrict__ Y, int ld, int64_t N, int M, const float* __restrict__ b, int act) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * M) return;
  const int64_t n = e / M;
  const int m = (int)(e - n * M);
  float v = Y[n * ld + m] + b[m];
  if (act == ACT_RELU) v = fmaxf(v, 0.f);
  if (act == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
  Y[n * ld + m] = v;
}

__global__ void k_copy_cols(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int64_t N, int C) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * C) return;
  const int64_t n = e / C;
  const int c = (int)(e - n * C);
  dst[n * ldd + c] = src[n * lds + c];
}

// dY *= (H > 0)
__global__ void k_relu_bwd(float* __restrict__ dY, int ldy, const float* __restrict__ H, int ldh, int64_t N, int M) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * M) return;
  const int64_t n = e / M;
  const int m = (int)(e - n * M);
  if (!(H[n * ldh + m] > 0.f)) dY[n * ldy + m] = 0.f;
}

// dZ[n,c] = d_out[n,c] * rgb (1 - rgb), c < 3 (out = [rgb, sigma], ld 4); dZ dense [N,3]
__global__ void k_sigmoid_bwd(const float* __restrict__ d_out, const float* __restrict__ out, int64_t N,
                              float* __restrict__ dZ) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * 3) return;
  const int64_t n = e / 3;
  const int c = (int)(e - n * 3);
  const float r = out[n * 4 + c];
  dZ[e] = d_out[n * 4 + c] * r * (1.0f - r);
}

// dH[n,:] += d_sigma[n] * w_sigma[:]   (d_sigma = d_out[:,3])
__global__ void k_add_rank1(float* __restrict__ dH, int ld, int64_t N, int M, const float* __restrict__ d_out,
                            const float* __restrict__ w) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * M) return;
  const int64_t n = e / M;
  const int m = (int)(e - n * M);
  dH[n * ld + m] += d_out[n * 4 + 3] * w[m];
}

// Column sums in two deterministic stages: partial[p][m] over row block p, then a fixed-order reduce.
constexpr int kColParts = 128;
__global__ void k_colsum_partial(const float* __restrict__ Y, int ld, int64_t N, int M, float* __restrict__ partial) {
  const int m = blockIdx.x * 64 + (threadIdx.x & 63);
  const int p = blockIdx.y;
  const int64_t rows = (N + kColParts - 1) / kColParts;
  const int64_t r0 = p * rows, r1 = min(r0 + rows, N);
  float acc = 0.f;
  if (m < M)
    for (int64_t n = r0 + (threadIdx.x >> 6); n < r1; n += 4) acc += Y[n * ld + m];
  __shared__ float s[256];
  s[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && m < M) partial[(int64_t)p * M + m] = (s[threadIdx.x] + s[threadIdx.x + 64]) + (s[threadIdx.x + 128] + s[threadIdx.x + 192]);
}
__global__ void k_colsum_final(const float* __restrict__ partial, int M, float* __restrict__ out) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float acc = 0.f;
  for (int p = 0; p < kColParts; ++p) acc += partial[(int64_t)p * M + m];
  out[m] = acc;
}

// posenc into a strided destination (same arithmetic as k_posenc in ray_ops.hip, models.py:43-50)
struct Freqs16 { float f[16]; };
__global__ void k_posenc_ld(const float* __restrict__ x, int64_t n, int n_freqs, Freqs16 fr, const float* __restrict__ mask,
                            float* __restrict__ out, int ldo) {
  const int d_out = 3 * (1 + 2 * n_freqs);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * d_out) return;
  const int64_t p = e / d_out;
  const int f = (int)(e - p * d_out);
  float v;
  if (f < 3) {
    v = x[p * 3 + f];
  } else {
    const int q = f - 3, band = q / 6, r = q - band * 6, c = r < 3 ? r : r - 3;
    const float a = x[p * 3 + c] * fr.f[band];
    v = r < 3 ? sinf(a) : cosf(a);
  }
  if (mask) v = v * mask[f];
  out[p * ldo + f] = v;
}

// ------------------------------------------------------------------ workspace layout
struct TrainLayout {
  int L, D, d_pe, d_de;
  int stride[kMaxLayers];  // row stride of H_l (D, or D + d_pe when layer l+1 is wide)
  int64_t off_xin, off_h[kMaxLayers], off_f, off_bo, off_g0, off_g1, off_dz, off_part, total;
};

static void make_layout(const fsn_mlp_desc& d, int64_t n, TrainLayout& T) {
  T.L = d.n_layers; T.D = d.d_hidden;
  T.d_pe = 3 * (1 + 2 * d.n_freqs_pos); T.d_de = 3 * (1 + 2 * d.n_freqs_dir);
  int64_t o = 0;
  T.off_xin = o; o += n * T.d_pe;
  for (int l = 0; l < T.L; ++l) {
    T.stride[l] = T.D + (((d.skip_mask >> l) & 1u) ? T.d_pe : 0);
    T.off_h[l] = o; o += n * T.stride[l];
  }
  T.off_f = o; o += n * (T.D + T.d_de);
  T.off_bo = o; o += n * (T.D / 2);
  const int gmax = T.D + (T.d_pe > T.d_de ? T.d_pe : T.d_de);
  T.off_g0 = o; o += n * gmax;
  T.off_g1 = o; o += n * gmax;
  T.off_dz = o; o += n * 3;
  T.off_part = o; o += (int64_t)kColParts * gmax;
  T.total = o;
}

static inline unsigned nb(int64_t n) { return (unsigned)((n + 255) / 256); }

static int colsum(hipStream_t s, const float* Y, int ld, int64_t N, int M, float* partial, float* out) {
  dim3 grid((M + 63) / 64, kColParts);
  k_colsum_partial<<<grid, 256, 0, s>>>(Y, ld, N, M, partial);
  FSN_LAUNCH_CHECK("k_colsum_partial");
  k_colsum_final<<<(M + 255) / 256, 256, 0, s>>>(partial, M, out);
  FSN_LAUNCH_CHECK("k_colsum_final");
  return FSN_OK;
}

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

extern "C" int64_t fsnref_train_workspace_floats(const fsn_mlp_desc* desc, int64_t n) {
  if (check_desc(desc) != FSN_OK) return FSN_E_INVALID;
  FSN_REQUIRE(n >= 0, FSN_E_INVALID, "fsnref_train_workspace_floats: n < 0");
  TrainLayout T;
  make_layout(*desc, n, T);
  return T.total;
}

extern "C" int fsnref_train_fwd(const fsn_mlp_desc* desc, const float* const* W, const float* const* b,
                                  const float* x, const float* dirs, const float* pos_mask, const float* dir_mask,
                                  int64_t n, float* ws, float* out, fsn_stream_t stream) {
  int rc = check_desc(desc);
  if (rc != FSN_OK) return rc;
  if (n == 0) return FSN_OK;
  FSN_REQUIRE(W && b && x && dirs && ws && out, FSN_E_INVALID, "fsnref_train_fwd: null pointer");
  FSN_REQUIRE(n < (1ll << 31), FSN_E_UNSUPPORTED, "fsnref_train_fwd: n too large for one call");
  hipStream_t s = as_stream(stream);
  RocBlas* rb;
  rc = rb_get(s, &rb);
  if (rc != FSN_OK) return rc;
  TrainLayout T;
  make_layout(*desc, n, T);
  const int L = T.L, D = T.D;
  Freqs16 fp{}, fd{};
  for (int i = 0; i < 16; ++i) { fp.f[i] = desc->freqs_pos[i]; fd.f[i] = desc->freqs_dir[i]; }
  float* XIN = ws + T.off_xin;
  k_posenc_ld<<<nb(n * T.d_pe), 256, 0, s>>>(x, n, desc->n_freqs_pos, fp, pos_mask, XIN, T.d_pe);
  FSN_LAUNCH_CHECK("k_posenc_ld");
  const float* in = XIN;
  int in_k = T.d_pe, in_ld = T.d_pe;
  for (int l = 0; l < L; ++l) {  // models.py:120-123
    float* H = ws + T.off_h[l];
    rc = gemm_xwT(rb, n, D, in_k, in, in_ld, W[l], in_k, H, T.stride[l]);
    if (rc != FSN_OK) return rc;
    k_bias_act<<<nb(n * D), 256, 0, s>>>(H, T.stride[l], n, D, b[l], ACT_RELU);
    FSN_LAUNCH_CHECK("k_bias_act");
    if (T.stride[l] > D) {
      k_copy_cols<<<nb(n * T.d_pe), 256, 0, s>>>(XIN, T.d_pe, H + D, T.stride[l], n, T.d_pe);
      FSN_LAUNCH_CHECK("k_copy_cols");
    }
    in = H; in_k = T.stride[l]; in_ld = T.stride[l];
  }
  const float* HL = ws + T.off_h[L - 1];
  const int ldh = T.stride[L - 1];
  // sigma -> out[:,3] (models.py:127), no activation
  rc = gemm_xwT(rb, n, 1, D, HL, ldh, W[L], D, out + 3, 4);
  if (rc != FSN_OK) return rc;
  k_bias_act<<<nb(n), 256, 0, s>>>(out + 3, 4, n, 1, b[L], ACT_NONE);
  FSN_LAUNCH_CHECK("k_bias_act");
  // connection (no activation) into F[:, :D]; F[:, D:] = dir encoding (models.py:130-132)
  float* F = ws + T.off_f;
  const int ldf = D + T.d_de;
  rc = gemm_xwT(rb, n, D, D, HL, ldh, W[L + 1], D, F, ldf);
  if (rc != FSN_OK) return rc;
  k_bias_act<<<nb(n * D), 256, 0, s>>>(F, ldf, n, D, b[L + 1], ACT_NONE);
  FSN_LAUNCH_CHECK("k_bias_act");
  k_posenc_ld<<<nb(n * T.d_de), 256, 0, s>>>(dirs, n, desc->n_freqs_dir, fd, dir_mask, F + D, ldf);
  FSN_LAUNCH_CHECK("k_posenc_ld");
  float* Bo = ws + T.off_bo;
  rc = gemm_xwT(rb, n, D / 2, ldf, F, ldf, W[L + 2], ldf, Bo, D / 2);
  if (rc != FSN_OK) return rc;
  k_bias_act<<<nb(n * (D / 2)), 256, 0, s>>>(Bo, D / 2, n, D / 2, b[L + 2], ACT_RELU);
  FSN_LAUNCH_CHECK("k_bias_act");
  rc = gemm_xwT(rb, n, 3, D / 2, Bo, D / 2, W[L + 3], D / 2, out, 4);
  if (rc != FSN_OK) return rc;
  k_bias_act<<<nb(n * 3), 256, 0, s>>>(out, 4, n, 3, b[L + 3], ACT_SIGMOID);
  FSN_LAUNCH_CHECK("k_bias_act");
  return FSN_OK;
}

extern "C" int fsnref_train_bwd(const fsn_mlp_desc* desc, const float* const* W, int64_t n, float* ws,
                                  const float* out, const float* d_out, float* const* dW, float* const* db,
                                  fsn_stream_t stream) {
  int rc = check_desc(desc);
  if (rc != FSN_OK) return rc;
  FSN_REQUIRE(W && dW && db, FSN_E_INVALID, "fsnref_train_bwd: null pointer");
  FSN_REQUIRE(n > 0 && ws && out && d_out, FSN_E_INVALID, "fsnref_train_bwd: needs the forward's workspace (n > 0)");
  FSN_REQUIRE(n < (1ll << 31), FSN_E_UNSUPPORTED, "fsnref_train_bwd: n too large for one call");
  hipStream_t s = as_stream(stream);
  RocBlas* rb;
  rc = rb_get(s, &rb);
  if (rc != FSN_OK) return rc;
  TrainLayout T;
  make_layout(*desc, n, T);
  const int L = T.L, D = T.D, Dh = D / 2, ldf = D + T.d_de;
  float* G0 = ws + T.off_g0; float* G1 = ws + T.off_g1; float* dZ = ws + T.off_dz; float* part = ws + T.off_part;
  const float* Bo = ws + T.off_bo; const float* F = ws + T.off_f;
  const float* HL = ws + T.off_h[L - 1];
  const int ldh = T.stride[L - 1];
#define FSN_RC(expr) do { rc = (expr); if (rc != FSN_OK) return rc; } while (0)
  // rgb head: dZ = d_rgb * rgb (1 - rgb)
  k_sigmoid_bwd<<<nb(n * 3), 256, 0, s>>>(d_out, out, n, dZ);
  FSN_LAUNCH_CHECK("k_sigmoid_bwd");
  FSN_RC(gemm_dyTx(rb, n, 3, Dh, dZ, 3, Bo, Dh, dW[L + 3], Dh));
  FSN_RC(colsum(s, dZ, 3, n, 3, part, db[L + 3]));
  FSN_RC(gemm_dyw(rb, n, 3, Dh, dZ, 3, W[L + 3], Dh, G0, Dh));  // dBo
  k_relu_bwd<<<nb(n * Dh), 256, 0, s>>>(G0, Dh, Bo, Dh, n, Dh);
  FSN_LAUNCH_CHECK("k_relu_bwd");
  // branch
  FSN_RC(gemm_dyTx(rb, n, Dh, ldf, G0, Dh, F, ldf, dW[L + 2], ldf));
  FSN_RC(colsum(s, G0, Dh, n, Dh, part, db[L + 2]));
  FSN_RC(gemm_dyw(rb, n, Dh, ldf, G0, Dh, W[L + 2], ldf, G1, ldf));  // dF (first D columns = d feat)
  // connection (no activation)
  FSN_RC(gemm_dyTx(rb, n, D, D, G1, ldf, HL, ldh, dW[L + 1], D));
  FSN_RC(colsum(s, G1, ldf, n, D, part, db[L + 1]));
  FSN_RC(gemm_dyw(rb, n, D, D, G1, ldf, W[L + 1], D, G0, D));  // dH (through connection)
  // sigma head: dw_sigma = d_sigma^T H, db_sigma = sum d_sigma, dH += d_sigma w_sigma
  FSN_RC(gemm_dyTx(rb, n, 1, D, d_out + 3, 4, HL, ldh, dW[L], D));
  FSN_RC(colsum(s, d_out + 3, 4, n, 1, part, db[L]));
  k_add_rank1<<<nb(n * D), 256, 0, s>>>(G0, D, n, D, d_out, W[L]);
  FSN_LAUNCH_CHECK("k_add_rank1");
  float* dH = G0;
  int ldg = D;
  float* other = G1;
  for (int l = L - 1; l >= 0; --l) {
    const float* H = ws + T.off_h[l];
    k_relu_bwd<<<nb(n * D), 256, 0, s>>>(dH, ldg, H, T.stride[l], n, D);
    FSN_LAUNCH_CHECK("k_relu_bwd");
    const float* in = l == 0 ? ws + T.off_xin : ws + T.off_h[l - 1];
    const int in_k = l == 0 ? T.d_pe : T.stride[l - 1];
    FSN_RC(gemm_dyTx(rb, n, D, in_k, dH, ldg, in, in_k, dW[l], in_k));
    FSN_RC(colsum(s, dH, ldg, n, D, part, db[l]));
    if (l > 0) {
      FSN_RC(gemm_dyw(rb, n, D, in_k, dH, ldg, W[l], in_k, other, in_k));  // d[h_{l-1} | x_in]: x_in part unused
      float* t = dH; dH = other; other = t;
      ldg = in_k;
    }
  }
#undef FSN_RC
  return FSN_OK;
}

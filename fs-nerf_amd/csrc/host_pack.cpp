// host_pack.cpp — the host side of the weight packer as plain C++ (no HIP): blob geometry (build_geom), the packer's
// index code (mlp_pack.hpp) and the two entry points that need no device.  Built ONLY by `make sanitize` into
// libfsnerf_host_san.so with -fsanitize=address,undefined: GPU AddressSanitizer is not available on this pool, so
// the part of the library that can run under a sanitizer - every index computation of the blob format - runs there
// (tests/test_pack_layout.py with FSN_LIB_PATH pointing at it).
#include <cstdarg>
#include <cstdio>

#include "mlp_pack.hpp"

static thread_local char g_err[512] = "";
static void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int fsn_version(void) { return 200; }
extern "C" const char* fsn_last_error(void) { return g_err; }

extern "C" int64_t fsn_mlp_blob_bytes(const fsn_mlp_desc* desc, int prec) {
  if (!desc) { set_error("fsn_mlp_blob_bytes: null desc"); return FSN_E_INVALID; }
  fsn::NetGeom G;
  const char* why;
  const int rc = fsn::build_geom(*desc, prec, G, &why);
  if (rc != FSN_OK) { set_error("fsn_mlp_blob_bytes: %s", why); return rc; }
  return G.total_bytes;
}

extern "C" int fsn_mlp_pack_host(const fsn_mlp_desc* desc, int prec, const float* const* weights,
                                 const float* const* biases, void* blob_host) {
  if (!blob_host) { set_error("fsn_mlp_pack_host: null blob"); return FSN_E_INVALID; }
  const char* why;
  const int rc = fsn::pack_blob_host(desc, prec, weights, biases, blob_host, &why);
  if (rc != FSN_OK) set_error("fsn_mlp_pack_host: %s", why);
  return rc;
}

extern "C" int fsn_mlp_pack_scaled_host(const fsn_mlp_desc* desc, int prec, const float* const* weights,
                                        const float* const* biases, const int32_t* layer_exps, void* blob_host) {
  if (!blob_host) { set_error("fsn_mlp_pack_scaled_host: null blob"); return FSN_E_INVALID; }
  const char* why;
  const int rc = fsn::pack_blob_host(desc, prec, weights, biases, blob_host, &why, layer_exps);
  if (rc != FSN_OK) set_error("fsn_mlp_pack_scaled_host: %s", why);
  return rc;
}

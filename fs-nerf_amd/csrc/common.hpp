// common.hpp — error plumbing and small device helpers shared by all translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "fsnerf_hip.h"

namespace fsn {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define FSN_REQUIRE(cond, code, ...)   \
  do {                                 \
    if (!(cond)) {                     \
      ::fsn::set_error(__VA_ARGS__);   \
      return (code);                   \
    }                                  \
  } while (0)

#define FSN_HIP(expr)                                        \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return ::fsn::hip_fail(_e, #expr); \
  } while (0)

// launch check: hipGetLastError right after <<<>>> (no sync; graph-capture safe)
#define FSN_LAUNCH_CHECK(name)                                         \
  do {                                                                 \
    hipError_t _e = hipGetLastError();                                 \
    if (_e != hipSuccess) return ::fsn::hip_fail(_e, "launch " name);  \
  } while (0)

static inline hipStream_t as_stream(fsn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int WAVE = 64;

// ---- debug build (-DFSN_DEBUG: `make debug` -> libfsnerf_hip_dbg.so; SURVEY 5 "LDS bounds asserts in debug builds").
// GPU AddressSanitizer is not available on this pool and a trapping kernel can take the node down, so an out-of-range
// index into one of the kernels' LDS arrays is RECORDED (count, site = source line, index, capacity, in a device word
// array of the translation unit) and clamped instead of trapped; fsn_debug_report() hands the record to the host and
// tests/test_debug_build.py asserts that it is empty after the small-shape parity tests ran through the debug library.
//   FSN_AT(arr, i)        arr[i]       with 0 <= i < extent(arr)
//   FSN_SPAN(arr, i, n)   &arr[i]      with 0 <= i and i + n <= extent(arr)   (pointers handed to device functions)
#ifdef FSN_DEBUG
#define FSN_DEBUG_DEFINE_RECORD(name) __device__ unsigned name[4] = {0u, 0u, 0u, 0u};
template <class T, int N>
__device__ __forceinline__ T& fsn_dbg_at(T (&arr)[N], long long i, int line, unsigned* rec) {
  if (i < 0 || i >= N) {
    if (atomicAdd(rec, 1u) == 0u) { rec[1] = (unsigned)line; rec[2] = (unsigned)i; rec[3] = (unsigned)N; }
    i = i < 0 ? 0 : N - 1;
  }
  return arr[i];
}
template <class T, int N>
__device__ __forceinline__ T* fsn_dbg_span(T (&arr)[N], long long i, long long n, int line, unsigned* rec) {
  if (i < 0 || n < 0 || i + n > N) {
    if (atomicAdd(rec, 1u) == 0u) { rec[1] = (unsigned)line; rec[2] = (unsigned)(i + n); rec[3] = (unsigned)N; }
    i = 0;
  }
  return arr + i;
}
#define FSN_AT(arr, i) ::fsn::fsn_dbg_at(arr, (long long)(i), __LINE__, FSN_DEBUG_RECORD)
#define FSN_SPAN(arr, i, n) ::fsn::fsn_dbg_span(arr, (long long)(i), (long long)(n), __LINE__, FSN_DEBUG_RECORD)
#else
#define FSN_DEBUG_DEFINE_RECORD(name)
#define FSN_AT(arr, i) (arr)[i]
#define FSN_SPAN(arr, i, n) ((arr) + (i))
#endif

}  // namespace fsn

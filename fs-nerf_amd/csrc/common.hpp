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

}  // namespace fsn

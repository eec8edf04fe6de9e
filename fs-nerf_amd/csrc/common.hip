// common.hip — thread-local error string, version, device query.
#include "common.hpp"

#include <cstring>
#include <mutex>

namespace fsn {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return FSN_E_HIP;
}
}  // namespace fsn

extern "C" int fsn_version(void) { return 100; }
extern "C" const char* fsn_last_error(void) { return fsn::g_err; }

extern "C" int fsn_device_cus(void) {
  int dev = 0;
  FSN_HIP(hipGetDevice(&dev));
  int cus = 0;
  FSN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  return cus;
}

// C-ABI plumbing shared by every entry point: thread-local error string, version, device probe.
#include "common.h"

namespace mspi {

static thread_local char g_err[512] = "";

int* g_status_word = nullptr;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace mspi

extern "C" int mspi_version(void) { return MSPI_ABI_VERSION; }

extern "C" const char* mspi_last_error(void) { return mspi::g_err; }

extern "C" int mspi_set_status_word(int32_t* device_visible_word) {
  mspi::g_status_word = device_visible_word;
  return MSPI_OK;
}

extern "C" int mspi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

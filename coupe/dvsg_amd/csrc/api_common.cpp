// Error reporting and version queries of the C ABI (include/dvsg_amd.h).
#include "common.h"

namespace dvsg {

char *error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

int fail(int status, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return status;
}

}  // namespace dvsg

extern "C" {
int dvsg_abi_version(void) { return DVSG_ABI_VERSION; }
const char *dvsg_last_error_string(void) { return dvsg::error_buffer(); }
const char *dvsg_target_arch(void) { return "gfx950"; }
}

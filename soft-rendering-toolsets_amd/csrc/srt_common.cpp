// Error channel of the C ABI: thread-local message + status helper.
#include "srt_common.h"

namespace srt {

char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return status;
}

}  // namespace srt

extern "C" const char* srt_last_error(void) { return srt::error_buffer(); }

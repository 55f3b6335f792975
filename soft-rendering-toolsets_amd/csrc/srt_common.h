// Shared host-side helpers of the C ABI (error channel, HIP status check).
#ifndef SRT_COMMON_H
#define SRT_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "srt_raster.h"

namespace srt {

// thread-local message returned by srt_last_error()
char* error_buffer();
int fail(int status, const char* fmt, ...);

}  // namespace srt

#define SRT_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return srt::fail(SRT_ERR_HIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                       __LINE__);                                                              \
  } while (0)

#endif

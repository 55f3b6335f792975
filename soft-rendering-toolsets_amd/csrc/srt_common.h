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

// Device buffers that live for one C-ABI call: freed when the call returns, on the error paths of SRT_HIP too.
#ifdef __cplusplus
#include <vector>
namespace srt {
class DeviceScratch {
 public:
  DeviceScratch() = default;
  DeviceScratch(const DeviceScratch&) = delete;
  DeviceScratch& operator=(const DeviceScratch&) = delete;
  ~DeviceScratch() { for (void* p : owned_) (void)hipFree(p); }
  template <typename T>
  hipError_t alloc(T** out, size_t bytes) {
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) { owned_.push_back(p); *out = static_cast<T*>(p); }
    return e;
  }
 private:
  std::vector<void*> owned_;
};
}  // namespace srt
#endif

#endif

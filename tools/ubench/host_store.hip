// How fast can a kernel write a 4 MiB framebuffer into pinned (hipHostRegister'ed) host memory, the way the last workgroup of a band
// of tiles would (one wave per 32 KB band, 8 bytes per lane per store)?  Compared with hipMemcpyAsync of the same bytes.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/host_store.hip -o /tmp/host_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void copy_bands(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, size_t words_per_band) {
  const unsigned long long* s = src + blockIdx.x * words_per_band;
  unsigned long long* d = dst + blockIdx.x * words_per_band;
  for (size_t i = threadIdx.x; i < words_per_band; i += blockDim.x)
    d[i] = __hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void spin(unsigned long long cycles) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} }
int main() {
  const size_t bytes = 4u << 20;
  std::vector<unsigned char> host(bytes + 4096);
  unsigned char* hp = (unsigned char*)(((uintptr_t)host.data() + 4095) & ~(uintptr_t)4095);
  CK(hipHostRegister(hp, bytes, hipHostRegisterDefault));
  void* hdev = nullptr; CK(hipHostGetDevicePointer(&hdev, hp, 0));
  void* dsrc = nullptr; CK(hipMalloc(&dsrc, bytes)); CK(hipMemset(dsrc, 0x5a, bytes));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int bands : {64, 128, 256, 512, 1024}) {
    for (int threads : {64, 256}) {
      const size_t wpb = bytes / 8 / bands;
      float best = 1e9f;
      for (int it = 0; it < 8; it++) {
        memset(hp, 0, bytes);
        CK(hipEventRecord(a, s));
        copy_bands<<<bands, threads, 0, s>>>((const unsigned long long*)dsrc, (unsigned long long*)hdev, wpb);
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
      }
      size_t bad = 0; for (size_t i = 0; i < bytes; i++) bad += hp[i] != 0x5a;
      printf("kernel stores: %4d bands x %3d threads: %.1f us (%.1f GB/s) bad %zu\n", bands, threads, best * 1e3, bytes / best / 1e6, bad);
    }
  }
  float best = 1e9f;
  for (int it = 0; it < 8; it++) {
    CK(hipEventRecord(a, s)); CK(hipMemcpyAsync(hp, dsrc, bytes, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  printf("hipMemcpyAsync D2H: %.1f us (%.1f GB/s)\n", best * 1e3, bytes / best / 1e6);
  return 0;
}

// Does a kernel's store into hipHostRegister'ed memory reach the pages the CPU sees when the SAME virtual range is freed, allocated
// and registered again (what DrawSVG::resize does with its framebuffer)?  A 4 MiB buffer is mmap'ed, registered, written by a kernel
// with a round number, checked, unregistered, munmap'ed - and again, many rounds; the copy engine (hipMemcpyAsync D2H) for comparison.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/host_reregister.hip -o tools/ubench/host_reregister
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint32_t* dst, size_t n, uint32_t v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v + (uint32_t)i;
}
int main() {
  const size_t bytes = 4u << 20, n = bytes / 4;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t* d = nullptr; CK(hipMalloc(&d, bytes));
  for (int mode = 0; mode < 2; mode++) {
    int bad_rounds = 0; size_t bad_words = 0; void* last = nullptr; int same_va = 0;
    for (uint32_t round = 1; round <= 40; round++) {
      uint32_t* hp = (uint32_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (hp == MAP_FAILED) return 2;
      same_va += hp == last; last = hp;
      memset(hp, 0xff, bytes);
      CK(hipHostRegister(hp, bytes, hipHostRegisterDefault));
      void* hdev = nullptr; CK(hipHostGetDevicePointer(&hdev, hp, 0));
      for (int frame = 0; frame < 3; frame++) {
        memset(hp, 0xff, bytes);
        if (mode == 0) fill<<<256, 256, 0, s>>>((uint32_t*)hdev, n, round * 1000u + frame);
        else { fill<<<256, 256, 0, s>>>(d, n, round * 1000u + frame); CK(hipMemcpyAsync(hp, d, bytes, hipMemcpyDeviceToHost, s)); }
        CK(hipStreamSynchronize(s));
        size_t bad = 0; for (size_t i = 0; i < n; i++) bad += hp[i] != round * 1000u + frame + (uint32_t)i;
        if (bad) { bad_rounds++; bad_words += bad; if (bad_rounds <= 3) printf("  mode %d round %u frame %d: %zu stale words, first page %zu\n", mode, round, frame, bad, [&]{ for (size_t i = 0; i < n; i++) if (hp[i] != round * 1000u + frame + (uint32_t)i) return i / 1024; return (size_t)0; }()); }
      }
      CK(hipHostUnregister(hp));
      munmap(hp, bytes);
    }
    printf("%s: 40 rounds x 3 frames, %d bad frames (%zu words), same address as the round before: %d times\n", mode == 0 ? "kernel stores" : "copy engine", bad_rounds, bad_words, same_va);
  }
  return 0;
}

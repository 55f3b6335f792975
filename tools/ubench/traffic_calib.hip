// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access patterns of the path tracer's ray state (the guide
// calibrates them for 16-B-per-lane streaming only).  Every kernel moves a known number of bytes over a buffer far larger
// than L2 + Infinity Cache; tools/traffic_calib.sh runs it under --pmc and prints counter / bytes.
//   rec_store<L>   lane t stores level L of its record row: two float4 at row t (512 B apart between lanes)  -> 32 B / lane
//   rec_load       lane t loads levels 0..2 of its row (96 B of one 128-B line)                                -> 96 B / lane
//   dword_store    lane t stores one dword, coalesced                                                          ->  4 B / lane
//   f4_stream      lane t stores one float4, coalesced (the guide's calibrated case)                           -> 16 B / lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void rec_store(float4* rec, size_t lanes, int level) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  const float v = (float)t;
  rec[(t * 16 + level) * 2 + 0] = make_float4(v, v, v, 1.0f);
  rec[(t * 16 + level) * 2 + 1] = make_float4(v, v, v, 2.0f);
}
__global__ void rec_load(const float4* rec, size_t lanes, float* out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  float acc = 0.0f;
  for (int level = 0; level < 3; level++) {
    const float4 a = rec[(t * 16 + level) * 2 + 0], b = rec[(t * 16 + level) * 2 + 1];
    acc += a.x + a.w + b.x + b.w;
  }
  if (acc == 12345.678f) out[0] = acc;   // never true: keeps the loads
}
__global__ void dword_store(float* p, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = (float)t;
}
__global__ void f4_stream(float4* p, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = make_float4((float)t, 0, 0, 0);
}

int main() {
  const size_t lanes = (size_t)8 << 20;                 // 8 Mi lanes x 512 B rows = 4 GiB
  float4* rec; float* out;
  CK(hipMalloc(&rec, lanes * 512));
  CK(hipMalloc(&out, 4096));
  CK(hipMemset(rec, 0, lanes * 512));
  const dim3 block(256), grid((unsigned)(lanes / 256));
  for (int level = 0; level < 4; level++) rec_store<<<grid, block>>>(rec, lanes, level);   // one 128-B line per lane, a quarter at a time
  rec_load<<<grid, block>>>(rec, lanes, out);
  const size_t n = (size_t)1 << 30;                     // 4 GiB of dwords / 4 GiB of float4 (n / 4)
  dword_store<<<dim3((unsigned)(n / 256)), block>>>((float*)rec, n);
  f4_stream<<<dim3((unsigned)(n / 4 / 256)), block>>>(rec, n / 4);
  CK(hipDeviceSynchronize());
  printf("bytes: rec_store %zu per launch, rec_load %zu, dword_store %zu, f4_stream %zu\n", lanes * 32, lanes * 96, n * 4, n * 4);
  return 0;
}

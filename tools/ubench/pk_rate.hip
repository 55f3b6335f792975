// Micro-benchmark: issue cost of v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 against their scalar-per-lane forms on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
  f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = {3.0f, 4.0f}, a2 = {5.0f, 6.0f}, a3 = {7.0f, 8.0f};
  f2 a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
  const f2 m = {1.0000001f, 0.9999999f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {       // 8 independent v_mul_f32
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x));)
    } else if (KIND == 1) {  // 8 independent v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (KIND == 2) {  // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (KIND == 3) {  // v_pk_mul_f32 with a broadcast (op_sel_hi) scalar-pair operand
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[1,0]\n"
                   "v_pk_mul_f32 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[1,0]\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));)
    } else if (KIND == 4) {  // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x));)
    } else {                 // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND>
void run(const char* name, int waves_per_simd) {
  const int iters = 2000, block = 64 * 4 * waves_per_simd, grid = 256;   // one block per CU, `waves_per_simd` waves on each SIMD
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * block * grid); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<grid, block>>>(out, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<grid, block>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 64.0;    // instructions per wave
  // per-SIMD instruction time from wall clock: waves_per_simd * n instructions per SIMD
  printf("%-28s waves/SIMD %d: %.2f ms, %.2f ns per wave-instruction per SIMD (wall), s_memtime ticks/instr %.3f\n", name, waves_per_simd, ms,
         ms * 1e6 / (n * waves_per_simd), (double)c / n);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_mul_f32", w); run<1>("v_pk_mul_f32", w); run<2>("v_pk_add_f32", w); run<3>("v_pk_mul_f32 sgpr bcast", w);
    run<4>("v_fma_f32", w); run<5>("v_pk_fma_f32", w);
  }
  return 0;
}

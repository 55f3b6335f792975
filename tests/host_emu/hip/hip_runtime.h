// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets tests/host_emu/flat_host.cpp compile the path tracer's
// per-lane device headers (pt_device.h, pt_trace.h, pt_flat.h) with g++ and run them one lane at a time on the
// CPU, so that the traversal logic can be checked against the oracle without a GPU.  Never part of the product.
#ifndef SRT_TEST_HOST_EMU_HIP_RUNTIME_H
#define SRT_TEST_HOST_EMU_HIP_RUNTIME_H
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define SRT_PIN_VGPR(x) ((void)(x))
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned long long __ballot(int pred) { return pred ? 1ull : 0ull; }   // a "wave" of one lane
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { const uint32_t old = *p; *p = old + v; return old; }   // one lane: no race
#endif

// Device-side arithmetic of the path tracer: vector / matrix helpers with the reference's operation
// order, SRT-RNG v1, SRT-MATH v2 (glibc's sincosf algorithm), primitive intersection.
//
// Everything here must round exactly like the x86-64 build of the reference: fp32 throughout, fp64
// only where the reference widens (unqualified sqrt/pow), no FMA contraction (-ffp-contract=off),
// IEEE divide/sqrt (hipcc default), denormals preserved.  Reference paths are relative to
// /root/reference/Assignments/Scotty3D/src/.
#ifndef SRT_PT_DEVICE_H
#define SRT_PT_DEVICE_H

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "pt_scene.h"

namespace srt {

#define SRT_DEV __device__ __forceinline__

constexpr float kEps = 0.00001f;                                   // EPS_F, lib/mathlib.h:16
constexpr float kPi = 3.14159265358979323846264338327950288f;      // PI_F,  lib/mathlib.h:17

struct V3 { float x, y, z; };
struct Spec { float r, g, b; };
struct Ray { V3 o, d; float b0, b1; };

// ---- IEEE divide / square root with the range handling hoisted out of the common case ----
// hipcc expands a correctly rounded fp32 divide to v_div_scale x2, v_rcp, a five-step FMA refinement, v_div_fmas and
// v_div_fixup, and a square root to v_sqrt plus a +-1 ulp correction wrapped in denormal scaling and a class
// fix-up.  When no operand needs scaling or fixing (every lane's operands are normal numbers well inside the
// exponent range: v_div_scale returns its input, VCC = 0, v_div_fixup and the sqrt fix-ups pass the result
// through) what remains is the refinement itself, and the part of it that only depends on the denominator can
// be shared by the quotients of one triangle test (u, v, t over det) or one normalisation (x, y, z over the norm).
// Used by the wave kernel's batch tests only (two or three rays at a time, so the serial refinement chains overlap).
// The fast path is taken when the whole wave qualifies; any other operand sends the wave through the compiler's
// own sequence, so the results are those of the plain `/` and sqrtf for every input
// (tests/test_pt_gpu.py::test_exact_division_and_sqrt_fast_paths sweeps both against them).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SRT_PLAIN_DIV_SQRT)
#define SRT_EXACT_FAST_PATHS 1
#else
#define SRT_EXACT_FAST_PATHS 0
#endif

#if SRT_EXACT_FAST_PATHS
SRT_DEV float div_refine(float num, float den, float r) {
  float q = num * r;
  const float e1 = __builtin_fmaf(-den, q, num);
  q = __builtin_fmaf(e1, r, q);
  const float e2 = __builtin_fmaf(-den, q, num);
  return __builtin_fmaf(e2, r, q);
}
#endif
// N square roots / 3 N quotients (the N rays of a batch): one range verdict for all operands, and the
// refinement chains of the three rays are independent instruction streams inside one basic block.
// zero[r] = "x[r] is exactly +0" as the caller knows it (origin on the triangle's plane: t = 0; a ray without a hit:
// distance of a point from itself): v_sqrt(0) = 0 and neither correction applies, so a zero may take the fast path.
template <int N>
SRT_DEV void sqrtN(const float* x, const bool* zero, float* o) {
#if SRT_EXACT_FAST_PATHS
  bool bad = false;
#pragma unroll
  for (int r = 0; r < N; r++) bad = bad || (!zero[r] && !(x[r] >= 0x1p-96f && x[r] <= FLT_MAX));
  if (__ballot(bad) == 0ull) {
#pragma unroll
    for (int r = 0; r < N; r++) {
      const float s = __builtin_amdgcn_sqrtf(x[r]);
      const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
      const float rm = __builtin_fmaf(-sm, s, x[r]), rp = __builtin_fmaf(-sp, s, x[r]);
      float v = (0.0f >= rm) ? sm : s;
      v = (0.0f < rp) ? sp : v;
      o[r] = v;
    }
    return;
  }
#endif
#pragma unroll
  for (int r = 0; r < N; r++) o[r] = sqrtf(x[r]);
}
// q[r][j] = n[r][j] / den[r].  SHARED_C2: the caller passes the same numerator in column 2 for all rays (the
// numerator of t), which is exactly zero for an origin on the triangle's plane; such a lane takes the fast path as
// well and v_div_fixup gives the signed zero the full sequence would.
template <int N, bool SHARED_C2>
SRT_DEV void divNx3(const float (*n)[3], const float* den, float (*q)[3]) {
#if SRT_EXACT_FAST_PATHS
  const float lo = 0x1p-40f, hi = 0x1p40f;     // quotient exponents stay within +-80: no scaling case of v_div_scale
  float mn = fabsf(den[0]), mx = fabsf(den[0]);
#pragma unroll
  for (int r = 1; r < N; r++) { mn = fminf(mn, fabsf(den[r])); mx = fmaxf(mx, fabsf(den[r])); }
#pragma unroll
  for (int r = 0; r < N; r++) {
    if (SHARED_C2) {
      mn = fminf(mn, fminf(fabsf(n[r][0]), fabsf(n[r][1])));
      mx = fmaxf(mx, fmaxf(fabsf(n[r][0]), fabsf(n[r][1])));
    } else {
      mn = fminf(fminf(mn, fabsf(n[r][0])), fminf(fabsf(n[r][1]), fabsf(n[r][2])));
      mx = fmaxf(fmaxf(mx, fabsf(n[r][0])), fmaxf(fabsf(n[r][1]), fabsf(n[r][2])));
    }
  }
  if (SHARED_C2) {
    const float c2 = (n[0][2] == 0.0f) ? 1.0f : fabsf(n[0][2]);
    mn = fminf(mn, c2); mx = fmaxf(mx, c2);
  }
  // (min / max skip a NaN operand: it stays a NaN through the refinement, as it would through the full sequence)
  if (__ballot(!(mn >= lo && mx <= hi)) == 0ull) {
#pragma unroll
    for (int r = 0; r < N; r++) {
      float rc = __builtin_amdgcn_rcpf(den[r]);
      const float e = __builtin_fmaf(-den[r], rc, 1.0f);
      rc = __builtin_fmaf(e, rc, rc);
#pragma unroll
      for (int j = 0; j < 3; j++) q[r][j] = div_refine(n[r][j], den[r], rc);
      if (SHARED_C2) q[r][2] = __builtin_amdgcn_div_fixupf(q[r][2], den[r], n[r][2]);
    }
    return;
  }
#endif
#pragma unroll
  for (int r = 0; r < N; r++)
#pragma unroll
    for (int j = 0; j < 3; j++) q[r][j] = n[r][j] / den[r];
}

SRT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
SRT_DEV V3 v3p(const float* p) { return v3(p[0], p[1], p[2]); }
SRT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
SRT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
SRT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }   // Vec3*float == float*Vec3
SRT_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
SRT_DEV V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
SRT_DEV float dot(V3 l, V3 r) { return l.x * r.x + l.y * r.y + l.z * r.z; }
SRT_DEV V3 cross(V3 l, V3 r) { return v3(l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x); }
SRT_DEV float norm2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
SRT_DEV float norm(V3 a) { return sqrtf(norm2(a)); }
SRT_DEV V3 unit(V3 a) { return a / norm(a); }
SRT_DEV float std_min(float a, float b) { return (b < a) ? b : a; }
SRT_DEV float std_max(float a, float b) { return (a < b) ? b : a; }

SRT_DEV Spec spec(float r, float g, float b) { Spec s; s.r = r; s.g = g; s.b = b; return s; }
SRT_DEV Spec operator+(Spec a, Spec b) { return spec(a.r + b.r, a.g + b.g, a.b + b.b); }
SRT_DEV Spec operator-(Spec a, Spec b) { return spec(a.r - b.r, a.g - b.g, a.b - b.b); }
SRT_DEV Spec operator*(Spec a, Spec b) { return spec(a.r * b.r, a.g * b.g, a.b * b.b); }
SRT_DEV Spec operator*(Spec a, float s) { return spec(a.r * s, a.g * s, a.b * s); }
SRT_DEV float luma(Spec a) { return 0.2126f * a.r + 0.7152f * a.g + 0.0722f * a.b; }   // lib/spectrum.h:111
SRT_DEV bool finite_f(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }
SRT_DEV bool valid(Spec a) { return finite_f(a.r) && finite_f(a.g) && finite_f(a.b); }  // lib/spectrum.h:115

// Mat4 * Vec3 (projective, lib/mat4.h:125-131) and Mat4::rotate (w = 0; the 0*col3 term is kept).
SRT_DEV V3 mat_point(const Mat4& m, V3 v) {
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; j++) o[j] = ((m.c[0][j] * v.x + m.c[1][j] * v.y) + m.c[2][j] * v.z) + m.c[3][j] * 1.0f;
  return v3(o[0] / o[3], o[1] / o[3], o[2] / o[3]);
}
SRT_DEV V3 mat_rotate(const Mat4& m, V3 v) {
  float o[3];
#pragma unroll
  for (int j = 0; j < 3; j++) o[j] = ((m.c[0][j] * v.x + m.c[1][j] * v.y) + m.c[2][j] * v.z) + m.c[3][j] * 0.0f;
  return v3(o[0], o[1], o[2]);
}
// m.T().rotate(v): component j = v.x*m[j][0] + v.y*m[j][1] + v.z*m[j][2] + 0*m[j][3]
SRT_DEV V3 mat_rotate_transposed(const Mat4& m, V3 v) {
  float o[3];
#pragma unroll
  for (int j = 0; j < 3; j++) o[j] = ((m.c[j][0] * v.x + m.c[j][1] * v.y) + m.c[j][2] * v.z) + m.c[j][3] * 0.0f;
  return v3(o[0], o[1], o[2]);
}

// Ray::transform (lib/ray.h:31-37)
SRT_DEV void ray_transform(Ray& r, const Mat4& m) {
  r.o = mat_point(m, r.o);
  r.d = mat_rotate(m, r.d);
  const float d = norm(r.d);
  r.b0 *= d;
  r.b1 *= d;
  r.d = r.d / d;
}
SRT_DEV Ray make_ray(V3 o, V3 d, float b0, float b1) {  // explicit Ray(point, dir, bounds): dir.unit()
  Ray r; r.o = o; r.d = unit(d); r.b0 = b0; r.b1 = b1; return r;
}
SRT_DEV V3 ray_at(const Ray& r, float t) { return r.o + r.d * t; }

// ---------------------------------------------------------------------------------------------------
// SRT-RNG v1 (replaces util/rand.cpp:13-25): PCG32 XSH-RR keyed by splitmix64(seed, pixel, sample)
// ---------------------------------------------------------------------------------------------------
struct Rng {
  uint64_t state, inc;
  uint32_t draws;
  SRT_DEV void key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    const uint64_t k = ((uint64_t)pixel << 32) | sample;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (k + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    inc = (k << 1) | 1;
    state = z * 6364136223846793005ull + inc;
    draws = 0;
  }
  SRT_DEV uint32_t next() {
    const uint64_t old = state;
    state = old * 6364136223846793005ull + inc;
    const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    const uint32_t rot = (uint32_t)(old >> 59);
    draws++;
    return (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  SRT_DEV float unit() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
  SRT_DEV int integer(int lo, int hi) { return lo + (int)(((uint64_t)next() * (uint64_t)(uint32_t)(hi - lo)) >> 32); }
  SRT_DEV bool coin(float p) { return unit() < p; }
};

// ---------------------------------------------------------------------------------------------------
// SRT-MATH v2: glibc 2.35 sinf/cosf (sysdeps/ieee754/flt-32/{s_sinf,s_cosf}.c, sincosf.h, sincosf_data.c)
// evaluated in fp64 without FMA.  Valid for |x| < 120 (the renderer uses [0, 2pi] and [-1, 1]).
// ---------------------------------------------------------------------------------------------------
SRT_DEV float sincos_poly(double x, double x2, bool negated_cos_table, int n) {
  const double sgn = negated_cos_table ? -1.0 : 1.0;  // table[1] negates the cosine coefficients only
  if ((n & 1) == 0) {
    const double x3 = x * x2;
    const double s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
    const double x7 = x3 * x2;
    const double s = x + x3 * -0x1.555545995a603p-3;
    return (float)(s + x7 * s1);
  } else {
    const double x4 = x2 * x2;
    const double c2 = (sgn * -0x1.6c087e89a359dp-10) + x2 * (sgn * 0x1.99343027bf8c3p-16);
    const double c1 = (sgn * 0x1p0) + x2 * (sgn * -0x1.ffffffd0c621cp-2);
    const double x6 = x4 * x2;
    const double c = c1 + x4 * (sgn * 0x1.55553e1068f19p-5);
    return (float)(c + x6 * c2);
  }
}
SRT_DEV float srt_sincosf(float y, int want_cos) {
  double x = (double)y;
  const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
  if (top < 0x3f4u) {                      // abstop12(y) < abstop12(pi/4 = 0x1.921FB6p-1f)
    if (top < 0x398u) return want_cos ? 1.0f : y;   // |y| < 2^-12
    return sincos_poly(x, x * x, false, want_cos);
  }
  if (top < 0x42fu) {                      // abstop12(120.0f)
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
    const int q = n & 3;
    const double s = (q == 1 || q == 2) ? -1.0 : 1.0;
    return sincos_poly(x * s, x * x, (n & 2) != 0, n ^ want_cos);
  }
  return __uint_as_float(0x7fc00000u);
}
// cosf(y) and sinf(y) together: both glibc routines reduce y the same way, so the reduction is shared.
SRT_DEV void srt_sincosf2(float y, float& c, float& s) {
  double x = (double)y;
  const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
  if (top < 0x3f4u) {
    if (top < 0x398u) { c = 1.0f; s = y; return; }
    const double x2 = x * x;
    c = sincos_poly(x, x2, false, 1);
    s = sincos_poly(x, x2, false, 0);
    return;
  }
  if (top < 0x42fu) {
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
    const int q = n & 3;
    const double sg = (q == 1 || q == 2) ? -1.0 : 1.0;
    const double xs = x * sg, x2 = x * x;
    c = sincos_poly(xs, x2, (n & 2) != 0, n ^ 1);
    s = sincos_poly(xs, x2, (n & 2) != 0, n);
    return;
  }
  c = s = __uint_as_float(0x7fc00000u);
}
// ---------------------------------------------------------------------------------------------------
// SRT-MATH v2, atan2f: glibc 2.35 __ieee754_atan2f / __atanf (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c — the
// fdlibm float code), fp32 throughout, no FMA.  Bit-identical to the host libm on 2e8 arguments (random bit
// patterns and the renderer's range); used by Spot_Light::sample (rays/light.cpp:22).
// ---------------------------------------------------------------------------------------------------
SRT_DEV float srt_atanf(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
  if (ix >= 0x4c000000) {                              // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;                 // NaN
    return (hx > 0) ? (atanhi[3] + atanlo[3]) : (-atanhi[3] - atanlo[3]);
  }
  int id;
  if (ix < 0x3ee00000) {                               // |x| < 0.4375
    if (ix < 0x31000000) return x;                     // |x| < 2^-29
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {                             // |x| < 1.1875
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }   // 7/16 <= |x| < 11/16
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                        // 11/16 <= |x| < 19/16
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }   // |x| < 2.4375
      else { id = 3; x = -1.0f / x; }
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f +
                   w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
  const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f +
                   w * -3.6531571299e-02f))));
  if (id < 0) return x - x * (s1 + s2);
  const float hi = id == 0 ? atanhi[0] : (id == 1 ? atanhi[1] : (id == 2 ? atanhi[2] : atanhi[3]));
  const float lo = id == 0 ? atanlo[0] : (id == 1 ? atanlo[1] : (id == 2 ? atanlo[2] : atanlo[3]));
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return (hx < 0) ? -r : r;
}

SRT_DEV float srt_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
              pi_lo = -8.7422776573e-08f;
  const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
  const int32_t hy = (int32_t)__float_as_uint(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;   // NaN
  if (hx == 0x3f800000) return srt_atanf(y);               // x = 1
  const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);   // 2 * sign(x) + sign(y)
  if (iy == 0) return (m < 2) ? y : (m == 2 ? pi + tiny : -pi - tiny);
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000)
      return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
    return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
  }
  if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                   // |y / x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0f;                    // |y| / x < -2^60
  else z = srt_atanf(fabsf(y / x));
  if (m == 0) return z;
  if (m == 1) return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}

// SRT-MATH v2, acosf: glibc 2.35 __ieee754_acosf (sysdeps/ieee754/flt-32/e_acosf.c, fdlibm's float code), fp32, no FMA.
// Bit-identical to the host libm on 7e8 arguments covering [-1, 1]; used by Samplers::Hemisphere::Uniform.
SRT_DEV float srt_acosf(float x) {
  const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
  const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) return (hx > 0) ? 0.0f : pi + 2.0f * pio2_lo;
  if (ix > 0x3f800000) return (x - x) / (x - x);
  if (ix < 0x3f000000 && ix <= 0x23000000) return pio2_hi + pio2_lo;       // |x| < 2^-57
  const float z = (ix < 0x3f000000) ? x * x : ((hx < 0) ? (1.0f + x) * 0.5f : (1.0f - x) * 0.5f);
  const float p = z * (1.6666667163e-01f + z * (-3.2556581497e-01f + z * (2.0121252537e-01f + z * (-4.0055535734e-02f +
                  z * (7.9153501429e-04f + z * 3.4793309169e-05f)))));
  const float q = 1.0f + z * (-2.4033949375e+00f + z * (2.0209457874e+00f + z * (-6.8828397989e-01f + z * 7.7038154006e-02f)));
  const float r = p / q;
  if (ix < 0x3f000000) return pio2_hi - (x - (pio2_lo - x * r));           // |x| < 0.5
  const float s = sqrtf(z);
  if (hx < 0) {                                                            // x < -0.5
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  }
  const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);     // x > 0.5
  const float c = (z - df * df) / (s + df);
  const float w = r * s + c;
  return 2.0f * (df + w);
}

SRT_DEV float srt_cosf(float x) { return srt_sincosf(x, 1); }
SRT_DEV float srt_sinf(float x) { return srt_sincosf(x, 0); }
// (float)pow(x, 2) / (float)pow(1 - c, 5) with the float promoted to double (student/bsdf.cpp:17-21,47,150)
SRT_DEV double pow2d(float x) { return (double)x * (double)x; }
SRT_DEV double pow5d(float x) { const double d = (double)x, d2 = d * d, d4 = d2 * d2; return d4 * d; }

// ---------------------------------------------------------------------------------------------------
// SRT-MATH v2, expf / powf: glibc 2.35 e_expf.c / e_powf.c (ARM optimized routines), x86-64 configuration
// (TOINT_INTRINSICS 0: the 0x1.8p52 shift, POWF_SCALE 1), fp64 with one final rounding.  The multiply-adds are fused
// as in glibc's FMA builds (__expf_fma / __powf_fma, what an AVX2 host's ifunc selects): with them the restatement is
// identical to the host libm for every float (expf) and for every normal x at the exponents tried (powf); the
// unfused forms differ for a handful of arguments.  powf: normal x > 0, finite y != 0, NaN otherwise.
// Used by the tone-mapping epilogue only (HDR_Image::tonemap_to, Spectrum::to_srgb).
static __device__ const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
static __device__ const double kPowfLog2Tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
SRT_DEV uint64_t dbl_bits(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
SRT_DEV double bits_dbl(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
// 2^(k/32) * 2^r from ki = bits of (k/32-scaled value + shift): table entry plus the exponent bits of ki
SRT_DEV double exp2f_scale(uint64_t ki) { return bits_dbl(kExp2fTab[ki % 32] + (ki << 47)); }
SRT_DEV float srt_expf(float x) {
  const uint32_t ux = __float_as_uint(x), abstop = (ux >> 20) & 0x7ff;
  if (abstop >= 0x42b) {                                 // |x| >= 88 or NaN
    if (ux == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);
    if (x < -0x1.9fe368p6f) return 0.0f;
  }
  const double xd = (double)x, invln2n = 0x1.71547652b82fep+5, shift = 0x1.8p+52;
  const double z = invln2n * xd;
  double kd = z + shift;
  const uint64_t ki = dbl_bits(kd);
  kd -= shift;
  const double r = __builtin_fma(invln2n, xd, -kd);
  const double s = exp2f_scale(ki);
  const double p = __builtin_fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
  const double r2 = r * r;
  double y = __builtin_fma(0x1.62e42ff0c52d6p-6, r, 1.0);
  y = __builtin_fma(p, r2, y);
  return (float)(y * s);
}
SRT_DEV float srt_powf(float x, float y) {
  const uint32_t ix = __float_as_uint(x), ay = __float_as_uint(y) & 0x7fffffffu;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || ay == 0 || ay >= 0x7f800000u) return __uint_as_float(0x7fc00000u);
  const uint32_t tmp = ix - 0x3f330000u;                 // log2_inline: x = 2^k z, z in [OFF, 2 OFF)
  const int i = (int)((tmp >> 19) % 16);
  const uint32_t top = tmp & 0xff800000u;
  const int k = (int32_t)top >> 23;
  const double z = (double)__uint_as_float(ix - top);
  const double r = __builtin_fma(z, kPowfLog2Tab[i][0], -1.0);
  const double y0 = kPowfLog2Tab[i][1] + (double)k;
  const double r2 = r * r;
  double yy = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
  const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
  const double r4 = r2 * r2;
  double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);
  q = __builtin_fma(p, r2, q);
  yy = __builtin_fma(yy, r4, q);
  const double ylogx = (double)y * yy;
  if (((dbl_bits(ylogx) >> 47) & 0xffff) >= (dbl_bits(126.0) >> 47)) {
    if (ylogx > 0x1.fffffffd1d571p+6) return __uint_as_float(0x7f800000u);
    if (ylogx <= -150.0) return 0.0f;
  }
  const double shift = 0x1.8p+47;                        // exp2_inline
  double kd = ylogx + shift;
  const uint64_t ki = dbl_bits(kd);
  kd -= shift;
  const double rr = ylogx - kd;
  const double s = exp2f_scale(ki);
  const double pz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
  const double rr2 = rr * rr;
  double o = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
  o = __builtin_fma(pz, rr2, o);
  return (float)(o * s);
}
// Spectrum::to_srgb (lib/spectrum.h:61-66) and the byte conversion of HDR_Image::tonemap_to (util/hdr_image.cpp:181):
// (unsigned char)std::round(v * 255.0f), i.e. on x86-64 cvttss2si and the low byte (0x80000000 when out of range / NaN).
SRT_DEV float to_srgb(float f) {
  if (f < 0.0031308f) return 12.92f * f;
  return 1.055f * srt_powf(f, 1.0f / 2.4f) - 0.055f;
}
SRT_DEV uint32_t srgb_byte(float v) {
  const float r = roundf(v * 255.0f);
  const int32_t i = (r >= -2147483648.0f && r < 2147483648.0f) ? (int32_t)r : INT32_MIN;
  return (uint32_t)i & 0xffu;
}

// ---------------------------------------------------------------------------------------------------
// BBox::hit (student/bbox.cpp:5-62): line/slab test; `times` is only narrowed when tmin/tmax fall inside it.
SRT_DEV bool box_hit(const Node& nd, const Ray& ray, float& tx, float& ty) {
  const float ix = 1.0f / ray.d.x, iy = 1.0f / ray.d.y, iz = 1.0f / ray.d.z;
  const bool sx = ix < 0, sy = iy < 0, sz = iz < 0;
  // bounds as values before the selects (an lvalue select would become a load from a selected address)
  const float mn0 = nd.mn[0], mn1 = nd.mn[1], mn2 = nd.mn[2], mx0 = nd.mx[0], mx1 = nd.mx[1], mx2 = nd.mx[2];
  float tmin = ((sx ? mx0 : mn0) - ray.o.x) * ix;
  float tmax = ((sx ? mn0 : mx0) - ray.o.x) * ix;
  const float tymin = ((sy ? mx1 : mn1) - ray.o.y) * iy;
  const float tymax = ((sy ? mn1 : mx1) - ray.o.y) * iy;
  if ((tmin > tymax) || (tymin > tmax)) return false;
  if (tymin > tmin) tmin = tymin;
  if (tymax < tmax) tmax = tymax;
  const float tzmin = ((sz ? mx2 : mn2) - ray.o.z) * iz;
  const float tzmax = ((sz ? mn2 : mx2) - ray.o.z) * iz;
  if ((tmin > tzmax) || (tzmin > tmax)) return false;
  if (tzmin > tmin) tmin = tzmin;
  if (tzmax < tmax) tmax = tzmax;
  if (tmin >= tx && tmin <= ty) tx = tmin;
  if (tmax >= tx && tmax <= ty) ty = tmax;
  return true;
}

// Triangle::hit (student/tri_mesh.cpp:32-111) on the precomputed {p0, e1, e2}.  Returns hit and, when hit,
// distance = |t*d| and the barycentric triple (u, v, t).
struct TriHit { bool hit; float dist, u, v, t; };
// Straight-line form: every quantity is computed for every lane and the verdict is assembled from the same
// comparisons the reference makes (det != 0; u, v, 1-u-v, t not negative; distance inside dist_bounds).  With
// det == 0 the quotients are inf/NaN and simply unused.  No branches: the three rays of a batch and the
// triangles of a leaf become independent instruction streams the scheduler can interleave.
SRT_DEV TriHit tri_hit(const Tri& g, const Ray& ray) {
  TriHit h;
  const V3 e1 = v3p(g.e1), e2 = v3p(g.e2);
  const V3 s = ray.o - v3p(g.p0);
  const V3 e1xd = cross(e1, ray.d);
  const float det = dot(e1xd, e2);
  const V3 sxe2 = cross(s, e2);
  const float nu = -1.0f * dot(sxe2, ray.d);
  const float nv = dot(e1xd, s);
  const float nt = -1.0f * dot(sxe2, e1);
  h.u = nu / det; h.v = nv / det; h.t = nt / det;
  const bool outside = (h.u < 0) || (h.v < 0) || ((1.0f - h.u - h.v) < 0) || (h.t < 0);
  h.dist = fabsf(norm(ray.d * h.t));
  const bool out_of_bounds = (h.dist < ray.b0) || (h.dist > ray.b1);
  h.hit = (det != 0) && !outside && !out_of_bounds;
  return h;
}

// Triangle::hit of one triangle for the N rays of a batch (shared origin): s, s x e2 and the numerator of t do
// not depend on the direction; the 3 N quotients and the N distances go through divNx3 / sqrtN.
template <int N>
SRT_DEV void tri_hitN(const Tri& g, V3 org, const V3* d, const float* b0, const float* b1, TriHit* h) {
  const V3 e1 = v3p(g.e1), e2 = v3p(g.e2);
  const V3 s = org - v3p(g.p0);
  const V3 sxe2 = cross(s, e2);
  const float nt = -1.0f * dot(sxe2, e1);
  float num[N][3], det[N], q[N][3], n2[N], nr[N];
#pragma unroll
  for (int r = 0; r < N; r++) {
    const V3 e1xd = cross(e1, d[r]);
    det[r] = dot(e1xd, e2);
    num[r][0] = -1.0f * dot(sxe2, d[r]);
    num[r][1] = dot(e1xd, s);
    num[r][2] = nt;
  }
  divNx3<N, true>(num, det, q);
#pragma unroll
  for (int r = 0; r < N; r++) n2[r] = norm2(d[r] * q[r][2]);
  const bool on_plane = nt == 0.0f;             // t = +-0 for every ray: the distances are exactly +0
  bool zero[N];
#pragma unroll
  for (int r = 0; r < N; r++) zero[r] = on_plane;
  sqrtN<N>(n2, zero, nr);
#pragma unroll
  for (int r = 0; r < N; r++) {
    h[r].u = q[r][0]; h[r].v = q[r][1]; h[r].t = q[r][2];
    const bool outside = (h[r].u < 0) || (h[r].v < 0) || ((1.0f - h[r].u - h[r].v) < 0) || (h[r].t < 0);
    h[r].dist = fabsf(nr[r]);
    const bool out_of_bounds = (h[r].dist < b0[r]) || (h[r].dist > b1[r]);
    h[r].hit = (det[r] != 0) && !outside && !out_of_bounds;
  }
}

// Triangle::hit of the (up to) four triangles of a BVH<Triangle> leaf for ONE ray: the twelve quotients and four
// distances go through divNx3 / sqrtN together (one range verdict, independent refinement chains).  Same arithmetic per
// triangle as tri_hit.
template <int N>
SRT_DEV void tri_hit_leafN(const Tri* g, const Ray& ray, TriHit* h) {
  float num[N][3], det[N], q[N][3], n2[N], nr[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    const V3 e1 = v3p(g[k].e1), e2 = v3p(g[k].e2);
    const V3 s = ray.o - v3p(g[k].p0);
    const V3 e1xd = cross(e1, ray.d);
    det[k] = dot(e1xd, e2);
    const V3 sxe2 = cross(s, e2);
    num[k][0] = -1.0f * dot(sxe2, ray.d);
    num[k][1] = dot(e1xd, s);
    num[k][2] = -1.0f * dot(sxe2, e1);
  }
  divNx3<N, false>(num, det, q);
  bool zero[N];
#pragma unroll
  for (int k = 0; k < N; k++) { n2[k] = norm2(ray.d * q[k][2]); zero[k] = false; }
  sqrtN<N>(n2, zero, nr);
#pragma unroll
  for (int k = 0; k < N; k++) {
    h[k].u = q[k][0]; h[k].v = q[k][1]; h[k].t = q[k][2];
    const bool outside = (h[k].u < 0) || (h[k].v < 0) || ((1.0f - h[k].u - h[k].v) < 0) || (h[k].t < 0);
    h[k].dist = fabsf(nr[k]);
    const bool out_of_bounds = (h[k].dist < ray.b0) || (h[k].dist > ray.b1);
    h[k].hit = (det[k] != 0) && !outside && !out_of_bounds;
  }
}
SRT_DEV void tri_hit_leaf4(const Tri* g, const Ray& ray, TriHit* h) { tri_hit_leafN<4>(g, ray, h); }

// Sphere::hit (student/shapes.cpp:17-80).  The reference's unqualified sqrt(delta) is the double overload,
// so the numerator sum and the quotient are fp64 before narrowing to t1/t2.  Straight-line form as above.
struct SphHit { bool hit; float t; };
SRT_DEV SphHit sphere_hit(float radius, const Ray& ray) {
  SphHit h;
  const float a = norm2(ray.d);
  const float b = 2.0f * dot(ray.o, ray.d);
  const float c = norm2(ray.o) - radius * radius;
  const float delta = b * b - 4.0f * a * c;
  // delta > 0: two roots
  const double m2od = (double)((-2.0f) * dot(ray.o, ray.d));
  const double sq = sqrt((double)delta);
  const double den = (double)(2.0f * norm2(ray.d));
  const float t1 = (float)((m2od + sq) / den);
  const float t2 = (float)((m2od - sq) / den);
  bool v1 = !(t1 < 0), v2 = !(t2 < 0);
  const float d1 = fabsf(norm(ray.d * t1));
  const float d2 = fabsf(norm(ray.d * t2));
  v1 = v1 && !(d1 < ray.b0 || d1 > ray.b1);
  v2 = v2 && !(d2 < ray.b0 || d2 > ray.b1);
  const float t_two = (v1 && v2) ? std_min(t1, t2) : (v1 ? t1 : t2);
  const bool hit_two = v1 || v2;
  // delta == 0: tangent, no validity checks in the reference
  const float t_one = ((-2.0f) * dot(ray.o, ray.d)) / (2.0f * norm2(ray.d));
  const bool two = delta > 0, one = delta == 0;
  h.hit = two ? hit_two : one;
  h.t = two ? (hit_two ? t_two : 0.0f) : (one ? t_one : 0.0f);
  return h;
}

// Mat4::rotate_to (lib/mat4.h:353-367): columns x, dir, z of the shading frame.
struct Frame { V3 x, y, z; };
SRT_DEV Frame rotate_to(V3 dir) {
  const float n = norm(dir);
  dir.x /= n; dir.y /= n; dir.z /= n;
  Frame f;
  if (fabsf(dir.y - 1.0f) < kEps) { f.x = v3(1, 0, 0); f.y = v3(0, 1, 0); f.z = v3(0, 0, 1); return f; }
  if (fabsf(dir.y + 1.0f) < kEps) { f.x = v3(1, 0, 0); f.y = v3(0, -1, 0); f.z = v3(0, 0, 1); return f; }
  f.x = unit(cross(dir, v3(0.0f, 1.0f, 0.0f)));
  f.z = unit(cross(f.x, dir));
  f.y = dir;
  return f;
}
// object_to_world.rotate(v): v.x*col0 + v.y*col1 + v.z*col2 + 0*col3 (col3 = (0,0,0,1))
SRT_DEV V3 frame_to_world(const Frame& f, V3 v) {
  return v3(((f.x.x * v.x + f.y.x * v.y) + f.z.x * v.z) + 0.0f * 0.0f,
            ((f.x.y * v.x + f.y.y * v.y) + f.z.y * v.z) + 0.0f * 0.0f,
            ((f.x.z * v.x + f.y.z * v.y) + f.z.z * v.z) + 0.0f * 0.0f);
}
// world_to_object = object_to_world.T(); its columns are (x.x, y.x, z.x, 0) ...
SRT_DEV V3 frame_to_local(const Frame& f, V3 v) {
  return v3(((f.x.x * v.x + f.x.y * v.y) + f.x.z * v.z) + 0.0f * 0.0f,
            ((f.y.x * v.x + f.y.y * v.y) + f.y.z * v.z) + 0.0f * 0.0f,
            ((f.z.x * v.x + f.z.y * v.y) + f.z.z * v.z) + 0.0f * 0.0f);
}

}  // namespace srt

#endif

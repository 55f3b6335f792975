/*
 * pt_oracle.c — CPU restatement of the Scotty3D path-tracer hot path
 * (Pathtracer::trace_pixel -> trace -> BVH::hit / Triangle::hit / Sphere::hit / BSDF::scatter).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path tracer is compared with; it is
 * called from tests/, from __graft_entry__.smoke() and from bench.py's cpu_baseline leg, and from
 * nowhere else.  The product (soft-rendering-toolsets_amd/) never links or loads it.
 *
 * Parity status: PINNED.  With math mode 0 (libm) tests/test_pt_oracle.py checks this restatement
 * against the reference's own sources compiled by oracle/Makefile (oracle/_ref/libref_pt.so, clang++,
 * direct-before-indirect evaluation order) — per-sample radiance, RNG draw counts, scene.hit results
 * and BVH node arrays, bit for bit — and against the fixtures under tests/golden/ produced from that
 * build.  Math mode 1 replaces the libm calls (cosf, sinf, pow) by the SRT-MATH v2 functions below —
 * a restatement of glibc's own sincosf algorithm, which the HIP kernel implements operation for
 * operation (OCML does not round like glibc).  tests/test_pt_oracle.py checks that the two modes agree.
 *
 * The reference is non-deterministic (util/rand.cpp seeds mt19937 from random_device); "a fixed RNG
 * seed" is realised by the SRT-RNG v1 generator, re-keyed per (pixel, sample), in the reference
 * build, here, and in the kernel alike.
 *
 * Citations are relative to /root/reference/Assignments/Scotty3D/src/.  All arithmetic is fp32
 * unless the reference itself widens (unqualified sqrt/pow resolve to the double overloads there).
 * Build with -ffp-contract=off and no -march flags (no FMA).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EPS_F 0.00001f                                   /* lib/mathlib.h:16 */
#define PI_F 3.14159265358979323846264338327950288f      /* lib/mathlib.h:17 */

typedef struct { float x, y, z; } v3;
typedef struct { float c[4][4]; } m4;                    /* c[col][row], Mat4::cols */
typedef struct { v3 mn, mx; } box3;
typedef struct { box3 b; uint32_t start, size, l, r; } node_t;       /* rays/bvh.h:35-39 */
typedef struct { int hit; float distance; v3 position, normal, origin; int material; } trace_t; /* rays/trace.h */
typedef struct { v3 point, dir; float b0, b1; uint32_t depth; } ray_t;                          /* lib/ray.h */
typedef struct { float r, g, b; } spec;

typedef struct { node_t* nodes; uint32_t nnodes, cap; uint32_t* prim; uint32_t nprims; } bvh_t;

enum { OBJ_MESH = 0, OBJ_SPHERE = 1 };
enum { MAT_LAMBERTIAN = 0, MAT_MIRROR = 1, MAT_GLASS = 2, MAT_DIFFUSE = 3, MAT_REFRACT = 4 };

typedef struct {
    int kind, has_trans, material, use_bvh;
    uint32_t id;
    m4 trans, itrans;
    v3 *pos, *nrm;          /* Tri_Mesh::verts */
    uint32_t nverts, ntri;
    uint32_t* tri;          /* 3 indices per triangle, input order */
    bvh_t bvh;              /* BVH<Triangle>; bvh.prim = triangle order after build */
    float radius;
    /* area-light copies only: Object::pdf's T = I*trans, iT = itrans*I and the transformed corners */
    m4 pdfT, pdfiT;
} object_t;

typedef struct { int type; spec a, b; float ior; } material_t;

/* Delta_Light (rays/light.h:57-96) */
struct delta_light { int type, has_trans; spec radiance; float angle_bounds[2]; m4 trans, itrans; };

typedef struct {
    uint64_t rays, box_tests, obj_entered, tri_tests, sphere_tests, tlas_nodes, blas_nodes, light_tri_tests;
} counters_t;

typedef struct {
    material_t* mats; uint32_t nmats;
    object_t* objs; uint32_t nobjs;
    object_t* lights; uint32_t nlights;
    struct delta_light* dlights; uint32_t ndlights;   /* Pathtracer::point_lights */
    int env_type; spec env_radiance;                  /* Pathtracer::env_light: 0 none, 1 Env_Sphere, 2 Env_Hemisphere, 3 Env_Map */
    float* env_map; uint32_t env_w, env_h;            /* Env_Map: HDR_Image pixels, index y * w + x */
    int use_bvh, committed;
    bvh_t tlas;             /* BVH<Object>; tlas.prim = object order after build */
    m4 iview; float vfov, ar;
    uint32_t w, h, max_depth;
    int math_mode;          /* 0 libm, 1 SRT-MATH v2 */
} scene_t;

typedef struct { uint64_t state, inc; uint32_t draws; } rng_t;
/* log: where the rays Pathtracer::log_ray receives go (srt_oracle_pt_epoch_rows_log): 10 floats per ray
 * {point, dir, t, pixel, sample, bounce} - NULL in every other entry point. */
typedef struct { float* buf; size_t cap, n; } raylog_t;
typedef struct { const scene_t* s; rng_t rng; counters_t cnt; raylog_t* log; } ctx_t;

/* ------------------------------------------------------------------------------------------------
 * SRT-RNG v1: replaces util/rand.cpp:13-25 (unit / integer / coin_flip).
 * ---------------------------------------------------------------------------------------------- */
static void rng_key(rng_t* r, uint64_t seed, uint32_t pixel, uint32_t sample) {
    const uint64_t k = ((uint64_t)pixel << 32) | sample;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (k + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    r->inc = (k << 1) | 1;
    r->state = z * 6364136223846793005ull + r->inc;
    r->draws = 0;
}
static uint32_t rng_next(rng_t* r) {
    const uint64_t old = r->state;
    r->state = old * 6364136223846793005ull + r->inc;
    const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    const uint32_t rot = (uint32_t)(old >> 59);
    r->draws++;
    return (xs >> rot) | (xs << ((32 - rot) & 31));
}
static float rng_unit(ctx_t* c) { return (float)(rng_next(&c->rng) >> 8) * (1.0f / 16777216.0f); }
static int rng_integer(ctx_t* c, int lo, int hi) {
    return lo + (int)(((uint64_t)rng_next(&c->rng) * (uint64_t)(uint32_t)(hi - lo)) >> 32);
}
static int rng_coin(ctx_t* c, float p) { return rng_unit(c) < p; }

/* ------------------------------------------------------------------------------------------------
 * SRT-MATH v2: cosf/sinf as glibc 2.35 computes them (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c,
 * sincosf.h, sincosf_data.c — the ARM optimized-routines sincosf by Szabolcs Nagy): the argument is
 * widened to fp64, reduced by pi/2 with one multiply-subtract (quadrant from a 2^24-scaled 2/pi), and a
 * degree-7 / degree-8 minimax polynomial is evaluated in fp64 and rounded once to fp32.  glibc is a
 * third-party dependency of the reference (not under /root/reference); the constants below are the
 * published ones and were checked against __sincosf_table in this image's libm.so.6
 * (Ubuntu GLIBC 2.35-0ubuntu3.11).  Evaluated without FMA, the restatement is bit-identical to
 * glibc's sinf/cosf (whose ifunc picks the FMA build on this host) for EVERY float in [0, 2pi] and
 * [-1, 1] - the only ranges this renderer uses: phi = 2*pi*xi and theta = cos(angle) - by the
 * exhaustive sweep srt_oracle_sweep_sincos_vs_libm; tests/test_pt_oracle.py repeats it on samples.
 * (Beyond 2pi the unfused form differs from the FMA build for 17 floats below 120.)
 * |x| >= 120 would need glibc's reduce_large; the renderer never gets there and NaN is returned.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sincos_tab;
static const sincos_tab SC_TAB[2] = {
    {0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static const double SC_SIGN[4] = {1.0, -1.0, -1.0, 1.0};
static uint32_t f_top12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
static float sc_poly(double x, double x2, const sincos_tab* p, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = p->s2 + x2 * p->s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * p->s1;
        return (float)(s + x7 * s1);
    } else {
        const double x4 = x2 * x2;
        const double c2 = p->c3 + x2 * p->c4;
        const double c1 = p->c0 + x2 * p->c1;
        const double x6 = x4 * x2;
        const double c = c1 + x4 * p->c2;
        return (float)(c + x6 * c2);
    }
}
static float srt_sincosf(float y, int want_cos) {
    double x = (double)y;
    const sincos_tab* p = &SC_TAB[0];
    if (f_top12(y) < f_top12(0x1.921FB6p-1f)) {               /* |y| < pi/4 */
        const double x2 = x * x;
        if (f_top12(y) < f_top12(0x1p-12f)) return want_cos ? 1.0f : y;
        return sc_poly(x, x2, p, want_cos);
    }
    if (f_top12(y) < f_top12(120.0f)) {
        const double r = x * p->hpi_inv;
        const int n = ((int32_t)r + 0x800000) >> 24;
        x = x - n * p->hpi;
        const double sgn = SC_SIGN[n & 3];
        if (n & 2) p = &SC_TAB[1];
        return sc_poly(x * sgn, x * x, p, n ^ want_cos);
    }
    return NAN;
}
static float srt_cosf(float x) { return srt_sincosf(x, 1); }
static float srt_sinf(float x) { return srt_sincosf(x, 0); }
/* SRT-MATH v2 atan2f: glibc 2.35 __ieee754_atan2f / __atanf (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c) */
static uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static float srt_atanf(float x) {
    static const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    static const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    static const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                                 9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                                 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    float w, s1, s2, z;
    int32_t hx = (int32_t)f_bits(x), ix = hx & 0x7fffffff, id;
    if (ix >= 0x4c000000) {
        if (ix > 0x7f800000) return x + x;
        return (hx > 0) ? (atanhi[3] + atanlo[3]) : (-atanhi[3] - atanlo[3]);
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x; w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}
static float srt_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    float z;
    int32_t hx = (int32_t)f_bits(x), ix = hx & 0x7fffffff, hy = (int32_t)f_bits(y), iy = hy & 0x7fffffff, k, m;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return srt_atanf(y);
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return (m < 2) ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000)
            return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = srt_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return bits_f(f_bits(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}
/* SRT-MATH v2 acosf: glibc 2.35 __ieee754_acosf (sysdeps/ieee754/flt-32/e_acosf.c) */
static float srt_acosf(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
                pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f,
                qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    float z, p, q, r, w, s, c, df;
    int32_t hx = (int32_t)f_bits(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return (hx > 0) ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = sqrtf(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    z = (one - x) * 0.5f;
    s = sqrtf(z);
    df = bits_f(f_bits(s) & 0xfffff000u);
    c = (z - df * df) / (s + df);
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    w = r * s + c;
    return 2.0f * (df + w);
}
static float m_acos(const scene_t* s, float x) { return s->math_mode ? srt_acosf(x) : acosf(x); }
static float m_atan2(const scene_t* s, float y, float x) { return s->math_mode ? srt_atan2f(y, x) : atan2f(y, x); }

static float m_cos(const scene_t* s, float x) { return s->math_mode ? srt_cosf(x) : cosf(x); }
static float m_sin(const scene_t* s, float x) { return s->math_mode ? srt_sinf(x) : sinf(x); }
/* (float)pow(x, 2) and (float)pow(x, 5) with x promoted to double (student/bsdf.cpp:17-21,47,150) */
static double m_pow2(const scene_t* s, float x) { return s->math_mode ? (double)x * (double)x : pow((double)x, 2.0); }
static double m_pow5(const scene_t* s, float x) {
    if (!s->math_mode) return pow((double)x, 5.0);
    const double d = (double)x, d2 = d * d, d4 = d2 * d2;
    return d4 * d;
}

/* ------------------------------------------------------------------------------------------------
 * lib/vec3.h, lib/spectrum.h
 * ---------------------------------------------------------------------------------------------- */
static v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 v_scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }       /* Vec3*float and float*Vec3 */
static v3 v_divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static float v_dot(v3 l, v3 r) { return l.x * r.x + l.y * r.y + l.z * r.z; }
static v3 v_cross(v3 l, v3 r) { return V(l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x); }
static float v_norm2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static float v_norm(v3 a) { return sqrtf(v_norm2(a)); }
static v3 v_unit(v3 a) { float n = v_norm(a); return V(a.x / n, a.y / n, a.z / n); }
static float min_f(float a, float b) { return (b < a) ? b : a; }                /* std::min */
static float max_f(float a, float b) { return (a < b) ? b : a; }                /* std::max */
static float v_get(v3 a, int i) { return i == 0 ? a.x : i == 1 ? a.y : a.z; }

static spec S(float r, float g, float b) { spec s = {r, g, b}; return s; }
static spec s_add(spec a, spec b) { return S(a.r + b.r, a.g + b.g, a.b + b.b); }
static spec s_sub(spec a, spec b) { return S(a.r - b.r, a.g - b.g, a.b - b.b); }
static spec s_mul(spec a, spec b) { return S(a.r * b.r, a.g * b.g, a.b * b.b); }
static spec s_scale(spec a, float s) { return S(a.r * s, a.g * s, a.b * s); }
static float s_luma(spec a) { return 0.2126f * a.r + 0.7152f * a.g + 0.0722f * a.b; }   /* spectrum.h:111 */
static int s_valid(spec a) { return isfinite(a.r) && isfinite(a.g) && isfinite(a.b); }  /* spectrum.h:115 */

/* ------------------------------------------------------------------------------------------------
 * lib/mat4.h
 * ---------------------------------------------------------------------------------------------- */
static m4 m_identity(void) {
    m4 r; memset(&r, 0, sizeof r);
    r.c[0][0] = r.c[1][1] = r.c[2][2] = r.c[3][3] = 1.0f;
    return r;
}
/* Mat4::operator*(Vec3): v0*col0 + v1*col1 + v2*col2 + 1*col3, then project (mat4.h:125-131) */
static v3 m_point(const m4* m, v3 v) {
    float o[4];
    for (int j = 0; j < 4; j++) o[j] = ((m->c[0][j] * v.x + m->c[1][j] * v.y) + m->c[2][j] * v.z) + m->c[3][j] * 1.0f;
    return V(o[0] / o[3], o[1] / o[3], o[2] / o[3]);
}
/* Mat4::rotate: the same sum with w = 0 (the 0*col3 term is kept: it can turn -0 into +0) */
static v3 m_rotate(const m4* m, v3 v) {
    float o[3];
    for (int j = 0; j < 3; j++) o[j] = ((m->c[0][j] * v.x + m->c[1][j] * v.y) + m->c[2][j] * v.z) + m->c[3][j] * 0.0f;
    return V(o[0], o[1], o[2]);
}
static m4 m_transpose(const m4* m) {
    m4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.c[i][j] = m->c[j][i];
    return r;
}
/* Mat4::operator*(const Mat4& m): ret[i][j] = sum_k m[i][k] * cols[k][j], accumulated from 0.0f (mat4.h:110-121) */
static m4 m_mul(const m4* self, const m4* m) {
    m4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        float acc = 0.0f;
        for (int k = 0; k < 4; k++) acc += m->c[i][k] * self->c[k][j];
        r.c[i][j] = acc;
    }
    return r;
}
/* Term order of Mat4::inverse (mat4.h:296-343) and Mat4::det (mat4.h:206-231); each digit pair is
 * (col,row).  The order of terms and factors fixes the rounding, so it is kept as data. */
static const char* INV_TERMS[16] = {
    "+122331-132231+132132-112332-122133+112233", "+032231-022331-032132+012332+022133-012233",
    "+021331-031231+031132-011332-021133+011233", "+031221-021321-031122+011322+021123-011223",
    "+132230-122330-132032+102332+122033-102233", "+022330-032230+032032-002332-022033+002233",
    "+031230-021330-031032+001332+021033-001233", "+021320-031220+031022-001322-021023+001223",
    "+112330-132130+132031-102331-112033+102133", "+032130-012330-032031+002331+012033-002133",
    "+011330-031130+031031-001331-011033+001133", "+031120-011320-031021+001321+011023-001123",
    "+122130-112230-122031+102231+112032-102132", "+012230-022130+022031-002231-012032+002132",
    "+021130-011230-021031+001231+011032-001132", "+011220-021120+021021-001221-011022+001122"};
static const char* DET_TERMS =
    "+03122130-02132130-03112230+01132230+02112330-01122330-03122031+02132031+03102231-00132231-02102331"
    "+00122331+03112032-01132032-03102132+00132132+01102332-00112332-02112033+01122033+02102133-00122133"
    "-01102233+00112233";
static float m_terms(const m4* m, const char* t, int nfac) {
    float acc = 0.0f;
    int first = 1;
    while (*t) {
        const int neg = (*t++ == '-');
        float p = m->c[t[0] - '0'][t[1] - '0'];
        for (int f = 1; f < nfac; f++) p = p * m->c[t[2 * f] - '0'][t[2 * f + 1] - '0'];
        t += 2 * nfac;
        if (first) { acc = p; first = 0; }
        else acc = neg ? acc - p : acc + p;
    }
    return acc;
}
static m4 m_inverse(const m4* m) {
    m4 r;
    for (int e = 0; e < 16; e++) r.c[e / 4][e % 4] = m_terms(m, INV_TERMS[e], 3);
    const float det = m_terms(m, DET_TERMS, 4);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.c[i][j] /= det;
    return r;
}
/* Mat4::rotate_to (mat4.h:353-367) */
static m4 m_rotate_to(v3 dir) {
    float n = v_norm(dir);
    dir.x /= n; dir.y /= n; dir.z /= n;
    m4 r = m_identity();
    if (fabsf(dir.y - 1.0f) < EPS_F) return r;
    if (fabsf(dir.y + 1.0f) < EPS_F) { r.c[1][1] = -1.0f; return r; }
    v3 x = v_unit(v_cross(dir, V(0.0f, 1.0f, 0.0f)));
    v3 z = v_unit(v_cross(x, dir));
    r.c[0][0] = x.x; r.c[0][1] = x.y; r.c[0][2] = x.z; r.c[0][3] = 0.0f;
    r.c[1][0] = dir.x; r.c[1][1] = dir.y; r.c[1][2] = dir.z; r.c[1][3] = 0.0f;
    r.c[2][0] = z.x; r.c[2][1] = z.y; r.c[2][2] = z.z; r.c[2][3] = 0.0f;
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * lib/bbox.h, student/bbox.cpp
 * ---------------------------------------------------------------------------------------------- */
static box3 box_empty(void) { box3 b = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}}; return b; }
static void box_enclose_pt(box3* b, v3 p) {
    b->mn = V(min_f(b->mn.x, p.x), min_f(b->mn.y, p.y), min_f(b->mn.z, p.z));
    b->mx = V(max_f(b->mx.x, p.x), max_f(b->mx.y, p.y), max_f(b->mx.z, p.z));
}
static void box_enclose(box3* b, box3 o) {
    b->mn = V(min_f(b->mn.x, o.mn.x), min_f(b->mn.y, o.mn.y), min_f(b->mn.z, o.mn.z));
    b->mx = V(max_f(b->mx.x, o.mx.x), max_f(b->mx.y, o.mx.y), max_f(b->mx.z, o.mx.z));
}
static v3 box_center(box3 b) { return v_scale(v_add(b.mn, b.mx), 0.5f); }
static float box_area(box3 b) {                                               /* bbox.h:50-54 */
    if (b.mn.x > b.mx.x || b.mn.y > b.mx.y || b.mn.z > b.mx.z) return 0.0f;
    v3 e = v_sub(b.mx, b.mn);
    return 2.0f * (e.x * e.z + e.x * e.y + e.y * e.z);
}
static box3 box_transform(box3 b, const m4* t) {                              /* bbox.h:57-73 */
    float amin[3] = {b.mn.x, b.mn.y, b.mn.z}, amax[3] = {b.mx.x, b.mx.y, b.mx.z};
    float mn[3], mx[3];
    for (int i = 0; i < 3; i++) mn[i] = mx[i] = t->c[3][i];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        float a = t->c[j][i] * amin[j], bb = t->c[j][i] * amax[j];
        if (a < bb) { mn[i] += a; mx[i] += bb; } else { mn[i] += bb; mx[i] += a; }
    }
    box3 r = {{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}};
    return r;
}
/* BBox::hit, student/bbox.cpp:5-62.  Returns 1 whenever the LINE meets the slabs; times only narrows. */
static int box_hit(ctx_t* c, box3 b, const ray_t* ray, float* tx, float* ty) {
    c->cnt.box_tests++;
    const float ix = 1.0f / ray->dir.x, iy = 1.0f / ray->dir.y, iz = 1.0f / ray->dir.z;
    const int sx = ix < 0, sy = iy < 0, sz = iz < 0;
    float tmin = ((sx ? b.mx.x : b.mn.x) - ray->point.x) * ix;
    float tmax = ((sx ? b.mn.x : b.mx.x) - ray->point.x) * ix;
    const float tymin = ((sy ? b.mx.y : b.mn.y) - ray->point.y) * iy;
    const float tymax = ((sy ? b.mn.y : b.mx.y) - ray->point.y) * iy;
    if ((tmin > tymax) || (tymin > tmax)) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    const float tzmin = ((sz ? b.mx.z : b.mn.z) - ray->point.z) * iz;
    const float tzmax = ((sz ? b.mn.z : b.mx.z) - ray->point.z) * iz;
    if ((tmin > tzmax) || (tzmin > tmax)) return 0;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    if (tmin >= *tx && tmin <= *ty) *tx = tmin;
    if (tmax >= *tx && tmax <= *ty) *ty = tmax;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * rays/trace.h
 * ---------------------------------------------------------------------------------------------- */
static trace_t trace_none(void) { trace_t t; memset(&t, 0, sizeof t); return t; }
static trace_t trace_min(trace_t l, trace_t r) {                              /* trace.h:15-23 */
    if (l.hit && r.hit) { if (l.distance < r.distance) return l; return r; }
    if (l.hit) return l;
    if (r.hit) return r;
    return trace_none();
}
static void trace_transform(trace_t* t, const m4* tr, const m4* norm) {       /* trace.h:25-30 */
    t->position = m_point(tr, t->position);
    t->origin = m_point(tr, t->origin);
    t->normal = v_unit(m_rotate(norm, t->normal));
    t->distance = v_norm(v_sub(t->position, t->origin));
}
static void ray_transform(ray_t* r, const m4* tr) {                           /* lib/ray.h:31-37 */
    r->point = m_point(tr, r->point);
    r->dir = m_rotate(tr, r->dir);
    float d = v_norm(r->dir);
    r->b0 *= d; r->b1 *= d;
    r->dir = v_divs(r->dir, d);
}
static ray_t ray_make(v3 point, v3 dir, float b0, float b1, uint32_t depth) { /* lib/ray.h:16-19 */
    ray_t r; r.point = point; r.dir = v_unit(dir); r.b0 = b0; r.b1 = b1; r.depth = depth; return r;
}
static v3 ray_at(const ray_t* r, float t) { return v_add(r->point, v_scale(r->dir, t)); }

/* ------------------------------------------------------------------------------------------------
 * student/tri_mesh.cpp, student/shapes.cpp
 * ---------------------------------------------------------------------------------------------- */
static box3 tri_bbox(const object_t* o, uint32_t t) {                         /* tri_mesh.cpp:7-30 */
    v3 p0 = o->pos[o->tri[3 * t]], p1 = o->pos[o->tri[3 * t + 1]], p2 = o->pos[o->tri[3 * t + 2]];
    float mnx = min_f(min_f(p0.x, p1.x), p2.x), mxx = max_f(max_f(p0.x, p1.x), p2.x);
    float mny = min_f(min_f(p0.y, p1.y), p2.y), mxy = max_f(max_f(p0.y, p1.y), p2.y);
    float mnz = min_f(min_f(p0.z, p1.z), p2.z), mxz = max_f(max_f(p0.z, p1.z), p2.z);
    mxx = (mnx >= mxx) ? (mnx + 1.0f) : mxx;
    mxy = (mny >= mxy) ? (mny + 1.0f) : mxy;
    mxz = (mnz >= mxz) ? (mnz + 1.0f) : mxz;
    box3 b = {{mnx, mny, mnz}, {mxx, mxy, mxz}};
    return b;
}
static trace_t tri_hit(ctx_t* c, const object_t* o, uint32_t t, const ray_t* ray) {   /* tri_mesh.cpp:32-111 */
    c->cnt.tri_tests++;
    const uint32_t i0 = o->tri[3 * t], i1 = o->tri[3 * t + 1], i2 = o->tri[3 * t + 2];
    const v3 p0 = o->pos[i0], p1 = o->pos[i1], p2 = o->pos[i2];
    int result = 1;
    v3 uvt = V(0, 0, 0);
    float distance = FLT_MIN;
    const v3 e1 = v_sub(p1, p0), e2 = v_sub(p2, p0), s = v_sub(ray->point, p0);
    const float det = v_dot(v_cross(e1, ray->dir), e2);
    if (det != 0) {
        v3 num = V(-1.0f * v_dot(v_cross(s, e2), ray->dir), v_dot(v_cross(e1, ray->dir), s),
                   -1.0f * v_dot(v_cross(s, e2), e1));
        uvt = v_divs(num, det);
        if (uvt.x < 0 || uvt.y < 0 || (1.0f - uvt.x - uvt.y) < 0 || uvt.z < 0) result = 0;
        distance = fabsf(v_norm(v_scale(ray->dir, uvt.z)));
        if (distance < ray->b0 || distance > ray->b1) result = 0;
    } else {
        result = 0;
    }
    trace_t ret = trace_none();
    ret.origin = ray->point;
    ret.hit = result;
    if (result) {
        ret.distance = distance;
        ret.position = ray_at(ray, uvt.z);
        /* weights as the fork wrote them: u*n0 + v*n1 + (1-u-v)*n2 (tri_mesh.cpp:104-106) */
        ret.normal = v_add(v_add(v_scale(o->nrm[i0], uvt.x), v_scale(o->nrm[i1], uvt.y)),
                           v_scale(o->nrm[i2], 1.0f - uvt.x - uvt.y));
    }
    return ret;
}
static trace_t sphere_hit(ctx_t* c, float radius, const ray_t* ray) {         /* shapes.cpp:17-80 */
    c->cnt.sphere_tests++;
    int result = 1;
    float t = 0;
    const float a = v_norm2(ray->dir);
    const float b = 2.0f * v_dot(ray->point, ray->dir);
    const float cc = v_norm2(ray->point) - radius * radius;
    const float delta = b * b - 4.0f * a * cc;
    if (delta > 0) {
        /* the reference's unqualified sqrt(delta) is the double overload: sum and quotient are fp64 */
        const float t1 = (float)(((double)((-2.0f) * v_dot(ray->point, ray->dir)) + sqrt((double)delta)) /
                                 (double)(2.0f * v_norm2(ray->dir)));
        const float t2 = (float)(((double)((-2.0f) * v_dot(ray->point, ray->dir)) - sqrt((double)delta)) /
                                 (double)(2.0f * v_norm2(ray->dir)));
        int v1 = (t1 < 0) ? 0 : 1, v2 = (t2 < 0) ? 0 : 1;
        const float d1 = fabsf(v_norm(v_scale(ray->dir, t1)));
        const float d2 = fabsf(v_norm(v_scale(ray->dir, t2)));
        if (d1 < ray->b0 || d1 > ray->b1) v1 = 0;
        if (d2 < ray->b0 || d2 > ray->b1) v2 = 0;
        if (v1 && v2) t = min_f(t1, t2);
        else if (!v1 && !v2) result = 0;
        else t = v1 ? t1 : t2;
    } else if (delta == 0) {
        t = ((-2.0f) * v_dot(ray->point, ray->dir)) / (2.0f * v_norm2(ray->dir));
    } else {
        result = 0;
    }
    trace_t ret = trace_none();
    ret.origin = ray->point;
    ret.hit = result;
    if (result) {
        ret.distance = fabsf(v_norm(v_sub(ray_at(ray, t), ray->point)));
        ret.position = ray_at(ray, t);
        ret.normal = v_sub(ray_at(ray, t), V(0.0f, 0.0f, 0.0f));
    }
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * BVH<Primitive>::build, student/bvh.inl:35-163 (level order, <= 9 SAH planes per axis,
 * libstdc++ std::partition permutation)
 * ---------------------------------------------------------------------------------------------- */
static uint32_t part_by_center(uint32_t* prim, const box3* pb, uint32_t first, uint32_t last, int axis, float line) {
    /* std::partition for bidirectional iterators (bits/stl_algo.h __partition) */
    for (;;) {
        for (;;) {
            if (first == last) return first;
            if (v_get(box_center(pb[prim[first]]), axis) < line) ++first; else break;
        }
        --last;
        for (;;) {
            if (first == last) return first;
            if (!(v_get(box_center(pb[prim[last]]), axis) < line)) --last; else break;
        }
        uint32_t tmp = prim[first]; prim[first] = prim[last]; prim[last] = tmp;
        ++first;
    }
}
static uint32_t bvh_new_node(bvh_t* b, box3 bx, uint32_t start, uint32_t size) {
    if (b->nnodes == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 64;
        b->nodes = (node_t*)realloc(b->nodes, b->cap * sizeof(node_t));
    }
    node_t n; n.b = bx; n.start = start; n.size = size; n.l = 0; n.r = 0;
    b->nodes[b->nnodes] = n;
    return b->nnodes++;
}
/* pb[i] = bbox of primitive i (input order).  Returns 0, or -1 if the build does not terminate
 * (the reference loops forever when a split leaves a child as large as its parent). */
static int bvh_build(bvh_t* b, const box3* pb, uint32_t n, uint32_t max_leaf) {
    typedef struct { box3 bl, br; int lc, rc; float line; } part_t;
    b->nodes = NULL; b->nnodes = 0; b->cap = 0; b->nprims = n;
    b->prim = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) b->prim[i] = i;
    box3 all = box_empty();
    for (uint32_t i = 0; i < n; i++) box_enclose(&all, pb[i]);
    bvh_new_node(b, all, 0, n);
    const uint32_t node_limit = 8u * n + 64u;
    for (uint32_t cur = 0; cur < b->nnodes; cur++) {
        if (b->nodes[cur].size <= max_leaf) continue;
        if (b->nnodes > node_limit) return -1;
        const node_t nd = b->nodes[cur];
        const int start = (int)nd.start, end = (int)nd.start + (int)nd.size;
        float best_xyz[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        part_t best[3];
        for (int axis = 0; axis < 3; axis++) {
            part_t ba; ba.bl = box_empty(); ba.br = box_empty(); ba.lc = 0; ba.rc = 0; ba.line = 0;
            const float lo = v_get(nd.b.mn, axis), hi = v_get(nd.b.mx, axis);
            const float interval = (hi - lo) / (float)10;
            for (float median = lo + interval; median < hi; median += interval) {
                const uint32_t mid = part_by_center(b->prim, pb, (uint32_t)start, (uint32_t)end, axis, median);
                part_t p; p.bl = box_empty(); p.br = box_empty(); p.lc = 0; p.rc = 0; p.line = median;
                for (int i = start; i < end; i++) {
                    if (i >= (int)mid) { box_enclose(&p.br, pb[b->prim[i]]); p.rc++; }
                    else { box_enclose(&p.bl, pb[b->prim[i]]); p.lc++; }
                }
                const float cost = box_area(p.bl) / box_area(nd.b) * (float)p.lc +
                                   box_area(p.br) / box_area(nd.b) * (float)p.rc + 1.0f;
                if (cost < best_xyz[axis]) { best_xyz[axis] = cost; ba = p; }
            }
            best[axis] = ba;
        }
        const float best_cost = min_f(best_xyz[0], min_f(best_xyz[1], best_xyz[2]));
        const int ax = (best_cost == best_xyz[0]) ? 0 : ((best_cost == best_xyz[1]) ? 1 : 2);
        const uint32_t l = b->nnodes;
        part_by_center(b->prim, pb, (uint32_t)start, (uint32_t)end, ax, best[ax].line);
        bvh_new_node(b, best[ax].bl, nd.start, (uint32_t)best[ax].lc);
        bvh_new_node(b, best[ax].br, nd.start + (uint32_t)best[ax].lc, (uint32_t)best[ax].rc);
        b->nodes[cur].l = l;
        b->nodes[cur].r = l + 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Object::hit (rays/object.h:57-65), BVH<>::hit / find_closest_hit (student/bvh.inl:166-276),
 * List<>::hit (rays/list.h:27-34)
 * ---------------------------------------------------------------------------------------------- */
static trace_t blas_fch(ctx_t* c, const object_t* o, const ray_t* ray, uint32_t n, float* tx, float* ty);
static trace_t object_hit(ctx_t* c, const object_t* o, ray_t ray) {
    if (o->has_trans) { c->cnt.obj_entered++; ray_transform(&ray, &o->itrans); }
    trace_t ret;
    if (o->kind == OBJ_SPHERE) {
        ret = sphere_hit(c, o->radius, &ray);
    } else if (o->use_bvh) {
        ret = trace_none();
        if (o->bvh.nnodes) {
            const float dn = v_norm(ray.dir);
            float tx = ray.b0 / dn, ty = ray.b1 / dn;
            ret = blas_fch(c, o, &ray, 0, &tx, &ty);
        }
    } else {
        ret = trace_none();
        for (uint32_t t = 0; t < o->ntri; t++) ret = trace_min(ret, tri_hit(c, o, t, &ray));
    }
    if (ret.hit) {
        if (o->material != -1) ret.material = o->material;
        if (o->has_trans) { m4 nt = m_transpose(&o->itrans); trace_transform(&ret, &o->trans, &nt); }
    }
    return ret;
}
#define FCH_BODY(LEAF_HIT, RECURSE, NODES)                                                              \
    trace_t ret = trace_none();                                                                         \
    const node_t* nd = &(NODES)[n];                                                                     \
    if (nd->l == nd->r) {                                                                               \
        for (uint32_t i = nd->start; i < nd->start + nd->size; i++) ret = trace_min(ret, LEAF_HIT);     \
        *tx = ret.distance;                                                                             \
        return ret;                                                                                     \
    }                                                                                                   \
    float t1x = *tx, t1y = *ty, t2x = *tx, t2y = *ty;                                                   \
    const int hl = box_hit(c, (NODES)[nd->l].b, ray, &t1x, &t1y);                                       \
    const int hr = box_hit(c, (NODES)[nd->r].b, ray, &t2x, &t2y);                                       \
    if (!hl && !hr) return ret;                                                                         \
    uint32_t closer, second;                                                                            \
    int hitboth = 0;                                                                                    \
    float cx = ray->b0, cy = ray->b1, fx = ray->b0, fy = ray->b1;                                       \
    if (hl && hr) {                                                                                     \
        hitboth = 1;                                                                                    \
        if (t1x < t2x) { closer = nd->l; second = nd->r; cx = t1x; cy = t1y; fx = t2x; fy = t2y; }      \
        else { closer = nd->r; second = nd->l; cx = t2x; cy = t2y; fx = t1x; fy = t1y; }                \
    } else if (hl) { closer = nd->l; second = nd->r; cx = t1x; cy = t1y; }                              \
    else { closer = nd->r; second = nd->l; cx = t2x; cy = t2y; }                                        \
    ret = RECURSE(closer, &cx, &cy);                                                                    \
    if (fx < ret.distance || (!ret.hit && hitboth)) {                                                   \
        trace_t h2 = RECURSE(second, &fx, &fy);                                                         \
        ret = trace_min(ret, h2);                                                                       \
    }                                                                                                   \
    return ret;

static trace_t blas_fch(ctx_t* c, const object_t* o, const ray_t* ray, uint32_t n, float* tx, float* ty) {
    c->cnt.blas_nodes++;
#define BLAS_REC(N, X, Y) blas_fch(c, o, ray, (N), (X), (Y))
    FCH_BODY(tri_hit(c, o, o->bvh.prim[i], ray), BLAS_REC, o->bvh.nodes)
#undef BLAS_REC
}
static trace_t tlas_fch(ctx_t* c, const ray_t* ray, uint32_t n, float* tx, float* ty) {
    const scene_t* s = c->s;
    c->cnt.tlas_nodes++;
#define TLAS_REC(N, X, Y) tlas_fch(c, ray, (N), (X), (Y))
    FCH_BODY(object_hit(c, &s->objs[s->tlas.prim[i]], *ray), TLAS_REC, s->tlas.nodes)
#undef TLAS_REC
}
/* scene.hit(ray): the scene Object has no transform and material -1 (rays/pathtracer.cpp:169-175) */
static trace_t scene_hit(ctx_t* c, const ray_t* ray) {
    const scene_t* s = c->s;
    c->cnt.rays++;
    trace_t ret = trace_none();
    if (s->use_bvh) {
        if (!s->tlas.nnodes) return ret;
        const float dn = v_norm(ray->dir);
        float tx = ray->b0 / dn, ty = ray->b1 / dn;
        return tlas_fch(c, ray, 0, &tx, &ty);
    }
    for (uint32_t i = 0; i < s->nobjs; i++) ret = trace_min(ret, object_hit(c, &s->objs[i], *ray));
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * student/bsdf.cpp, student/samplers.cpp
 * ---------------------------------------------------------------------------------------------- */
typedef struct { spec attenuation; v3 direction; } scatter_t;
static v3 bsdf_reflect(v3 d) { return V((-1.0f) * d.x, d.y, (-1.0f) * d.z); }             /* bsdf.cpp:7-15 */
static float schlick(const scene_t* s, float cosine, float ior) {                         /* bsdf.cpp:17-21 */
    float r0 = (1 - ior) / (1 + ior);
    r0 = r0 * r0;
    return r0 + (1 - r0) * (float)m_pow5(s, 1 - cosine);
}
static v3 bsdf_refract(const scene_t* s, v3 out, float ior, int* internal) {              /* bsdf.cpp:23-64 */
    const float cos_i = out.y;
    float ni, nt;
    if (cos_i > 0) { nt = ior; ni = 1.0f; } else { nt = 1.0f; ni = ior; }
    const float ratio = ni / nt;
    const float cos_t_sq = 1.0f - (float)m_pow2(s, ratio) * (1.0f - (float)m_pow2(s, cos_i));
    *internal = (cos_t_sq < 0);
    if (*internal) return bsdf_reflect(out);
    /* unqualified sqrt -> double; the product with -1.0f is double too, then narrowed */
    const float cos_t = (cos_i >= 0) ? (float)((double)(-1.0f) * sqrt((double)cos_t_sq)) : (float)sqrt((double)cos_t_sq);
    v3 in;
    in.x = (-1.0f) * out.x * ratio;
    in.y = cos_t;
    in.z = (-1.0f) * out.z * ratio;
    return in;
}
static spec lambert_evaluate(const scene_t* s, const material_t* m, v3 out) {             /* bsdf.cpp:92-105 */
    v3 u = v_unit(out);
    float theta = v_dot(u, V(0.0f, 1.0f, 0.0f));
    return s_scale(m->a, m_cos(s, theta));
}
static float lambert_pdf(const scene_t* s, v3 out) {                                      /* bsdf.cpp:107-117 */
    float theta = v_dot(out, V(0.0f, 1.0f, 0.0f));
    float ct = m_cos(s, theta);
    ct = min_f(max_f(ct, 0.0f), 1.0f);
    return ct / PI_F;
}
static v3 cosine_hemisphere(ctx_t* c) {                                                   /* samplers.cpp:166-177 */
    float phi = rng_unit(c) * 2.0f * PI_F;
    float cos_t = sqrtf(rng_unit(c));
    float sin_t = sqrtf(1 - cos_t * cos_t);
    float x = m_cos(c->s, phi) * sin_t;
    float z = m_sin(c->s, phi) * sin_t;
    return V(x, cos_t, z);
}
static scatter_t bsdf_scatter(ctx_t* c, const material_t* m, v3 out) {
    const scene_t* s = c->s;
    scatter_t r;
    switch (m->type) {
    case MAT_LAMBERTIAN:                                                                  /* bsdf.cpp:69-87 */
        r.direction = cosine_hemisphere(c);
        r.attenuation = lambert_evaluate(s, m, out);
        break;
    case MAT_MIRROR:                                                                      /* bsdf.cpp:119-126 */
        r.direction = bsdf_reflect(out);
        r.attenuation = m->a;
        break;
    case MAT_GLASS: {                                                                     /* bsdf.cpp:128-154 */
        int internal = 0;
        v3 refr = bsdf_refract(s, out, m->ior, &internal);
        float fresnel = schlick(s, fabsf(out.y), m->ior);
        int flip = rng_coin(c, fresnel);          /* always drawn: coin_flip is the left operand of || */
        if (flip || internal) {
            r.direction = bsdf_reflect(out);
            r.attenuation = m->b;
        } else {
            r.direction = refr;
            float ratio = (out.y > 0) ? (1.0f / m->ior) : m->ior;
            r.attenuation = s_scale(m->a, (float)m_pow2(s, ratio));
        }
    } break;
    default:                                                                              /* BSDF_Refract stub, bsdf.cpp:156-166 */
        r.direction = V(0, 0, 0);
        r.attenuation = S(0, 0, 0);
        break;
    }
    return r;
}
static int mat_discrete(const material_t* m) { return m->type == MAT_MIRROR || m->type == MAT_GLASS || m->type == MAT_REFRACT; }
static int mat_sided(const material_t* m) { return m->type == MAT_GLASS || m->type == MAT_REFRACT; }
static spec mat_emissive(const material_t* m) { return m->type == MAT_DIFFUSE ? m->a : S(0, 0, 0); }

/* ------------------------------------------------------------------------------------------------
 * Area lights: Pathtracer::sample_area_lights / area_lights_pdf (rays/pathtracer.cpp:301-325),
 * List::sample/pdf (rays/list.h:43-55), Object::sample/pdf (rays/object.h:77-101),
 * Triangle::sample/pdf (student/tri_mesh.cpp:117-143), Samplers::Triangle (samplers.cpp:143-149)
 * ---------------------------------------------------------------------------------------------- */
static float std_clamp(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }   /* std::clamp */
static spec lerp_spec(float ratio, spec a, spec b) {           /* lerpSpectrum, student/env_light.cpp:33-35 */
    float om = 1 - ratio;
    return S(om * a.r + ratio * b.r, om * a.g + ratio * b.g, om * a.b + ratio * b.b);
}
/* HDR_Image::at(x, y); float -> int conversions that are undefined / assert in the reference are clamped into the image */
static spec env_texel(const scene_t* s, float x, float y) {
    int xi = (x >= 0.0f && x < 2147483648.0f) ? (int)x : 0, yi = (y >= 0.0f && y < 2147483648.0f) ? (int)y : 0;
    if (xi >= (int)s->env_w) xi = (int)s->env_w - 1;
    if (yi >= (int)s->env_h) yi = (int)s->env_h - 1;
    const float* p = s->env_map + 3 * ((size_t)yi * s->env_w + (size_t)xi);
    return S(p[0], p[1], p[2]);
}
/* Env_Map::evaluate, student/env_light.cpp:37-93 */
static spec env_map_evaluate(const scene_t* s, v3 dir) {
    float r = v_norm(dir);
    float theta = PI_F - m_acos(s, dir.y / r);
    float phi = m_atan2(s, dir.z, dir.x);
    if (phi < 0) phi = phi + 2.f * PI_F;
    theta = std_clamp(theta / PI_F, 0.f, 1.f);
    phi = std_clamp(phi / (2.0f * PI_F), 0.f, 1.f);
    float h = (float)s->env_h, w = (float)s->env_w;
    float u = phi * w, v = theta * h;
    float u0 = floorf(u), v0 = floorf(v), u1, v1, t;
    if (u - u0 < 0.5f) { u1 = u0 - 1; t = u0; u0 = u1; u1 = t; } else { u1 = u0 + 1.f; }
    if (v - v0 < 0.5f) { v1 = v0 - 1; t = v0; v0 = v1; v1 = t; } else { v1 = v0 + 1.f; }
    u0 = min_f(max_f(u0, 0.f), w - 1.f); u1 = min_f(max_f(u1, 0.f), w - 1.f);
    v0 = min_f(max_f(v0, 0.f), h - 1.f); v1 = min_f(max_f(v1, 0.f), h - 1.f);
    float ru = min_f(max_f(u - u0 - 0.5f, 0.f), 1.f), rv = min_f(max_f(v - v0 - 0.5f, 0.f), 1.f);
    spec h1 = lerp_spec(ru, env_texel(s, u0, v0), env_texel(s, u1, v0));
    spec h2 = lerp_spec(ru, env_texel(s, u0, v1), env_texel(s, u1, v1));
    return lerp_spec(rv, h1, h2);
}
/* Env_Sphere / Env_Hemisphere / Env_Map::evaluate, student/env_light.cpp:37-118 */
static spec env_evaluate(const scene_t* s, v3 dir) {
    if (s->env_type == 3) return env_map_evaluate(s, dir);
    if (s->env_type == 2) return (dir.y > 0.0f) ? s->env_radiance : S(0, 0, 0);
    return s->env_radiance;
}
/* Env_*::sample = Samplers::Hemisphere::Uniform::sample, student/samplers.cpp:151-164 (Sphere::Uniform returns the same
 * upper-hemisphere sample, :17-26) */
static v3 env_sample(ctx_t* c) {
    const scene_t* s = c->s;
    float xi1 = rng_unit(c);
    float xi2 = rng_unit(c);
    float theta = m_acos(s, xi1);
    float phi = 2.0f * PI_F * xi2;
    float xs = m_sin(s, theta) * m_cos(s, phi);
    float ys = m_cos(s, theta);
    float zs = m_sin(s, theta) * m_sin(s, phi);
    return V(xs, ys, zs);
}
static v3 sample_area_lights(ctx_t* c, v3 from) {                /* rays/pathtracer.cpp:301-311 */
    const scene_t* s = c->s;
    if (s->env_type) {
        if (!s->nlights) return env_sample(c);
        if (rng_coin(c, 0.5f)) return env_sample(c);
    }
    if (!s->nlights) return V(0, 0, 0);
    const object_t* o = &s->lights[rng_integer(c, 0, (int)s->nlights)];
    if (o->has_trans) from = m_point(&o->itrans, from);
    const uint32_t t = (uint32_t)rng_integer(c, 0, (int)o->ntri);
    const v3 v0 = o->pos[o->tri[3 * t]], v1 = o->pos[o->tri[3 * t + 1]], v2 = o->pos[o->tri[3 * t + 2]];
    const float u = sqrtf(rng_unit(c));
    const float v = rng_unit(c);
    const float a = u * (1.0f - v);
    const float b = u * v;
    const v3 pos = v_add(v_add(v_scale(v0, a), v_scale(v1, b)), v_scale(v2, 1.0f - a - b));
    v3 dir = v_unit(v_sub(pos, from));
    if (o->has_trans) dir = v_unit(m_rotate(&o->trans, dir));
    return dir;
}
static float area_lights_pdf(ctx_t* c, v3 from, v3 dir) {
    const scene_t* s = c->s;
    int n = 0;
    float pdf = 0.0f;
    if (s->nlights) {
        const ray_t wray = ray_make(from, dir, 0.0f, FLT_MAX, 0);
        float ret = 0.0f;
        for (uint32_t li = 0; li < s->nlights; li++) {
            const object_t* o = &s->lights[li];
            float sum = 0.0f;
            const m4 iTt = m_transpose(&o->pdfiT);
            for (uint32_t t = 0; t < o->ntri; t++) {
                ray_t tray = wray;
                ray_transform(&tray, &o->pdfiT);
                c->cnt.light_tri_tests++;
                trace_t tr = tri_hit(c, o, t, &tray);
                c->cnt.tri_tests--;
                float p = 0.0f;
                if (tr.hit) {
                    trace_transform(&tr, &o->pdfT, &iTt);
                    const v3 w0 = m_point(&o->pdfT, o->pos[o->tri[3 * t]]);
                    const v3 w1 = m_point(&o->pdfT, o->pos[o->tri[3 * t + 1]]);
                    const v3 w2 = m_point(&o->pdfT, o->pos[o->tri[3 * t + 2]]);
                    const float a = 2.0f / v_norm(v_cross(v_sub(w1, w0), v_sub(w2, w0)));
                    const float g = v_norm2(v_sub(tr.position, wray.point)) / fabsf(v_dot(tr.normal, wray.dir));
                    p = a * g;
                }
                sum += p;
            }
            ret += sum / (float)o->ntri;
        }
        pdf += ret / (float)s->nlights;
        n++;
    }
    if (s->env_type) {                          /* Env_Sphere::pdf 1/(4 PI), Env_Hemisphere::pdf 1/(2 PI) */
        pdf += (s->env_type == 2) ? (1.0f / (2.0f * PI_F)) : (1.0f / (4.0f * PI_F));
        n++;
    }
    if (n) pdf /= n;
    return pdf;
}

/* ------------------------------------------------------------------------------------------------
 * student/pathtracer.cpp: trace (174-218), sample_direct_lighting (78-172),
 * sample_indirect_lighting (42-76), trace_pixel (14-40); student/camera.cpp:7-34
 * ---------------------------------------------------------------------------------------------- */
typedef struct { spec emissive, reflected; } pair_t;
typedef struct { const material_t* bsdf; m4 w2o, o2w; v3 pos, out_dir, normal; uint32_t depth; } shading_t;
static pair_t pt_trace(ctx_t* c, const ray_t* ray);

/* Delta_Light::sample (rays/light.h:86-91) over Directional / Point / Spot_Light::sample (rays/light.cpp:5-31) */
typedef struct { spec radiance; v3 direction; float distance; } light_sample_t;
static light_sample_t delta_light_sample(const scene_t* s, const struct delta_light* l, v3 from) {
    light_sample_t r;
    if (l->has_trans) from = m_point(&l->itrans, from);
    r.radiance = l->radiance;
    if (l->type == 0) {                                              /* Directional_Light */
        r.direction = V(0.0f, -1.0f, 0.0f);
        r.distance = INFINITY;
    } else {
        r.direction = v_neg(v_unit(from));                           /* Point_Light / Spot_Light */
        r.distance = v_norm(from);
        if (l->type == 2) {
            float angle = m_atan2(s, sqrtf(from.x * from.x + from.z * from.z), from.y);   /* Vec2(x, z).norm() */
            angle = fabsf(angle * (180.0f / PI_F));                                       /* std::abs(Degrees(angle)) */
            const float e0 = l->angle_bounds[0] / 2.0f, e1 = l->angle_bounds[1] / 2.0f;
            const float t = min_f(max_f((angle - e0) / (e1 - e0), 0.0f), 1.0f);           /* smoothstep, lib/mathlib.h:47-50 */
            r.radiance = s_scale(r.radiance, 1.0f - t * t * (3.0f - 2.0f * t));
        }
    }
    if (l->has_trans) r.direction = m_rotate(&l->trans, r.direction);
    return r;
}

static trace_t scene_hit(ctx_t* c, const ray_t* ray);
/* Pathtracer::point_lighting, rays/pathtracer.cpp:327-348 */
static spec point_lighting(ctx_t* c, const shading_t* h) {
    const scene_t* s = c->s;
    spec radiance = S(0, 0, 0);
    if (mat_discrete(h->bsdf)) return radiance;
    for (uint32_t i = 0; i < s->ndlights; i++) {
        light_sample_t ls = delta_light_sample(s, &s->dlights[i], h->pos);
        /* in_dir = world_to_object.rotate(sample.direction) only feeds evaluate(), which ignores it for Lambertian */
        spec att = lambert_evaluate(s, h->bsdf, h->out_dir);
        if (s_luma(att) == 0.0f) continue;
        ray_t shadow = ray_make(h->pos, ls.direction, EPS_F, ls.distance - EPS_F, 0);
        trace_t t = scene_hit(c, &shadow);
        if (!t.hit) radiance = s_add(radiance, s_mul(att, ls.radiance));
    }
    return radiance;
}

static spec sample_direct(ctx_t* c, const shading_t* h) {
    const scene_t* s = c->s;
    spec radiance = point_lighting(c, h);
    scatter_t in = bsdf_scatter(c, h->bsdf, h->out_dir);
    const v3 world_in = m_rotate(&h->o2w, in.direction);
    ray_t wr = ray_make(h->pos, world_in, EPS_F, FLT_MAX, 0);
    spec direct = pt_trace(c, &wr).emissive;
    float pdf;
    if (mat_discrete(h->bsdf)) {
        direct = s_mul(direct, in.attenuation);
    } else {
        pdf = lambert_pdf(s, h->out_dir);
        direct = s_scale(s_mul(direct, in.attenuation), 1.0f / pdf);
    }
    radiance = s_add(radiance, direct);
    if (mat_discrete(h->bsdf)) return radiance;
    radiance = s_sub(radiance, direct);
    const v3 to_light = sample_area_lights(c, h->pos);
    const v3 chosen = rng_coin(c, 0.5f) ? world_in : to_light;
    ray_t r6 = ray_make(h->pos, chosen, EPS_F, FLT_MAX, 0);
    /* if(RNG::coin_flip(0.0005f)) log_ray(world_ray_task6, 5.0f);  (pathtracer.cpp:148 -> rays/pathtracer.cpp:191-193 ->
     * Gui::Widget_Render::log_ray, gui/widgets.cpp:625-628) - the coin is always flipped */
    if (rng_coin(c, 0.0005f) && c->log) {
        raylog_t* L = c->log;
        if (L->n < L->cap) {
            float* e = L->buf + 10 * L->n;
            e[0] = r6.point.x; e[1] = r6.point.y; e[2] = r6.point.z;
            e[3] = r6.dir.x; e[4] = r6.dir.y; e[5] = r6.dir.z;
            e[6] = 5.0f;
            e[7] = (float)(uint32_t)(c->rng.inc >> 33);                   /* pixel = y * w + x (exact below 2^24) */
            e[8] = (float)(uint32_t)((c->rng.inc >> 1) & 0xffffffu);      /* sample */
            e[9] = (float)(s->max_depth - h->depth);                      /* bounce: the camera ray carries depth = max_depth */
        }
        L->n++;
    }
    spec d6 = pt_trace(c, &r6).emissive;
    const float pdf_area = area_lights_pdf(c, h->pos, to_light);
    const float pdf4 = lambert_pdf(s, h->out_dir);
    pdf = (pdf4 + pdf_area) / 2.0f;
    const spec att6 = lambert_evaluate(s, h->bsdf, h->out_dir);
    d6 = s_scale(s_mul(d6, att6), 1.0f / pdf);
    radiance = s_add(radiance, d6);
    return radiance;
}
static spec sample_indirect(ctx_t* c, const shading_t* h) {
    const scene_t* s = c->s;
    spec radiance = S(0, 0, 0);
    scatter_t in = bsdf_scatter(c, h->bsdf, h->out_dir);
    const v3 world_in = m_rotate(&h->o2w, in.direction);
    ray_t wr = ray_make(h->pos, world_in, EPS_F, FLT_MAX, h->depth - 1);
    spec indirect = pt_trace(c, &wr).reflected;
    if (mat_discrete(h->bsdf)) {
        indirect = s_mul(indirect, in.attenuation);
    } else {
        float pdf = lambert_pdf(s, h->out_dir);
        indirect = s_scale(s_mul(indirect, in.attenuation), 1.0f / pdf);
    }
    radiance = s_add(radiance, indirect);
    return radiance;
}
static pair_t pt_trace(ctx_t* c, const ray_t* ray) {
    const scene_t* s = c->s;
    pair_t out = {{0, 0, 0}, {0, 0, 0}};
    trace_t result = scene_hit(c, ray);
    if (!result.hit) {                          /* student/pathtracer.cpp:182-188 */
        if (s->env_type) out.emissive = env_evaluate(s, ray->dir);
        return out;
    }
    const material_t* bsdf = &s->mats[result.material];
    if (!mat_sided(bsdf) && v_dot(result.normal, ray->dir) > 0.0f) result.normal = v_neg(result.normal);
    spec emissive = mat_emissive(bsdf);
    if (s_luma(emissive) > 0.0f) { out.emissive = emissive; return out; }
    if (ray->depth == 0) return out;
    shading_t h;
    h.bsdf = bsdf;
    h.o2w = m_rotate_to(result.normal);
    h.w2o = m_transpose(&h.o2w);
    h.out_dir = v_unit(m_rotate(&h.w2o, v_sub(ray->point, result.position)));
    h.pos = result.position;
    h.normal = result.normal;
    h.depth = ray->depth;
    /* operands of '+' are unsequenced in C++; the reference build (clang) evaluates direct first */
    const spec d = sample_direct(c, &h);
    const spec i = sample_indirect(c, &h);
    out.emissive = emissive;
    out.reflected = s_add(d, i);
    return out;
}
static ray_t camera_ray(const scene_t* s, float sx, float sy) {                /* student/camera.cpp:7-34 */
    const float sh = tanf((s->vfov * (PI_F / 180.0f)) / 2.0f) * 1.0f * 2.0f;
    const float sw = s->ar * sh;
    const float px = sx * sw - 0.5f * sw;
    const float py = sy * sh - 0.5f * sh;
    ray_t r;
    r.point = V(0, 0, 0);
    r.dir = V(px, py, -1.0f);
    r.b0 = 0.0f; r.b1 = INFINITY;
    r.depth = 0;
    ray_transform(&r, &s->iview);
    return r;
}
static spec trace_pixel(ctx_t* c, uint32_t x, uint32_t y) {
    const scene_t* s = c->s;
    const float jx = rng_unit(c) * 1.0f;
    const float jy = rng_unit(c) * 1.0f;
    ray_t ray = camera_ray(s, ((float)x + jx) / (float)s->w, ((float)y + jy) / (float)s->h);
    ray.depth = s->max_depth;
    pair_t p = pt_trace(c, &ray);
    return s_add(p.emissive, p.reflected);
}

/* ------------------------------------------------------------------------------------------------
 * Scene assembly: Object ctor (rays/object.h:18-34), Tri_Mesh::build (tri_mesh.cpp:145-170),
 * tail of build_scene (rays/pathtracer.cpp:165-175)
 * ---------------------------------------------------------------------------------------------- */
static m4 m_from(const float* f) { m4 r; memcpy(&r, f, sizeof r); return r; }
static int m_is_identity(const m4* m) { m4 i = m_identity(); return memcmp(m, &i, sizeof i) == 0 ? 1 : 0; }
static int m_ne_identity(const m4* m) {        /* operator!= compares values (so -0.0 == 0.0) */
    const m4 i = m_identity();
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) if (m->c[a][b] != i.c[a][b]) return 1;
    return 0;
}

void* srt_oracle_pt_create(void) {
    scene_t* s = (scene_t*)calloc(1, sizeof(scene_t));
    s->iview = m_identity(); s->vfov = 90.0f; s->ar = 1.0f; s->w = s->h = 1; s->max_depth = 8;
    return s;
}
int srt_oracle_pt_set_math(void* h, int mode) { ((scene_t*)h)->math_mode = mode ? 1 : 0; return 0; }

int srt_oracle_pt_add_material(void* h, int type, const float a[3], const float b[3], float ior) {
    scene_t* s = (scene_t*)h;
    if (type < 0 || type > 4) return -1;
    s->mats = (material_t*)realloc(s->mats, (s->nmats + 1) * sizeof(material_t));
    material_t* m = &s->mats[s->nmats];
    m->type = type; m->a = S(a[0], a[1], a[2]); m->b = S(b[0], b[1], b[2]); m->ior = ior;
    if (type == MAT_LAMBERTIAN) m->a = S(a[0] / PI_F, a[1] / PI_F, a[2] / PI_F);   /* rays/bsdf.h:26 */
    return (int)s->nmats++;
}
static void object_init(object_t* o, const float T[16], int material, uint32_t id) {
    memset(o, 0, sizeof *o);
    o->trans = m_from(T);
    o->itrans = m_inverse(&o->trans);
    o->has_trans = m_ne_identity(&o->trans);
    o->material = material;
    o->id = id;
    (void)m_is_identity;
}
static void mesh_fill(object_t* o, const float* pos, const float* nrm, uint32_t nv, const uint32_t* idx, uint32_t ni) {
    o->kind = OBJ_MESH;
    o->nverts = nv; o->ntri = ni / 3;
    o->pos = (v3*)malloc(nv * sizeof(v3)); o->nrm = (v3*)malloc(nv * sizeof(v3));
    memcpy(o->pos, pos, nv * sizeof(v3)); memcpy(o->nrm, nrm, nv * sizeof(v3));
    o->tri = (uint32_t*)malloc(o->ntri * 3 * sizeof(uint32_t));
    memcpy(o->tri, idx, o->ntri * 3 * sizeof(uint32_t));
}
int srt_oracle_pt_add_mesh(void* h, const float* pos, const float* nrm, uint32_t nv, const uint32_t* idx, uint32_t ni,
                           const float T[16], uint32_t material, int is_light) {
    scene_t* s = (scene_t*)h;
    if (s->committed) return -1;
    const uint32_t id = s->nobjs + 1;
    if (is_light) {
        s->lights = (object_t*)realloc(s->lights, (s->nlights + 1) * sizeof(object_t));
        object_t* l = &s->lights[s->nlights++];
        object_init(l, T, (int)material, id);
        mesh_fill(l, pos, nrm, nv, idx, ni);
        l->use_bvh = 0;
        const m4 I = m_identity();
        l->pdfT = I; l->pdfiT = I;
        if (l->has_trans) { l->pdfT = m_mul(&I, &l->trans); l->pdfiT = m_mul(&l->itrans, &I); }
    }
    s->objs = (object_t*)realloc(s->objs, (s->nobjs + 1) * sizeof(object_t));
    object_t* o = &s->objs[s->nobjs++];
    object_init(o, T, (int)material, id);
    mesh_fill(o, pos, nrm, nv, idx, ni);
    return 0;
}
/* An emissive Shape (rays/pathtracer.cpp:105-131): the analytic sphere in the scene, its mesh approximation in area_lights */
int srt_oracle_pt_add_sphere_light(void* h, float radius, const float T[16], uint32_t material, const float* pos,
                                   const float* nrm, uint32_t nv, const uint32_t* idx, uint32_t ni) {
    scene_t* s = (scene_t*)h;
    if (s->committed) return -1;
    const uint32_t id = s->nobjs + 1;
    s->lights = (object_t*)realloc(s->lights, (s->nlights + 1) * sizeof(object_t));
    object_t* l = &s->lights[s->nlights++];
    object_init(l, T, (int)material, id);
    mesh_fill(l, pos, nrm, nv, idx, ni);
    l->use_bvh = 0;
    const m4 I = m_identity();
    l->pdfT = I; l->pdfiT = I;
    if (l->has_trans) { l->pdfT = m_mul(&I, &l->trans); l->pdfiT = m_mul(&l->itrans, &I); }
    s->objs = (object_t*)realloc(s->objs, (s->nobjs + 1) * sizeof(object_t));
    object_t* o = &s->objs[s->nobjs];
    object_init(o, T, (int)material, s->nobjs + 1);
    o->kind = OBJ_SPHERE; o->radius = radius;
    s->nobjs++;
    return 0;
}
/* Pathtracer::build_lights (rays/pathtracer.cpp:26-64): type 0 directional, 1 point, 2 spot */
int srt_oracle_pt_add_light(void* h, uint32_t type, const float radiance[3], const float angle_bounds[2], const float T[16]) {
    scene_t* s = (scene_t*)h;
    if (s->committed || type > 2) return -1;
    s->dlights = (struct delta_light*)realloc(s->dlights, (s->ndlights + 1) * sizeof(struct delta_light));
    struct delta_light* l = &s->dlights[s->ndlights++];
    memset(l, 0, sizeof *l);
    l->type = (int)type;
    l->radiance = S(radiance[0], radiance[1], radiance[2]);
    if (angle_bounds) { l->angle_bounds[0] = angle_bounds[0]; l->angle_bounds[1] = angle_bounds[1]; }
    l->trans = m_from(T);
    l->itrans = m_inverse(&l->trans);
    l->has_trans = m_ne_identity(&l->trans);
    return 0;
}

/* Pathtracer::env_light: 0 none, 1 Env_Sphere(radiance), 2 Env_Hemisphere(radiance) */
int srt_oracle_pt_set_env_light(void* h, uint32_t type, const float radiance[3]) {
    scene_t* s = (scene_t*)h;
    if (s->committed || type > 2) return -1;
    s->env_type = (int)type;
    s->env_radiance = type ? S(radiance[0], radiance[1], radiance[2]) : S(0, 0, 0);
    return 0;
}

/* Env_Map(image): rgb = w * h * 3 floats, pixel (x, y) at y * w + x */
int srt_oracle_pt_set_env_map(void* h, uint32_t w, uint32_t hh, const float* rgb) {
    scene_t* s = (scene_t*)h;
    if (s->committed || !w || !hh || !rgb) return -1;
    free(s->env_map);
    s->env_map = (float*)malloc(sizeof(float) * 3 * (size_t)w * hh);
    if (!s->env_map) return -1;
    memcpy(s->env_map, rgb, sizeof(float) * 3 * (size_t)w * hh);
    s->env_w = w; s->env_h = hh; s->env_type = 3;
    return 0;
}

int srt_oracle_pt_add_sphere(void* h, float radius, const float T[16], uint32_t material) {
    scene_t* s = (scene_t*)h;
    if (s->committed) return -1;
    s->objs = (object_t*)realloc(s->objs, (s->nobjs + 1) * sizeof(object_t));
    object_t* o = &s->objs[s->nobjs];
    object_init(o, T, (int)material, s->nobjs + 1);
    o->kind = OBJ_SPHERE; o->radius = radius;
    s->nobjs++;
    return 0;
}
static box3 object_bbox(const object_t* o) {                                   /* rays/object.h:51-55 */
    box3 b;
    if (o->kind == OBJ_SPHERE) {                                               /* shapes.cpp:9-15 */
        b = box_empty();
        box_enclose_pt(&b, V(-o->radius, -o->radius, -o->radius));
        box_enclose_pt(&b, V(o->radius, o->radius, o->radius));
    } else if (o->use_bvh) {
        b = o->bvh.nodes[0].b;
    } else {
        b = box_empty();
        for (uint32_t t = 0; t < o->ntri; t++) box_enclose(&b, tri_bbox(o, t));
    }
    if (o->has_trans) b = box_transform(b, &o->trans);
    return b;
}
int srt_oracle_pt_commit(void* h, int use_bvh) {
    scene_t* s = (scene_t*)h;
    if (s->committed) return -1;
    s->use_bvh = use_bvh ? 1 : 0;
    for (uint32_t i = 0; i < s->nobjs; i++) {
        object_t* o = &s->objs[i];
        if (o->kind != OBJ_MESH) continue;
        o->use_bvh = s->use_bvh;
        if (o->use_bvh) {
            box3* pb = (box3*)malloc((o->ntri ? o->ntri : 1) * sizeof(box3));
            for (uint32_t t = 0; t < o->ntri; t++) pb[t] = tri_bbox(o, t);
            int rc = bvh_build(&o->bvh, pb, o->ntri, 4);                       /* tri_mesh.cpp:164 */
            free(pb);
            if (rc) return -2;
        }
    }
    if (s->use_bvh) {
        box3* pb = (box3*)malloc((s->nobjs ? s->nobjs : 1) * sizeof(box3));
        for (uint32_t i = 0; i < s->nobjs; i++) pb[i] = object_bbox(&s->objs[i]);
        int rc = bvh_build(&s->tlas, pb, s->nobjs, 1);                         /* rays/bvh.h:14 */
        free(pb);
        if (rc) return -2;
    }
    s->committed = 1;
    return 0;
}
int srt_oracle_pt_set_camera(void* h, const float iview[16], float vfov, float ar) {
    scene_t* s = (scene_t*)h;
    s->iview = m_from(iview); s->vfov = vfov; s->ar = ar;
    return 0;
}
int srt_oracle_pt_set_params(void* h, uint32_t w, uint32_t hh, uint32_t max_depth) {
    scene_t* s = (scene_t*)h;
    if (!w || !hh) return -1;
    s->w = w; s->h = hh; s->max_depth = max_depth;
    return 0;
}
static void cnt_out(const counters_t* c, uint64_t out[8]) {
    if (!out) return;
    out[0] += c->rays; out[1] += c->box_tests; out[2] += c->obj_entered; out[3] += c->tri_tests;
    out[4] += c->sphere_tests; out[5] += c->tlas_nodes; out[6] += c->blas_nodes; out[7] += c->light_tri_tests;
}

/* trace_pixel for a list of (x, y, sample); counters (optional, 8 x u64) are ADDED to. */
int srt_oracle_pt_trace_samples(void* h, uint64_t seed, const uint32_t* xs, const uint32_t* ys, const uint32_t* ss,
                                size_t n, float* rgb_out, uint32_t* draws_out, uint32_t* rays_out, uint64_t counters[8]) {
    const scene_t* s = (const scene_t*)h;
    if (!s->committed) return -1;
    ctx_t c; memset(&c, 0, sizeof c); c.s = s;
    for (size_t k = 0; k < n; k++) {
        rng_key(&c.rng, seed, ys[k] * s->w + xs[k], ss[k]);
        const uint64_t r0 = c.cnt.rays;
        spec p = trace_pixel(&c, xs[k], ys[k]);
        rgb_out[3 * k] = p.r; rgb_out[3 * k + 1] = p.g; rgb_out[3 * k + 2] = p.b;
        if (draws_out) draws_out[k] = c.rng.draws;
        if (rays_out) rays_out[k] = (uint32_t)(c.cnt.rays - r0);
    }
    cnt_out(&c.cnt, counters);
    return 0;
}

/* One epoch of do_trace (rays/pathtracer.cpp:209-231) restricted to rows [y0, y1): per pixel the
 * mean of the valid samples sample_base .. sample_base+samples-1.  img: w*h*3 floats, row 0 = bottom.
 * Thread-safe for disjoint row ranges (bench.py's cpu_baseline runs one range per host thread). */
int srt_oracle_pt_epoch_rows(void* h, uint64_t seed, uint32_t sample_base, uint32_t samples, uint32_t y0, uint32_t y1,
                             float* img, uint64_t counters[8]) {
    const scene_t* s = (const scene_t*)h;
    if (!s->committed || y1 > s->h) return -1;
    ctx_t c; memset(&c, 0, sizeof c); c.s = s;
    for (uint32_t j = y0; j < y1; j++) {
        for (uint32_t i = 0; i < s->w; i++) {
            spec acc = S(0, 0, 0);
            size_t sampled = 0;
            for (uint32_t k = 0; k < samples; k++) {
                rng_key(&c.rng, seed, j * s->w + i, sample_base + k);
                spec p = trace_pixel(&c, i, j);
                if (s_valid(p)) { acc = s_add(acc, p); sampled++; }
            }
            if (sampled > 0) acc = s_scale(acc, 1.0f / sampled);
            float* o = img + 3 * ((size_t)j * s->w + i);
            o[0] = acc.r; o[1] = acc.g; o[2] = acc.b;
        }
    }
    cnt_out(&c.cnt, counters);
    return 0;
}

/* srt_oracle_pt_epoch_rows, and the rays the epoch hands to Pathtracer::log_ray in the order a single-threaded do_trace logs them
 * (rows, pixels, samples, bounces): log10 = 10 floats per ray {point[3], dir[3], t, pixel, sample, bounce}; *n_logged counts all of
 * them, also those beyond cap. */
int srt_oracle_pt_epoch_rows_log(void* h, uint64_t seed, uint32_t sample_base, uint32_t samples, uint32_t y0, uint32_t y1,
                                 float* img, float* log10, size_t cap, size_t* n_logged) {
    const scene_t* s = (const scene_t*)h;
    if (!s->committed || y1 > s->h) return -1;
    raylog_t L; L.buf = log10; L.cap = log10 ? cap : 0; L.n = 0;
    ctx_t c; memset(&c, 0, sizeof c); c.s = s; c.log = &L;
    for (uint32_t j = y0; j < y1; j++) {
        for (uint32_t i = 0; i < s->w; i++) {
            spec acc = S(0, 0, 0);
            size_t sampled = 0;
            for (uint32_t k = 0; k < samples; k++) {
                rng_key(&c.rng, seed, j * s->w + i, sample_base + k);
                spec p = trace_pixel(&c, i, j);
                if (s_valid(p)) { acc = s_add(acc, p); sampled++; }
            }
            if (sampled > 0) acc = s_scale(acc, 1.0f / sampled);
            if (img) { float* o = img + 3 * ((size_t)j * s->w + i); o[0] = acc.r; o[1] = acc.g; o[2] = acc.b; }
        }
    }
    if (n_logged) *n_logged = L.n;
    return 0;
}

/* Pathtracer::accumulate (rays/pathtracer.cpp:195-207): running mean of epoch means. */
int srt_oracle_pt_accumulate(float* accumulator, const float* epoch, size_t nfloats, uint32_t accumulator_samples) {
    for (size_t i = 0; i < nfloats; i++) accumulator[i] += (epoch[i] - accumulator[i]) * (1.0f / accumulator_samples);
    return 0;
}

/* scene.hit for explicit rays; out: 9 floats per ray {hit, distance, position, normal, material}. */
int srt_oracle_pt_hit(void* h, const float* org, const float* dir, const float* bounds, size_t n, float* out9) {
    const scene_t* s = (const scene_t*)h;
    ctx_t c; memset(&c, 0, sizeof c); c.s = s;
    for (size_t k = 0; k < n; k++) {
        ray_t r;
        r.point = V(org[3 * k], org[3 * k + 1], org[3 * k + 2]);
        r.dir = V(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
        r.b0 = bounds[2 * k]; r.b1 = bounds[2 * k + 1]; r.depth = 0;
        trace_t t = scene_hit(&c, &r);
        float* o = out9 + 9 * k;
        o[0] = t.hit ? 1.0f : 0.0f; o[1] = t.distance;
        o[2] = t.position.x; o[3] = t.position.y; o[4] = t.position.z;
        o[5] = t.normal.x; o[6] = t.normal.y; o[7] = t.normal.z;
        o[8] = (float)t.material;
    }
    return 0;
}

/* Scene_Particles::Particle::update (student/particles.cpp:5-59), the loop body of Scene_Particles::step2
 * (scene/particles.cpp:134-138): the particle flies at constant velocity for what is left of dt, bounces off what
 * scene.hit(Ray(pos, velocity)) reports (default bounds [0, inf], direction NOT normalised), gravity acts on the velocity
 * after every leg.  The unqualified sqrt is the double overload.  max_iter = 0: loop like the reference (which never
 * returns when hit_time stays <= 0); else give up after that many legs (what the device kernel does). */
int srt_oracle_pt_particles_update(void* h, float* pos, float* vel, float* age, size_t n, float dt, float radius, unsigned char* alive,
                                   uint32_t max_iter) {
    const scene_t* s = (const scene_t*)h;
    ctx_t c; memset(&c, 0, sizeof c); c.s = s;
    const v3 acceleration = V(0.0f, -9.8f, 0.0f);
    for (size_t k = 0; k < n; k++) {
        v3 p = V(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]);
        v3 velocity = V(vel[3 * k], vel[3 * k + 1], vel[3 * k + 2]);
        float remain = dt;
        uint32_t it = 0;
        while (remain > 0) {
            if (max_iter && it++ >= max_iter) break;
            ray_t r;
            r.point = p; r.dir = velocity; r.b0 = 0.0f; r.b1 = INFINITY; r.depth = 0;
            trace_t t = scene_hit(&c, &r);
            float cos_t = v_dot(t.normal, v_scale(velocity, -1.0f)) / (v_norm(velocity) * v_norm(t.normal));
            v3 surface_normal = v_divs(t.normal, v_norm(t.normal));
            if (cos_t < 0) {
                cos_t = (float)sqrt((double)(1 - cos_t * cos_t));
                surface_normal = v_scale(surface_normal, -1.0f);
            }
            const float interval = fabsf(radius / cos_t);
            const float hit_time = (t.distance - interval) / v_norm(r.dir);
            if (!t.hit || hit_time > remain || cos_t == 0) {
                p = v_add(p, v_scale(velocity, remain));
                velocity = v_add(velocity, v_scale(acceleration, remain));
                break;
            }
            p = v_sub(t.position, v_divs(v_scale(velocity, interval), v_norm(velocity)));
            velocity = v_sub(velocity, v_scale(surface_normal, 2.0f * v_dot(velocity, surface_normal)));
            velocity = v_add(velocity, v_scale(acceleration, hit_time));
            remain -= hit_time;
        }
        age[k] -= dt;
        alive[k] = age[k] > 0 ? 1 : 0;
        pos[3 * k] = p.x; pos[3 * k + 1] = p.y; pos[3 * k + 2] = p.z;
        vel[3 * k] = velocity.x; vel[3 * k + 1] = velocity.y; vel[3 * k + 2] = velocity.z;
    }
    return 0;
}

/* Node arrays: which = -1 the scene BVH<Object> (order[] = object ids, 1-based insertion index, in BVH
 * primitive order); which >= 0 the BVH<Triangle> of the which-th scene-BVH primitive (order[] = first
 * vertex index of each triangle in BVH primitive order). */
long srt_oracle_pt_dump_bvh(void* h, int which, float* boxes, uint32_t* links, size_t cap, uint32_t* order) {
    const scene_t* s = (const scene_t*)h;
    if (!s->use_bvh) return -1;
    const bvh_t* b = &s->tlas;
    const object_t* o = NULL;
    if (which >= 0) {
        if ((uint32_t)which >= s->nobjs) return -1;
        o = &s->objs[s->tlas.prim[which]];
        if (o->kind != OBJ_MESH) return -2;
        b = &o->bvh;
    }
    for (uint32_t i = 0; i < b->nnodes && i < cap; i++) {
        const node_t* nd = &b->nodes[i];
        float* bx = boxes + 6 * i;
        bx[0] = nd->b.mn.x; bx[1] = nd->b.mn.y; bx[2] = nd->b.mn.z;
        bx[3] = nd->b.mx.x; bx[4] = nd->b.mx.y; bx[5] = nd->b.mx.z;
        uint32_t* l = links + 4 * i;
        l[0] = nd->start; l[1] = nd->size; l[2] = nd->l; l[3] = nd->r;
    }
    if (order) {
        for (uint32_t i = 0; i < b->nprims; i++)
            order[i] = o ? o->tri[3 * b->prim[i]] : s->objs[b->prim[i]].id;
    }
    return (long)b->nnodes;
}

/* cosf/sinf of SRT-MATH v2, exposed so tests can compare them with libm and with the kernel. */
int srt_oracle_math_acos(const float* x, size_t n, float* out) {
    for (size_t i = 0; i < n; i++) out[i] = srt_acosf(x[i]);
    return 0;
}
int srt_oracle_math_atan2(const float* y, const float* x, size_t n, float* out) {
    for (size_t i = 0; i < n; i++) out[i] = srt_atan2f(y[i], x[i]);
    return 0;
}
int srt_oracle_math_cos_sin(const float* x, size_t n, float* cos_out, float* sin_out) {
    for (size_t i = 0; i < n; i++) { cos_out[i] = srt_cosf(x[i]); sin_out[i] = srt_sinf(x[i]); }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * SRT-MATH v2, expf / powf: glibc 2.35 sysdeps/ieee754/flt-32/e_expf.c, e_powf.c, e_exp2f_data.c,
 * e_powf_log2_data.c (the ARM optimized-routines code by Szabolcs Nagy), x86-64 configuration
 * (TOINT_INTRINSICS 0, hence POWF_SCALE 1 and the 0x1.8p52 shift trick).  fp64 arithmetic, one rounding
 * to fp32 at the end.  The multiply-adds are FUSED: on an x86-64 host with FMA (every AVX2 machine) glibc's
 * ifunc selects __expf_fma / __powf_fma, the same source compiled with -mfma, where GCC contracts each a*b+c.
 * Exhaustive comparison with this host's libm decides it: with fma() the restatement of expf is identical for
 * every float and powf for every x in [2^-11, 2^9) at y = 5, 10, 7, 3, -4, 2.4 and every x in the sRGB
 * range at y = 1/2.4; evaluated without FMA, expf differs for 2 of 2^32 arguments and powf for a few in 10^8.
 * (sinf / cosf above have FMA builds too; over every float of [0, 2pi] and [-1, 1] the unfused restatement
 * is identical to them.)  The tables were read against __exp2f_data / __powf_log2_data in this image's
 * libm.so.6; exp2f's is asuint64(2^(i/32)) - (i << 52)/32.  Needed by HDR_Image::tonemap_to
 * (util/hdr_image.cpp:161-187: 1 - exp(-sample * exposure)) and Spectrum::to_srgb
 * (lib/spectrum.h:61-66: pow(f, 1/2.4)).  Domain restated: expf for every float; powf for normal x > 0 and
 * finite y != 0 (to_srgb calls it with 0.0031308 <= x <= 1) - NaN for any other operand.
 * tests/test_pt_oracle.py compares both with this host's libm.
 * ---------------------------------------------------------------------------------------------- */
static const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
    0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
    0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
static const double EXP2F_POLY[3] = {0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1};
static uint64_t d_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static double bits_d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

static float srt_expf(float x) {
    const double xd = (double)x;
    const uint32_t abstop = (f_bits(x) >> 20) & 0x7ff;
    if (abstop >= 0x42b) {                              /* |x| >= 88 or NaN */
        if (f_bits(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return bits_f(0x7f800000u);      /* overflow */
        if (x < -0x1.9fe368p6f) return 0.0f;                    /* underflow to zero */
    }
    const double invln2n = 0x1.71547652b82fep+5, shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-20, c1 = 0x1.ebfce50fac4f3p-13, c2 = 0x1.62e42ff0c52d6p-6;   /* poly_scaled */
    double z = invln2n * xd;
    double kd = z + shift;
    const uint64_t ki = d_bits(kd);
    kd -= shift;
    const double r = fma(invln2n, xd, -kd);              /* z - kd with the product of z fused in */
    uint64_t t = EXP2F_TAB[ki % 32];
    t += ki << (52 - 5);
    const double s = bits_d(t);
    z = fma(c0, r, c1);
    const double r2 = r * r;
    double y = fma(c2, r, 1.0);
    y = fma(z, r2, y);
    y = y * s;
    return (float)y;
}

static const double POWF_LOG2_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
static const double POWF_LOG2_POLY[5] = {0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2,
                                         -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp+0};
static float srt_powf(float x, float y) {
    const uint32_t ix = f_bits(x), iy = f_bits(y);
    const uint32_t ay = iy & 0x7fffffffu;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || ay == 0 || ay >= 0x7f800000u) return bits_f(0x7fc00000u);
    /* log2_inline */
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = POWF_LOG2_TAB[i][0], logc = POWF_LOG2_TAB[i][1];
    const double z = (double)bits_f(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double* A = POWF_LOG2_POLY;
    const double r2 = r * r;
    double yy = fma(A[0], r, A[1]);
    const double p = fma(A[2], r, A[3]);
    const double r4 = r2 * r2;
    double q = fma(A[4], r, y0);
    q = fma(p, r2, q);
    yy = fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    if (((d_bits(ylogx) >> 47) & 0xffff) >= (d_bits(126.0) >> 47)) {                             /* |y log2 x| >= 126 */
        if (ylogx > 0x1.fffffffd1d571p+6) return bits_f(0x7f800000u);                            /* overflow */
        if (ylogx <= -150.0) return 0.0f;                                                        /* underflow */
    }
    /* exp2_inline, sign_bias 0 */
    const double shift = 0x1.8p+47;                       /* 0x1.8p52 / 32 */
    double kd = ylogx + shift;
    const uint64_t ki = d_bits(kd);
    kd -= shift;
    const double rr = ylogx - kd;
    uint64_t t = EXP2F_TAB[ki % 32];
    t += ki << (52 - 5);
    const double s = bits_d(t);
    double zz = fma(EXP2F_POLY[0], rr, EXP2F_POLY[1]);
    const double rr2 = rr * rr;
    double o = fma(EXP2F_POLY[2], rr, 1.0);
    o = fma(zz, rr2, o);
    o = o * s;
    return (float)o;
}
int srt_oracle_math_exp(const float* x, size_t n, float* out) {
    for (size_t i = 0; i < n; i++) out[i] = srt_expf(x[i]);
    return 0;
}
int srt_oracle_math_pow(const float* x, const float* y, size_t n, float* out) {
    for (size_t i = 0; i < n; i++) out[i] = srt_powf(x[i], y[i]);
    return 0;
}
/* Brute-force helpers for the tests (the sweeps are too large to ship as arrays): number of floats with bit
 * patterns in [lo, hi) where the restatement differs from this host's libm (NaN == NaN). */
uint64_t srt_oracle_sweep_exp_vs_libm(uint32_t lo, uint32_t hi, uint32_t* first_bad) {
    uint64_t bad = 0;
    for (uint64_t u = lo; u < hi; u++) {
        const float x = bits_f((uint32_t)u);
        const float a = srt_expf(x), b = expf(x);
        if (f_bits(a) != f_bits(b) && !(a != a && b != b)) { if (!bad && first_bad) *first_bad = (uint32_t)u; bad++; }
    }
    return bad;
}
uint64_t srt_oracle_sweep_sincos_vs_libm(uint32_t lo, uint32_t hi, uint32_t* first_bad) {
    uint64_t bad = 0;
    for (uint64_t u = lo; u < hi; u++) {
        const float x = bits_f((uint32_t)u);
        const float a = srt_sinf(x), b = sinf(x), c = srt_cosf(x), d = cosf(x);
        if (f_bits(a) != f_bits(b) || f_bits(c) != f_bits(d)) { if (!bad && first_bad) *first_bad = (uint32_t)u; bad++; }
    }
    return bad;
}
uint64_t srt_oracle_sweep_pow_vs_libm(uint32_t lo, uint32_t hi, float y, uint32_t* first_bad) {
    uint64_t bad = 0;
    for (uint64_t u = lo; u < hi; u++) {
        const float x = bits_f((uint32_t)u);
        const float a = srt_powf(x, y), b = powf(x, y);
        if (f_bits(a) != f_bits(b) && !(a != a && b != b)) { if (!bad && first_bad) *first_bad = (uint32_t)u; bad++; }
    }
    return bad;
}

/* HDR_Image::tonemap_to (util/hdr_image.cpp:161-187) with Spectrum::to_srgb (lib/spectrum.h:61-75): rows flipped,
 * 1 - exp(-c * exposure), sRGB transfer, (unsigned char)std::round(c * 255), alpha 255.  rgb: h*w*3 floats, row 0 first.
 * `exposure` is the image's exposure member, which is what the reference's loop reads (its argument e is unused there). */
static float to_srgb(float f) {
    if (f < 0.0031308f) return 12.92f * f;
    return 1.055f * srt_powf(f, 1.0f / 2.4f) - 0.055f;
}
/* (unsigned char)std::round(v * 255.0f) as x86-64 evaluates it: cvttss2si (0x80000000 for NaN / out of range), low byte */
static unsigned char to_byte(float v) {
    const float r = roundf(v * 255.0f);
    const int32_t i = (r >= -2147483648.0f && r < 2147483648.0f) ? (int32_t)r : INT32_MIN;
    return (unsigned char)((uint32_t)i & 0xffu);
}
int srt_oracle_tonemap(uint32_t w, uint32_t h, const float* rgb, float exposure, unsigned char* rgba) {
    for (uint32_t j = 0; j < h; j++)
        for (uint32_t i = 0; i < w; i++) {
            const float* s = rgb + 3 * ((size_t)(h - j - 1) * w + i);
            unsigned char* d = rgba + 4 * ((size_t)j * w + i);
            for (int c = 0; c < 3; c++) d[c] = to_byte(to_srgb(1.0f - srt_expf(-s[c] * exposure)));
            d[3] = 255;
        }
    return 0;
}

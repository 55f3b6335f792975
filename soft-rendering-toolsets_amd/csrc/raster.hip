// MI355X (gfx950) rasterizer: ordered triangle / point fill into a supersample tile held in LDS,
// fused box-filter resolve, coalesced RGBA8 write-back.  C ABI: include/srt_raster.h.
//
// What is being reproduced (reference paths relative to Assignments/DrawSVG/src/):
//   rasterize_triangle  software_renderer.cpp:456-516   float bbox, inclusive double loops, sample = corner
//   inside_triangle     software_renderer.cpp:519-538   fp64 edge functions rounded to fp32, fp32 sign products
//   fill_sample         software_renderer.cpp:634-658   non-premultiplied "over" on a float RGBA sample in [0,255]
//   rasterize_point     software_renderer.cpp:272-301   sr x sr block, double -> int truncation
//   rasterize_line      software_renderer.cpp:303-318   = rasterize_line_xiaolinwu :365-454: float end point / gradient math,
//                                                       the serial `intery += gradient` chain, its rasterize_point calls
//   rasterize_image     software_renderer.cpp:540-570   float x/y loops, fp64 u/v, Sampler2DImp::sample_trilinear
//   sample_bilinear     texture.cpp:145-169             4 texels of one mip level, lerpColor in fp32
//   resolve             software_renderer.cpp:573-622   fp32 box sum (x-offset outer, y-offset inner), /sr^2, (uint8_t)
//
// Execution model: one wavefront (64 lanes) owns one tile of at most 32x32 samples.  The supersample
// buffer of the reference (16 B/sample, 256 MiB at 1024^2 x 16 spp) is never materialised in HBM: a
// tile lives in LDS from clear to resolve (`tile[TS * TSY]` float4 in raster_tiles: 4 KiB for the 8-row tiles
// every sample rate up to 8 takes) - lane = (column lane & 31, row parity lane >> 5), two sample rows per
// iteration - and only 4 B/pixel leave the CU.  (A variant that kept the tile in registers was built, was
// bit-exact and measured slower)  The wave walks its bin's primitive list IN ORDER
// (painter's algorithm is order dependent), 64 bounding boxes per step, and rasterizes the overlapping
// ones into its tile.
//
// Numerics: compiled with -ffp-contract=off (the x86-64 reference build has no FMA), IEEE fp32
// division (hipcc default) and fp32 denormals preserved, so every coverage verdict and every blended
// sample is bit-identical to the CPU reference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "srt_common.h"

namespace {

constexpr int TS = 32;  // tile side in samples (max); tile_s = (TS / sr) * sr <= TS
constexpr int WAVE = 64;

struct RasterParams {
  uint32_t w, h, sr;
  uint32_t ssw, ssh;
  uint32_t tile_px;  // pixels per tile, x
  uint32_t tile_s;   // samples per tile, x (<= TS)
  uint32_t tile_py;  // pixels per tile, y
  uint32_t tile_sy;  // samples per tile, y (<= 16 for sample rates up to 16, else <= 32)
  uint32_t tiles_x, tiles_y;
  uint32_t nprims;
  uint32_t coarse_tiles;         // a coarse bin is coarse_tiles x coarse_tiles tiles
  uint32_t coarse_x, coarse_y;   // coarse grid
  uint32_t super_bx, super_by;   // a super-bin is super_bx x super_by coarse bins (first binning level)
  uint32_t super_x, super_y;     // super grid (<= 8 x 8)
  uint32_t super_stride;         // entries reserved per super-bin (= nprims)
  uint32_t packed_ok;            // a coarse bin spans <= 256 samples per side: the bin lists' packed boxes (4 x 8 bits, bin-relative) are valid
};

// Everything about one SRT_PRIM_IMAGE record that does not depend on the sample, prepared on the host at upload
// (upload_stream): the parameters after rasterize_image's `x0 *= sample_rate` (float *= size_t), the mip level
// arithmetic of sample_trilinear (it depends on the image extent only and uses the host's log2f, as the reference
// does), the two mip levels involved, and the values the reference's float loops `for (float x = x0; x <= x1; x++)`
// take: for every sample column / row of the target the (at most two) loop values that fill_sample's double->int
// conversion sends there, in loop order (NaN = none).  Two values per column only happen at column 0, where
// truncation toward zero folds (-1, 0) and [0, 1) together.
struct ImageAux {
  float x0s, y0s, x1s, y1s;
  int32_t mode;              // 0: magenta (level >= mipmap.size()), 1: bilinear at `low`, 2: lerp(frac, low, low + 1)
  int32_t low;
  float frac;
  uint32_t pad;
  uint32_t off[2], w[2], h[2];  // byte offset into the texel blob / size of levels low and low + 1
  uint32_t xtab, ytab;          // float offsets into the table buffer: [xtab, xtab + ssw) first values, then the second
  int32_t bx0, by0, bx1, by1;   // inclusive sample bbox of the filled columns / rows (bx0 > bx1: touches nothing)
};

// One SRT_PRIM_LINE record after raster_setup: rasterize_line_xiaolinwu's end points and main loop in the form the tile
// kernel needs.  Coordinates are pixels (the arguments of the reference's rasterize_point calls), kept as the reference's
// floats; (major, minor) = (x, y), or (y, x) for a steep line.  The main loop's serial `intery += gradient` chain is run
// once, by the setup thread of the line, into a table: entry k - k0 = intery at step k, for the steps whose major
// coordinate lies inside the target (k0 .. kmax).
struct LineAux {
  float e_major[2], e_minor[2];   // end point 1, 2: (xpxl, ypxl)
  float e_top[2], e_bot[2];       // coverage of pixel (xpxl, ypxl) and of (xpxl, ypxl + 1)
  int32_t first;                  // major coordinate of step 0 (xpxl1 + 1; integer-valued, |.| < 2^24 wherever steps exist)
  int32_t k0, kmax;               // steps with a table entry (kmax < k0: none)
  uint32_t tab;                   // offset of step k0's entry in the table
  uint32_t steep;
  uint32_t flat;                  // gradient == 0 (an axis-aligned line - the canvas outline, rectangles): intery never changes, x + 0 = x,
  float flat_y;                   // so the chain and the table are skipped and every step's value is this one
  uint32_t pad;
};
static_assert(sizeof(LineAux) == 64, "LineAux is read with scalar loads, 64 bytes");

// Device words a frame reports back (copied to the host with the image): what the line tables and the packed bin lists
// needed (the host grows the buffers and repeats the frame when they did not fit), refusals.
enum { FS_TABLE_NEED = 0, FS_LIST_NEED, FS_FLAGS, FS_COUNT };
constexpr uint32_t kFlagLineUnwalkable = 1u;   // a line whose main loop the reference's `++x` on a float could not walk

// stats slots (unsigned long long each)
enum { ST_TESTS_REF = 0, ST_TESTS_TARGET, ST_FRAGMENTS, ST_POINT_SAMPLES, ST_BIN_ENTRIES, ST_COUNT };

__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }  // std::min
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }  // std::max

// CMU462::clamp (CMU462/include/CMU462/misc.h:69): std::min(std::max(x, lo), hi)
__device__ __forceinline__ float clamp255(float x) { return std_min(std_max(x, 0.0f), 255.0f); }

// fill_sample's read-modify-write on one float RGBA sample (software_renderer.cpp:646-650).
__device__ __forceinline__ float4 blend_over(float4 s, float r, float g, float b, float one_minus_a) {
  s.x = clamp255((r + one_minus_a * (s.x / 255.0f)) * 255.0f);
  s.y = clamp255((g + one_minus_a * (s.y / 255.0f)) * 255.0f);
  s.z = clamp255((b + one_minus_a * (s.z / 255.0f)) * 255.0f);
  s.w = clamp255((1.0f - (one_minus_a * (1 - (s.w / 255.0f)))) * 255.0f);
  return s;
}

// ---------------------------------------------------------------------------------------------
// Pass 1: per-primitive sample-space bounding box clipped to the target, int4 {x0,y0,x1,y1}
// (inclusive; x0 > x1 marks "touches nothing").  Also the reference's own (unclipped) test count.
// ---------------------------------------------------------------------------------------------
// ipart / fpart / rfpart, software_renderer.cpp:355-363 (float in, float out)
__device__ __forceinline__ float wu_ipart(float x) { return floorf(x); }
__device__ __forceinline__ float wu_fpart(float x) { return x - floorf(x); }
__device__ __forceinline__ float wu_rfpart(float x) { return 1 - wu_fpart(x); }

// rasterize_line_xiaolinwu (software_renderer.cpp:365-454) for one LINE record: everything but the fills.  Writes the record's
// LineAux and its slice of the intery table (reserved with one atomic on status[FS_TABLE_NEED]; a slice that does not fit is
// not written - the host sees the need, grows the table and repeats the frame) and returns the clipped sample-space bounding
// box of the pixels the line's rasterize_point calls touch.
__device__ int4 setup_line(const RasterParams& P, const srt_prim& p, LineAux* __restrict__ out, float* __restrict__ table,
                           uint32_t table_cap, uint32_t* __restrict__ status) {
  float x0 = p.v.tri[0], y0 = p.v.tri[1], x1 = p.v.tri[2], y1 = p.v.tri[3];
  const bool steep = fabsf(x1 - x0) < fabsf(y1 - y0);
  if (steep) { float t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
  if (x0 > x1) { float t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
  const float dx = x1 - x0, dy = y1 - y0;
  const float gradient = (dx == 0.0f) ? 1.0f : dy / dx;
  LineAux A;
  A.steep = steep ? 1u : 0u; A.flat = 0u; A.flat_y = 0.0f; A.pad = 0u;
  // first end point
  float xend = roundf(x0);
  float yend = y0 + gradient * (xend - x0);
  float xgap = wu_rfpart(x0 + 0.5f);
  const float xpxl1 = xend, ypxl1 = wu_ipart(yend);
  A.e_major[0] = xpxl1; A.e_minor[0] = ypxl1; A.e_top[0] = wu_rfpart(yend) * xgap; A.e_bot[0] = wu_fpart(yend) * xgap;
  float intery = yend + gradient;   // first y-intersection for the main loop
  // second end point
  xend = roundf(x1);
  yend = y1 + gradient * (xend - x1);
  xgap = wu_fpart(x1 + 0.5f);
  const float xpxl2 = xend, ypxl2 = wu_ipart(yend);
  A.e_major[1] = xpxl2; A.e_minor[1] = ypxl2; A.e_top[1] = wu_rfpart(yend) * xgap; A.e_bot[1] = wu_fpart(yend) * xgap;
  // main loop: for (float x = xpxl1 + 1; x <= xpxl2 - 1 * sample_rate; ++x)
  const float first = xpxl1 + 1, last = xpxl2 - (float)P.sr;
  A.first = 0; A.k0 = 0; A.kmax = -1; A.tab = 0;
  // pixel ranges the fills can touch, in (major, minor); the minor range grows with the steps inside the target
  float mlo = fminf(ypxl1, ypxl2), mhi = fmaxf(ypxl1 + 1, ypxl2 + 1);   // (NaN operands drop out: a NaN end point fills nothing)
  if (first <= last) {
    if (!(first > -16777216.0f && last < 16777216.0f)) atomicOr(&status[FS_FLAGS], kFlagLineUnwalkable);
    else {
      // (first, last: integer-valued, |.| < 2^24; the differences need 26 bits: double)
      const double ext = (double)(steep ? P.h : P.w);                     // major extent of the target, pixels
      const double ka = first < 0.0f ? -(double)first : 0.0;              // first step with major >= 0
      const double kb = fmin((double)last - (double)first, ext - 1.0 - (double)first);   // last step with major < ext
      if (ka <= kb) {
        A.first = (int32_t)first; A.k0 = (int32_t)ka; A.kmax = (int32_t)kb;
        const uint32_t n = (uint32_t)(A.kmax - A.k0) + 1u;
        if (gradient == 0.0f) {
          // intery + 0 == intery at every step (a -0 start becomes +0 after the first addition: the same pixel, the same fractional
          // part +0): no chain to run, no table to fill - the four 1000-step outline lines of BASELINE configs[1] were 8 of setup's 14 us
          A.flat = 1u; A.flat_y = intery;
          mlo = fminf(mlo, wu_ipart(intery)); mhi = fmaxf(mhi, wu_ipart(intery) + 1);
        } else {
        A.tab = atomicAdd(&status[FS_TABLE_NEED], n);
        const bool fits = (uint64_t)A.tab + n <= (uint64_t)table_cap;
        // the reference's serial chain, step by step: first the steps left of / above the target (nothing to keep), then the
        // steps inside it, four table entries per store
        for (int32_t k = 0; k < A.k0; k++) intery += gradient;
        const float ilo = intery;
        float ihi = intery;
        float* __restrict__ dst = table + A.tab;
        uint32_t j = 0;
        if (fits) {
          for (; j + 4u <= n; j += 4u) {
            const float a = intery, b = a + gradient, c = b + gradient, d = c + gradient;
            dst[j] = a; dst[j + 1] = b; dst[j + 2] = c; dst[j + 3] = d;
            ihi = d;
            intery = d + gradient;
          }
        }
        for (; j < n; j++) {
          if (fits) dst[j] = intery;
          ihi = intery;
          intery += gradient;
        }
        // (x + g is monotone in the number of steps: the chain's extremes are its ends)
        mlo = fminf(mlo, wu_ipart(fminf(ilo, ihi))); mhi = fmaxf(mhi, wu_ipart(fmaxf(ilo, ihi)) + 1);
        if (!fits) A.kmax = A.k0 - 1;
        }
      }
    }
  }
  *out = A;
  // bounding box: major [xpxl1, xpxl2] (the steps lie between the end points), minor [mlo, mhi]; pixels -> samples
  const float Mlo = fminf(xpxl1, xpxl2), Mhi = fmaxf(xpxl1, xpxl2);
  int4 bb = make_int4(1, 1, 0, 0);
  if (Mlo == Mlo && mlo == mlo) {
    const double sr = (double)P.sr;
    double xlo = (steep ? (double)mlo : (double)Mlo) * sr, xhi = (steep ? (double)mhi : (double)Mhi) * sr + (sr - 1.0);
    double ylo = (steep ? (double)Mlo : (double)mlo) * sr, yhi = (steep ? (double)Mhi : (double)mhi) * sr + (sr - 1.0);
    const double wx = (double)(P.ssw - 1), wy = (double)(P.ssh - 1);
    if (xhi >= 0.0 && xlo <= wx && yhi >= 0.0 && ylo <= wy) {
      bb.x = (int)(xlo < 0.0 ? 0.0 : xlo); bb.y = (int)(ylo < 0.0 ? 0.0 : ylo);
      bb.z = (int)(xhi > wx ? wx : xhi);   bb.w = (int)(yhi > wy ? wy : yhi);
    }
  }
  return bb;
}

__global__ void raster_setup(RasterParams P, const srt_prim* __restrict__ prims, const ImageAux* __restrict__ aux,
                             int4* __restrict__ bbox, unsigned long long* __restrict__ stats, LineAux* __restrict__ laux,
                             float* __restrict__ ltable, uint32_t ltable_cap, uint32_t* __restrict__ status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.nprims) return;
  const srt_prim p = prims[i];
  if (p.kind == SRT_PRIM_LINE) {   // `reserved` of the device copy = ordinal of the record among the stream's lines
    bbox[i] = setup_line(P, p, laux + p.reserved, ltable, ltable_cap, status);
    return;
  }
  int4 bb = make_int4(1, 1, 0, 0);
  double lox, hix, loy, hiy;  // inclusive integer-valued ranges the reference loops over
  bool ok = false;
  if (p.kind == SRT_PRIM_TRIANGLE) {
    const float x0 = p.v.tri[0], y0 = p.v.tri[1], x1 = p.v.tri[2], y1 = p.v.tri[3], x2 = p.v.tri[4],
                y2 = p.v.tri[5];
    float xmin = floorf(std_min(x0, std_min(x1, x2)));
    float ymin = floorf(std_min(y0, std_min(y1, y2)));
    float xmax = ceilf(std_max(x0, std_max(x1, x2)));
    float ymax = ceilf(std_max(y0, std_max(y1, y2)));
    const float srf = (float)P.sr;
    xmin *= srf; xmax *= srf; ymin *= srf; ymax *= srf;  // float *= size_t (cpp:504)
    lox = xmin; hix = xmax; loy = ymin; hiy = ymax;
    ok = (lox <= hix) && (loy <= hiy);  // false for NaN
    if (ok && stats) {
      const double nx = hix - lox + 1.0, ny = hiy - loy + 1.0;
      if (nx * ny < 1.8e19) atomicAdd(&stats[ST_TESTS_REF], (unsigned long long)(nx * ny));
      // ... and the part of them inside the sample grid (what the tile kernel evaluates, or proves empty without evaluating)
      const double wx = (double)(P.ssw - 1), wy = (double)(P.ssh - 1);
      if (hix >= 0.0 && lox <= wx && hiy >= 0.0 && loy <= wy) {
        const double cx = (hix > wx ? wx : hix) - (lox < 0.0 ? 0.0 : lox) + 1.0, cy = (hiy > wy ? wy : hiy) - (loy < 0.0 ? 0.0 : loy) + 1.0;
        atomicAdd(&stats[ST_TESTS_TARGET], (unsigned long long)(cx * cy));
      }
    }
  } else if (p.kind == SRT_PRIM_POINT) {
    // fill_sample((int)(x*sr + i), (int)(y*sr + j)) for i,j in [0,sr)   (cpp:296-300)
    const double fx = p.v.point[0] * (double)P.sr, fy = p.v.point[1] * (double)P.sr;
    const double lim = 2147483000.0;  // outside int range x86 yields INT_MIN -> rejected by the bounds check
    ok = (fx > -lim) && (fx < lim) && (fy > -lim) && (fy < lim);
    lox = trunc(fx); hix = trunc(fx + (double)(P.sr - 1));
    loy = trunc(fy); hiy = trunc(fy + (double)(P.sr - 1));
  }
  if (p.kind == SRT_PRIM_IMAGE) {   // `reserved` of the device copy = index of the record's ImageAux
    const ImageAux a = aux[p.reserved];
    bbox[i] = make_int4(a.bx0, a.by0, a.bx1, a.by1);
    return;
  }
  if (ok) {
    const double wx = (double)(P.ssw - 1), wy = (double)(P.ssh - 1);
    if (hix >= 0.0 && lox <= wx && hiy >= 0.0 && loy <= wy) {
      bb.x = (int)(lox < 0.0 ? 0.0 : lox);
      bb.y = (int)(loy < 0.0 ? 0.0 : loy);
      bb.z = (int)(hix > wx ? wx : hix);
      bb.w = (int)(hiy > wy ? wy : hiy);
    }
  }
  bbox[i] = bb;
}

// ---------------------------------------------------------------------------------------------
// Pass 1b: ordered binning in two levels.  A block walks a candidate list IN ORDER, 1024 bounding boxes per step, and
// appends the indices of those that overlap its rectangle with a block-wide ORDERED compaction (ballot + prefix over the
// waves), so every output list is sorted by stream position and painter's order survives.
//   level 1: one block per SUPER-bin (<= 8 x 8 of them), candidates = the whole stream -> super lists
//   level 2: one block per coarse bin (coarse_tiles^2 tiles), candidates = its super-bin's list -> the lists the tiles scan
// Work is (super-bins x primitives) + (bins x their super list) instead of (bins x primitives); the bin lists are packed
// back to back at offsets that follow from the super counts (bin b of super-bin s gets room for s's whole list), so their
// storage follows the entries the frame really has instead of bins x primitives.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bins_of_super(const RasterParams& P, uint32_t sb) {
  const uint32_t sx = sb % P.super_x, sy = sb / P.super_x;
  const uint32_t bw = min(P.super_bx, P.coarse_x - sx * P.super_bx), bh = min(P.super_by, P.coarse_y - sy * P.super_by);
  return bw * bh;
}

// Does a triangle cover NO sample of the rectangle [x0, x1] x [y0, y1] (sample indices, inside the target)?  Each edge function, as
// the reference rounds it - c = (float)(fl64(e.x * fl64(p.y - v.y)) - fl64(e.y * fl64(p.x - v.x))) with p = index / sample_rate in
// fp64 - is monotone in the sample's column and in its row (every rounding is a non-decreasing map), so over the rectangle it lies
// between its values at the four corners.  If every edge keeps one strict sign there, well away from zero (inside_triangle's
// sign products c_i * c_j then cannot underflow to a zero that passes both its tests), and the three signs are not all equal, no
// sample of the rectangle is covered.  Conservative: anything doubtful (an edge that changes sign or comes near zero, NaN)
// answers false.  Same expressions, same order as the tile kernel's row loop.
__device__ __forceinline__ bool triangle_misses_corners(const float* t, const double px[2], const double py[2]) {
  const double ax = (double)t[0], ay = (double)t[1], bx = (double)t[2], by = (double)t[3];
  const double cx = (double)t[4], cy = (double)t[5];
  const double ex[3] = {bx - ax, cx - bx, ax - cx}, ey[3] = {by - ay, cy - by, ay - cy};
  const double vx[3] = {ax, bx, cx}, vy[3] = {ay, by, cy};
  bool all_pos = true, all_neg = true, decided = true;
#pragma unroll
  for (int e = 0; e < 3; e++) {
    bool pos = true, neg = true;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const double k = ey[e] * (px[i] - vx[e]);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const float c = (float)(ex[e] * (py[j] - vy[e]) - k);
        pos = pos && c > 1e-18f;
        neg = neg && c < -1e-18f;
      }
    }
    decided = decided && (pos || neg);
    all_pos = all_pos && pos; all_neg = all_neg && neg;
  }
  return decided && !all_pos && !all_neg;
}
// ... for the rectangle of samples [x0, x1] x [y0, y1] (sample coordinates of the target)
__device__ __forceinline__ bool triangle_misses_rect(const srt_prim& p, double sr, int x0, int y0, int x1, int y1) {
  const double px[2] = {(double)x0 / sr, (double)x1 / sr}, py[2] = {(double)y0 / sr, (double)y1 / sr};
  return triangle_misses_corners(p.v.tri, px, py);
}

#ifndef SRT_BIN_K1
#define SRT_BIN_K1 24
#endif
#ifndef SRT_BIN_K2
#define SRT_BIN_K2 8
#endif
template <int LEVEL>
__global__ __launch_bounds__(1024) void raster_bin_pass(RasterParams P, const int4* __restrict__ bbox, const uint32_t* __restrict__ in_lists,
                                                         const uint32_t* __restrict__ in_counts, uint32_t* __restrict__ out_lists,
                                                         uint32_t* __restrict__ out_counts, uint32_t* __restrict__ offs,
                                                         uint32_t list_cap, uint32_t* __restrict__ status, const srt_prim* __restrict__ prims) {
  __shared__ uint32_t s_base;
  __shared__ uint32_t s_fits;
  bool fits = true;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  int x0, y0, x1, y1;
  const uint32_t* __restrict__ in = nullptr;
  uint32_t ncand;
  uint32_t* out;
  if (LEVEL == 1) {
    const uint32_t sb = blockIdx.x, sx = sb % P.super_x, sy = sb / P.super_x;
    const int span = (int)(P.super_bx * P.coarse_tiles * P.tile_s), span_y = (int)(P.super_by * P.coarse_tiles * P.tile_sy);
    x0 = (int)sx * span; y0 = (int)sy * span_y; x1 = x0 + span - 1; y1 = y0 + span_y - 1;
    ncand = P.nprims;
    out = out_lists + (size_t)sb * P.super_stride;
  } else {
    const uint32_t bin = blockIdx.x, cx = bin % P.coarse_x, cy = bin / P.coarse_x;
    const int span = (int)(P.coarse_tiles * P.tile_s), span_y = (int)(P.coarse_tiles * P.tile_sy);
    x0 = (int)cx * span; y0 = (int)cy * span_y; x1 = x0 + span - 1; y1 = y0 + span_y - 1;
    const uint32_t sxi = cx / P.super_bx, syi = cy / P.super_by, sb = syi * P.super_x + sxi;
    in = in_lists + (size_t)sb * P.super_stride;
    ncand = in_counts[sb];
    // where this bin's list starts: room for every earlier super-bin's bins, then this bin's rank inside its super-bin
    if (threadIdx.x < 64) {
      unsigned long long part = 0;
      for (uint32_t k = (uint32_t)lane; k < sb; k += 64u) part += (unsigned long long)in_counts[k] * bins_of_super(P, k);
      for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
      if (lane == 0) {
        const uint32_t bw = min(P.super_bx, P.coarse_x - sxi * P.super_bx);
        const uint32_t local = (cy - syi * P.super_by) * bw + (cx - sxi * P.super_bx);
        const unsigned long long base = part + (unsigned long long)local * ncand, end = base + ncand;
        // The lists' storage is the caller's guess (it is kept across frames): a list that would not fit is left empty and the
        // room the frame needs is reported - the host grows the storage and repeats the frame (status[FS_LIST_NEED])
        s_fits = end <= (unsigned long long)list_cap ? 1u : 0u;
        if (!s_fits) atomicMax(&status[FS_LIST_NEED], end > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)end);
        s_base = s_fits ? (uint32_t)base : 0u;
      }
    }
    __syncthreads();
    out = out_lists + 2 * (size_t)s_base;                 // (level 2 writes two words per entry: the index and the packed box)
    if (threadIdx.x == 0) offs[bin] = s_base;
    fits = s_fits != 0u;
  }
  // K candidates per thread and step, candidate j * blockDim + thread of the step (coalesced loads; a thread's K loads are
  // independent and in flight together).  A step's order is (j, wave, lane): the hits of (j, wave) are a ballot mask kept in
  // LDS, one wave turns the K x 16 mask populations into exclusive offsets, and a thread with a hit needs two LDS reads to
  // know where it goes.  (The first two-level version had every thread add up sixteen wave counts per candidate and took
  // three steps for cfg2's stream: 17.6 us at level 1; K consecutive candidates per thread - 64 lines per load - cost the same.)
  constexpr int K = (LEVEL == 1) ? SRT_BIN_K1 : SRT_BIN_K2;
  constexpr int NM = K * 16;                            // masks per step
  __shared__ unsigned long long s_mask[NM];
  __shared__ uint32_t s_pref[NM + 1];
  uint32_t total = 0;
  if (!fits) ncand = 0;                                  // (a list without room stays empty)
  for (uint32_t base = 0; base < ncand; base += blockDim.x * K) {
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
      const uint32_t k = base + (uint32_t)j * blockDim.x + threadIdx.x;
      int4 bb = make_int4(1, 1, 0, 0);
      uint32_t pi = 0;
      if (k < ncand) { pi = (LEVEL == 1) ? k : in[k]; bb = bbox[pi]; }
      bool ov = (bb.x <= bb.z) && (bb.x <= x1) && (bb.z >= x0) && (bb.y <= y1) && (bb.w >= y0);

      if (LEVEL == 2 && ov) {
        // a triangle whose box reaches into this bin but which covers none of the bin's samples is not listed: a sliver's box is
        // most of the target, its samples a thin band (the stress frame: 11.7 M (primitive, tile) pairs listed by boxes alone)
        const srt_prim p = prims[pi];
        if (p.kind == SRT_PRIM_TRIANGLE && triangle_misses_rect(p, (double)P.sr, max(bb.x, x0), max(bb.y, y0), min(bb.z, x1), min(bb.w, y1))) ov = false;
      }
      const unsigned long long m = __ballot(ov);
      if (lane == 0) s_mask[j * 16 + wave] = (wave < nwaves) ? m : 0ull;
      mine |= (ov ? 1u : 0u) << j;
    }
    if (nwaves < 16 && threadIdx.x < 64)                 // (blocks of fewer than 16 waves: the other columns count nothing)
      for (int e = lane; e < NM; e += 64) if ((e & 15) >= nwaves) s_mask[e] = 0ull;
    __syncthreads();
    if (wave == 0) {
      // exclusive offsets of the NM masks in (j, wave) order: NM / 64 consecutive entries per lane, then a wave scan
      constexpr int PER = (NM + 63) / 64;
      uint32_t cnt[PER], sum = 0;
#pragma unroll
      for (int e = 0; e < PER; e++) { const int at = lane * PER + e; cnt[e] = at < NM ? (uint32_t)__popcll(s_mask[at]) : 0u; sum += cnt[e]; }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off); if (lane >= off) incl += t; }
      uint32_t run = incl - sum;
#pragma unroll
      for (int e = 0; e < PER; e++) { const int at = lane * PER + e; if (at < NM) s_pref[at] = run; run += cnt[e]; }
      if (lane == 63) s_pref[NM] = incl;
    }
    __syncthreads();
    uint32_t m = mine;
    while (m) {
      const uint32_t j = (uint32_t)__ffs((int)m) - 1u;
      m &= m - 1u;
      const uint32_t k = base + j * blockDim.x + threadIdx.x;
      const unsigned long long wm = s_mask[j * 16 + wave];
      const uint32_t at = total + s_pref[j * 16 + wave] + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
      if (LEVEL == 1) out[at] = k;
      else {
        // the entry: the primitive and its box clipped to this bin, bin-relative, 4 x 8 bits (read again from L1 / L2: keeping the
        // step's K boxes in registers cost the kernel three of its seven waves per SIMD)
        const uint32_t pi = in[k];
        const int4 bb = bbox[pi];
        const uint32_t box = (uint32_t)(max(bb.x, x0) - x0) | ((uint32_t)(max(bb.y, y0) - y0) << 8) | ((uint32_t)(min(bb.z, x1) - x0) << 16) |
                             ((uint32_t)(min(bb.w, y1) - y0) << 24);
        reinterpret_cast<uint2*>(out)[at] = make_uint2(pi, box);
      }
    }
    total += s_pref[NM];
    __syncthreads();                                     // (the next step rewrites the masks)
  }
  if (threadIdx.x == 0) out_counts[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------
// Texture sampling (texture.cpp).  Colors are float4 {r, g, b, a}.
// ---------------------------------------------------------------------------------------------
// CMU462::clamp<float> = std::min(std::max(x, lo), hi)
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return std_min(std_max(x, lo), hi); }

// GetColorFromTexture (texture.cpp:19-25): index 4 * (x + y * width) in size_t arithmetic; texels past the end of
// the level (undefined in the reference: one column / row past the level at the right / bottom border) read as zero.
__device__ __forceinline__ float4 get_texel(const uint8_t* __restrict__ level, uint32_t w, uint32_t h, int x, int y) {
  const unsigned long long idx = 4ull * ((unsigned long long)(long long)x + (unsigned long long)(long long)y * w);
  uint32_t t = 0;
  if (idx + 3 < 4ull * w * h) t = *reinterpret_cast<const uint32_t*>(level + idx);
  return make_float4((float)(t & 255u) / 255.0f, (float)((t >> 8) & 255u) / 255.0f, (float)((t >> 16) & 255u) / 255.0f,
                     (float)(t >> 24) / 255.0f);
}

// lerpColor<float> (texture.cpp:14-17): (1 - ratio) * start + ratio * ends
__device__ __forceinline__ float4 lerp_color(float ratio, float4 a, float4 b) {
  const float om = 1 - ratio;
  return make_float4(om * a.x + ratio * b.x, om * a.y + ratio * b.y, om * a.z + ratio * b.z, om * a.w + ratio * b.w);
}

// Sampler2DImp::sample_bilinear (texture.cpp:145-169) on one level.
__device__ __forceinline__ float4 sample_bilinear(const uint8_t* __restrict__ level, uint32_t w, uint32_t h, float u, float v) {
  const float wf = (float)w, hf = (float)h;
  const float su = clampf(u, 0.0f, 0.99999f) * wf;
  const float sv = clampf(v, 0.0f, 0.99999f) * hf;
  float u0 = floorf(su) + 0.5f, v0 = floorf(sv) + 0.5f, u1, v1;
  if (su - (float)(int)su < 0.5f) { u1 = u0; u0 = clampf(u0 - 1, 0.0f, wf); }   // clamp, then swap(u1, u0)
  else { u1 = clampf(u0 + 1, 0.0f, wf); }
  if (sv - (float)(int)sv < 0.5f) { v1 = v0; v0 = clampf(v0 - 1, 0.0f, hf); }
  else { v1 = clampf(v0 + 1, 0.0f, hf); }
  const float4 c00 = get_texel(level, w, h, (int)u0, (int)v0), c10 = get_texel(level, w, h, (int)u1, (int)v0);
  const float4 c01 = get_texel(level, w, h, (int)u0, (int)v1), c11 = get_texel(level, w, h, (int)u1, (int)v1);
  const float ru = (su - u0) / (u1 - u0), rv = (sv - v0) / (v1 - v0);
  return lerp_color(rv, lerp_color(ru, c00, c10), lerp_color(ru, c01, c11));
}

// Sampler2DImp::sample_trilinear (texture.cpp:171-193) with the level arithmetic already done (ImageAux).
__device__ __forceinline__ float4 sample_image(const ImageAux& A, const uint8_t* __restrict__ texels, float u, float v) {
  if (A.mode == 0) return make_float4(1.0f, 0.0f, 1.0f, 1.0f);
  const float4 lo = sample_bilinear(texels + A.off[0], A.w[0], A.h[0], u, v);
  if (A.mode == 1) return lo;
  const float4 hi = sample_bilinear(texels + A.off[1], A.w[1], A.h[1], u, v);
  return lerp_color(A.frac, lo, hi);
}

// ---------------------------------------------------------------------------------------------
// Pass 2: one wave per tile.
// ---------------------------------------------------------------------------------------------
// TSY: tile height in samples (16: 8 KiB of LDS per tile and twice the waves per CU - the kernel is bound by the
// latency of a wave's own instruction stream, not by throughput - and tighter culling; 32 for sample rates > 16).
// IMG: the build for streams with SRT_PRIM_IMAGE records.  The texture sampling (two bilinear look-ups, unrolled over the loop
// values that fold onto a sample) is two thirds of that build's code and weighs on its register allocation; frames without
// images - nearly all - run the build without it.
//
// The tile stays in LDS.  Round 3 measured the alternative the review asked for - the tile in registers, each lane owning the
// samples of its column in every other row as TSY / 2 float4 values with static indices, LDS touched only for the resolve -
// bit-exact on every fixture, and slower: cfg2 0.215 ms per frame against 0.178, the stress frame 8.9 ms against 6.9
// (DESIGN.md section 1).  These frames are bound by the instructions spent per (primitive, tile) pair and per row pair, not by
// the fills: the 1000 slivers of the stress frame cover 11.7 M samples of 5.9 G tested, and the unrolled row loops with their
// per-row range tests, the bookkeeping of which register holds which row and the 15-60 spilled registers cost more than the
// 2.5 M LDS instructions of a frame.
// WPB tiles per workgroup, one wavefront each (independent: a wave's tile, coordinates and barriers are its own).  ONE: measured in
// round 4 with four - 65 536 tiles of a 1024^2 x 16 frame as 16 384 workgroups - BASELINE configs[1]'s tile kernel went from 90 to
// 135 us and the stress frame from 1.29 to 1.65 ms: a workgroup keeps its LDS and wave slots until its LAST wave is done, and
// tiles differ widely in length (bit-exact either way; -DSRT_RASTER_WPB=4 rebuilds it).
#ifndef SRT_RASTER_WPB
#define SRT_RASTER_WPB 1
#endif
constexpr int kTileWavesPerBlock = SRT_RASTER_WPB;
// a barrier among the lanes of ONE wave: LDS operations of a wave execute in program order, so what is needed is that the
// compiler keeps them there
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
template <bool STATS, int TSY, bool IMG>
// (five waves per SIMD where the tile's LDS leaves room for them - 96 VGPRs, 7 spilled: 8-sample-high tiles at 20 waves per CU
//  measure 6 % faster on cfg2 than 16-high ones at 16; six and eight waves per SIMD lose to their spills)
#ifndef SRT_RASTER_OCC
#define SRT_RASTER_OCC 5
#endif
#ifndef SRT_RASTER_OCC_NOIMG
#define SRT_RASTER_OCC_NOIMG 6
#endif
// (the image-free build needs 79 VGPRs: six waves per SIMD without a spill - 24 waves per CU with 8-high tiles, whose 4 KiB of LDS
//  per wave leave room; 16-high tiles stay LDS-limited at 19.  Measured: five to eight waves per SIMD all give 0.164-0.168 ms per
//  cfg2 frame - the kernel is not short of waves.  Two tiles per wave in a loop: cfg2 0.143 -> 0.135 ms, the stress frame 6.65 -> 7.0 ms,
//  and the loop itself cost the one-tile case 7 %: not kept.)
__global__ __launch_bounds__(WAVE * (TSY == 32 ? 1 : SRT_RASTER_WPB), TSY == 32 ? 2 : ((IMG || TSY == 16) ? SRT_RASTER_OCC : SRT_RASTER_OCC_NOIMG)) void raster_tiles(RasterParams P, const srt_prim* __restrict__ prims,
                                                     const int4* __restrict__ bbox,
                                                     const uint32_t* __restrict__ lists,
                                                     const uint32_t* __restrict__ counts,
                                                     const uint32_t* __restrict__ offs,
                                                     const ImageAux* __restrict__ aux, const float* __restrict__ tabs,
                                                     const uint8_t* __restrict__ texels,
                                                     const LineAux* __restrict__ laux, const float* __restrict__ ltable,
                                                     uint32_t* __restrict__ rgba_out,
                                                     float4* __restrict__ samples_out,
                                                     unsigned long long* __restrict__ stats,
                                                     uint32_t* __restrict__ status, uint32_t* __restrict__ host_status) {
  constexpr int WPB = TSY == 32 ? 1 : kTileWavesPerBlock;
  // per wave: the tile's slice of super_sample_buffer (4 / 8 / 16 KiB), y / sample_rate for each tile row and x / sample_rate for each
  // tile column (fp64 divisions done once) - ONE block of LDS per wave, so that one (scalar) base serves all three
  struct TileLds { float4 tile[TS * TSY]; double rowy[TSY]; double colx[TS]; };
  __shared__ TileLds lds_all[WPB];
  // The frame's status words (what setup and binning needed, refusals) go to the host from here: one lane copies them into pinned
  // host memory and clears them for the next frame - a memset before and a copy after the frame were two more launches of ~5 us.
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < FS_COUNT; k++) { host_status[k] = status[k]; status[k] = 0u; }
  }

  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));   // (wave-uniform: kept in a scalar register)
  TileLds& L = lds_all[wv];
  float4* const tile = L.tile;
  double* const rowy = L.rowy;
  double* const colx = L.colx;
  const uint32_t tile_id = blockIdx.x * (uint32_t)WPB + (uint32_t)wv;
  if (tile_id >= P.tiles_x * P.tiles_y) return;           // (the last workgroup of a frame whose tile count is no multiple of WPB)
  const int tx = (int)(tile_id % P.tiles_x);
  const int ty = (int)(tile_id / P.tiles_x);
  const int sx0 = tx * (int)P.tile_s, sy0 = ty * (int)P.tile_sy;  // tile origin (samples)
  const int tsw = min((int)P.tile_s, (int)P.ssw - sx0);           // valid extent inside the target
  const int tsh = min((int)P.tile_sy, (int)P.ssh - sy0);
  const int sx1 = sx0 + tsw - 1, sy1 = sy0 + tsh - 1;

  // clear_target: every sample starts at 255.0f (software_renderer.h:93-98) - done when the first primitive reaches the tile
  // (`touched`, below).  A tile no primitive reaches - 40 % of BASELINE configs[1]'s tiles - never clears its LDS, computes no
  // coordinates and runs no box filter: sr^2 samples of 255.0f sum and divide back to exactly 255.0f, the pixel is 0xFFFFFFFF.
  const float4 white = make_float4(255.0f, 255.0f, 255.0f, 255.0f);
  bool touched = false, coords = false;

  const int lx = lane & (TS - 1);
  const int lrow = lane >> 5;
  // Pixel of a sample (lines compare pixels): the tile's origin is a whole pixel (tx * tile_px, ty * tile_py) and the offset inside
  // the tile is below 32, so floor(o / sr) = (int)((o + 0.5) * (1 / sr)) exactly - (o + 0.5) / sr is at least 1 / 64 away from
  // every integer, far more than the rounding of the reciprocal and the product - and no integer division is spent on it.
  const float rsr = 1.0f / (float)P.sr;
  const int tpx0 = tx * (int)P.tile_px, tpy0 = ty * (int)P.tile_py;           // pixel rectangle of the tile (uniform)
  const int tpx1 = tpx0 + (int)(((float)(tsw - 1) + 0.5f) * rsr), tpy1 = tpy0 + (int)(((float)(tsh - 1) + 0.5f) * rsr);
  const int colpix = tpx0 + (int)(((float)lx + 0.5f) * rsr);

  unsigned long long n_frags = 0, n_pts = 0, n_bins = 0;

  // this tile's coarse bin: an ordered list of primitive indices
  const uint32_t bin = (uint32_t)(ty / (int)P.coarse_tiles) * P.coarse_x + (uint32_t)(tx / (int)P.coarse_tiles);
  // A list entry is {primitive, its box clipped to the bin - bin-relative, 4 x 8 bits} (raster_bin_pass<2>): the scan needs no second,
  // dependent fetch of the box, and what a lane keeps of it across the ordered loop is ONE register - the rectangle inside this tile.
  // (Bins of more than 256 samples per side - targets beyond ~20 000 pixels - cannot pack: the box then comes from the bbox array.)
  const uint2* __restrict__ list = reinterpret_cast<const uint2*>(lists) + offs[bin];
  const uint32_t n = counts[bin];
  const int binx0 = (tx / (int)P.coarse_tiles) * (int)(P.coarse_tiles * P.tile_s), biny0 = (ty / (int)P.coarse_tiles) * (int)(P.coarse_tiles * P.tile_sy);
  // software prefetch: the next 64 entries are in flight while the current ones are rasterized
  uint2 nent = make_uint2(0u, 0u);
  if ((uint32_t)lane < n) nent = list[lane];
  for (uint32_t base = 0; base < n; base += WAVE) {
    const uint32_t myidx = nent.x;
    int bx0 = binx0 + (int)(nent.y & 255u), by0 = biny0 + (int)((nent.y >> 8) & 255u);
    int bx1 = binx0 + (int)((nent.y >> 16) & 255u), by1 = biny0 + (int)(nent.y >> 24);
    if (!P.packed_ok && base + lane < n) { const int4 fb = bbox[myidx]; bx0 = fb.x; by0 = fb.y; bx1 = fb.z; by1 = fb.w; }
    if (base + WAVE + lane < n) nent = list[base + WAVE + lane];
    const bool overlaps = (base + lane < n) && (bx0 <= sx1) && (bx1 >= sx0) && (by0 <= sy1) && (by1 >= sy0);
    // the primitive's rectangle inside the tile, tile-local sample coordinates (each < 32)
    const uint32_t rect = (uint32_t)(max(bx0, sx0) - sx0) | ((uint32_t)(max(by0, sy0) - sy0) << 8) | ((uint32_t)(min(bx1, sx1) - sx0) << 16) |
                          ((uint32_t)(min(by1, sy1) - sy0) << 24);
    unsigned long long mask = __ballot(overlaps);
    if (mask != 0ull && !coords) {                       // (uniform) the first candidate of this tile: the tile's sample coordinates
      // x / sample_rate, y / sample_rate (cpp:510).  For a power-of-two rate the quotient is the product with the exact reciprocal
      // (both are the exact value, correctly rounded: identical) and two fp64 divisions per lane are not spent on it.
      const bool pow2 = (P.sr & (P.sr - 1u)) == 0u;
      const double rs = 1.0 / (double)P.sr;
      if (lane < TSY) rowy[lane] = pow2 ? (double)(sy0 + lane) * rs : (double)(sy0 + lane) / (double)P.sr;
      if (lane < TS) colx[lane] = pow2 ? (double)(sx0 + lx) * rs : (double)(sx0 + lx) / (double)P.sr;
      coords = true;
      wave_sync();
    }
    // every overlapping lane fetches ITS primitive record now (three 16-byte loads in flight per lane); the ordered
    // loop below then reads records lane by lane with v_readlane instead of paying one memory round trip per primitive
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0;
    if (overlaps) {
      const uint4* __restrict__ pp = reinterpret_cast<const uint4*>(prims + myidx);
      q0 = pp[0]; q1 = pp[1]; q2 = pp[2];
    }
    // Triangles that cover no sample of their rectangle inside this tile leave the step here, all 64 candidates at once - every
    // lane tests ITS triangle at the rectangle's four corners (triangle_misses_corners: the reference's own edge functions, in the
    // row loop's expressions, on the row loop's coordinates) - instead of one by one in the ordered loop: 63 % of BASELINE
    // configs[1]'s (triangle, tile) pairs and 87 % of the stress frame's are such, ear-clipped slivers whose boxes cross the tile.
    bool misses = false;
    if (overlaps && q0.x == (uint32_t)SRT_PRIM_TRIANGLE) {
      const float t[6] = {__uint_as_float(q0.z), __uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w)};
      const double cpx[2] = {colx[rect & 255u], colx[(rect >> 16) & 255u]}, cpy[2] = {rowy[(rect >> 8) & 255u], rowy[rect >> 24]};
      misses = triangle_misses_corners(t, cpx, cpy);
    }
    mask &= ~__ballot(misses);
    if (mask != 0ull && !touched) {                      // (uniform) the first primitive that reaches this tile: clear_target
      for (int i = lane; i < TS * TSY; i += WAVE) tile[i] = white;
      touched = true;
      wave_sync();
    }

    while (mask) {  // ascending bit order == stream order
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      // srt_prim: {kind, reserved, v[6 floats | 2 doubles], rgba[4]}
      const uint32_t kind = __builtin_amdgcn_readlane(q0.x, b);
      const uint32_t w2 = __builtin_amdgcn_readlane(q0.z, b), w3 = __builtin_amdgcn_readlane(q0.w, b);
      const uint32_t w4 = __builtin_amdgcn_readlane(q1.x, b), w5 = __builtin_amdgcn_readlane(q1.y, b);
      const uint32_t w6 = __builtin_amdgcn_readlane(q1.z, b), w7 = __builtin_amdgcn_readlane(q1.w, b);
      // rectangle of this primitive inside the tile, tile-local sample coordinates
      const uint32_t rc = __builtin_amdgcn_readlane(rect, b);
      const int rx0 = (int)(rc & 255u), ry0 = (int)((rc >> 8) & 255u), rx1 = (int)((rc >> 16) & 255u), ry1 = (int)(rc >> 24);
      const float cr = __uint_as_float(__builtin_amdgcn_readlane(q2.x, b));
      const float cg = __uint_as_float(__builtin_amdgcn_readlane(q2.y, b));
      const float cb = __uint_as_float(__builtin_amdgcn_readlane(q2.z, b));
      const float one_minus_a = 1 - __uint_as_float(__builtin_amdgcn_readlane(q2.w, b));
      if (STATS) n_bins++;

      if (kind == SRT_PRIM_TRIANGLE) {
        const double ax = (double)__uint_as_float(w2), ay = (double)__uint_as_float(w3);
        const double bx = (double)__uint_as_float(w4), by = (double)__uint_as_float(w5);
        const double cx = (double)__uint_as_float(w6), cy = (double)__uint_as_float(w7);
        const double e0x = bx - ax, e0y = by - ay;  // t0t1
        const double e1x = cx - bx, e1y = cy - by;  // t1t2
        const double e2x = ax - cx, e2y = ay - cy;  // t2t0
        const double px = colx[lx];                 // (from LDS where it is needed: two registers fewer across the whole scan)
        const double d0x = px - ax, d1x = px - bx, d2x = px - cx;
        // e.y * (p.x - v.x) does not depend on the row: one fp64 product per edge per primitive instead of per sample
        const double k0 = e0y * d0x, k1 = e1y * d1x, k2 = e2y * d2x;
        const bool xin = (lx >= rx0) && (lx <= rx1);

        for (int row = ry0 + lrow; row <= ry1; row += 2) {
          const double py = rowy[row];
          const double d0y = py - ay, d1y = py - by, d2y = py - cy;
          const float c1 = (float)(e0x * d0y - k0);
          const float c2 = (float)(e1x * d1y - k1);
          const float c3 = (float)(e2x * d2y - k2);
          const float p12 = c1 * c2, p23 = c2 * c3, p13 = c1 * c3;
          const bool ccw = (p12 >= 0) && (p23 >= 0) && (p13 >= 0);
          const bool cw = (p12 <= 0) && (p23 <= 0) && (p13 <= 0);
          const bool covered = xin && (ccw || cw);
          if (covered) {
            const int si = row * TS + lx;
            tile[si] = blend_over(tile[si], cr, cg, cb, one_minus_a);
          }
          if (STATS) n_frags += __popcll(__ballot(covered));
        }
      } else if (kind == SRT_PRIM_LINE) {
        // rasterize_line_xiaolinwu's fills (raster_setup did the arithmetic), per sample: a fill is one PIXEL - rasterize_point's
        // sr x sr block at integer coordinates - so a sample takes it iff its pixel is the fill's.  For one sample the reference's
        // order is: first end point, second end point, main loop (the pixels of different stages can coincide on short lines;
        // within a stage they are distinct).
        const LineAux A = laux[__builtin_amdgcn_readlane(q0.y, b)];     // (uniform: scalar loads)
        const bool steep = A.steep != 0u;
        const int tM0 = steep ? tpy0 : tpx0, tM1 = steep ? tpy1 : tpx1, tm0 = steep ? tpx0 : tpy0, tm1 = steep ? tpx1 : tpy1;
        // which stages can touch this tile at all (uniform)
        bool st_end[2];
#pragma unroll
        for (int e = 0; e < 2; e++)
          st_end[e] = A.e_major[e] >= (float)tM0 && A.e_major[e] <= (float)tM1 && A.e_minor[e] + 1 >= (float)tm0 && A.e_minor[e] <= (float)tm1;   // (false for NaN)
        bool st_main = false;
        if (A.kmax >= A.k0) {
          // main loop: steps ka .. kb cross the tile's major range; intery is monotone in the step (x + g repeated), so the pixels
          // they fill lie between the two ends' values
          const int ka = max(tM0 - A.first, A.k0), kb = min(tM1 - A.first, A.kmax);
          if (ka <= kb) {
            const float ya = A.flat ? A.flat_y : ltable[A.tab + (uint32_t)(ka - A.k0)], yb = A.flat ? A.flat_y : ltable[A.tab + (uint32_t)(kb - A.k0)];
            const float ylo = floorf(fminf(ya, yb)), yhi = floorf(fmaxf(ya, yb)) + 1;
            st_main = !(yhi < (float)tm0) && !(ylo > (float)tm1);          // (NaN: stays in, fills nothing)
          }
        }
        if (st_end[0] || st_end[1] || st_main) {
          const bool xin = (lx >= rx0) && (lx <= rx1);
          for (int row = ry0 + lrow; row <= ry1; row += 2) {
            const int rp = tpy0 + (int)(((float)row + 0.5f) * rsr);          // pixel row
            const int Mi = steep ? rp : colpix;
            const float M = (float)Mi, m = (float)(steep ? colpix : rp);
            const int si = row * TS + lx;
#pragma unroll
            for (int e = 0; e < 2; e++) {
              if (st_end[e]) {
                const float em = A.e_minor[e];                                // rasterize_point(xpxl, ypxl), (xpxl, ypxl + 1)
                const bool ht = xin && M == A.e_major[e] && m == em, hb = xin && M == A.e_major[e] && m == em + 1;
                if (ht || hb) tile[si] = blend_over(tile[si], cr, cg, cb, 1 - (ht ? A.e_top[e] : A.e_bot[e]));   // fill_sample's (1 - c.a)
                if (STATS) n_pts += __popcll(__ballot(ht || hb));
              }
            }
            if (st_main) {
              const int k = Mi - A.first;
              const bool in_k = xin && k >= A.k0 && k <= A.kmax;
              float iy = A.flat_y;
              if (in_k && !A.flat) iy = ltable[A.tab + (uint32_t)(k - A.k0)];
              const float ip = floorf(iy), fp = iy - ip;                      // ipart, fpart
              const bool ht = in_k && m == ip, hb = in_k && m == ip + 1;      // rasterize_point(x, ipart), (x, ipart + 1)
              if (ht || hb) tile[si] = blend_over(tile[si], cr, cg, cb, ht ? 1 - (1 - fp) : 1 - fp);   // 1 - rfpart / 1 - fpart
              if (STATS) n_pts += __popcll(__ballot(ht || hb));
            }
          }
        }
      } else if (IMG && kind == SRT_PRIM_IMAGE) {
        // rasterize_image: every (x, y) pair of the float loops whose truncation lands on this lane's sample, x outer
        const ImageAux A = aux[__builtin_amdgcn_readlane(q0.y, b)];
        const int sx = sx0 + lx;
        const bool xin = (lx >= rx0) && (lx <= rx1);
        float xv[2] = {std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::quiet_NaN()};
        if (xin) { xv[0] = tabs[A.xtab + sx]; xv[1] = tabs[A.xtab + P.ssw + sx]; }
        const double xden = (double)(A.x1s - A.x0s), yden = (double)(A.y1s - A.y0s);
        for (int row = ry0 + lrow; row <= ry1; row += 2) {
          const int sy = sy0 + row;
          const float yv[2] = {tabs[A.ytab + sy], tabs[A.ytab + P.ssh + sy]};
          const int si = row * TS + lx;
#pragma unroll
          for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
              if (xv[i] == xv[i] && yv[j] == yv[j]) {   // not NaN
                const float u = (float)(((double)xv[i] + 0.5 - (double)A.x0s) / xden);
                const float v = (float)(((double)yv[j] + 0.5 - (double)A.y0s) / yden);
                const float4 c = sample_image(A, texels, u, v);
                tile[si] = blend_over(tile[si], c.x, c.y, c.z, 1 - c.w);
              }
            }
          }
        }
      } else if (kind == SRT_PRIM_POINT) {
        const uint32_t sr = P.sr;
        const double fx = __hiloint2double((int)w3, (int)w2) * (double)sr;   // v.point[0], v.point[1]
        const double fy = __hiloint2double((int)w5, (int)w4) * (double)sr;
        // Distinct (i,j) land on distinct samples unless truncation toward zero folds two of them
        // onto column/row 0 (fx+i in (-1,0)); then the block must be applied one sample at a time.
        const bool folds = (fx < 0.0 && fx != trunc(fx) && fx + (double)(sr - 1) > -1.0) ||
                           (fy < 0.0 && fy != trunc(fy) && fy + (double)(sr - 1) > -1.0);
        const uint32_t nblk = sr * sr;
        if (!folds) {
          for (uint32_t k0 = 0; k0 < nblk; k0 += WAVE) {
            const uint32_t k = k0 + lane;
            bool hit = false;
            int si = 0;
            if (k < nblk) {
              const int i = (int)(k / sr), j = (int)(k % sr);
              const int sx = (int)(fx + i) - sx0, sy = (int)(fy + j) - sy0;
              hit = (sx >= 0) && (sx < tsw) && (sy >= 0) && (sy < tsh);
              si = sy * TS + sx;
            }
            if (hit) tile[si] = blend_over(tile[si], cr, cg, cb, one_minus_a);
            if (STATS) n_pts += __popcll(__ballot(hit));
          }
        } else {
          for (uint32_t k = 0; k < nblk; k++) {  // i outer, j inner (cpp:296-297)
            const int i = (int)(k / sr), j = (int)(k % sr);
            const int sx = (int)(fx + i) - sx0, sy = (int)(fy + j) - sy0;
            const bool hit = (sx >= 0) && (sx < tsw) && (sy >= 0) && (sy < tsh);
            if (hit && lane == 0) {
              const int si = sy * TS + sx;
              tile[si] = blend_over(tile[si], cr, cg, cb, one_minus_a);
            }
            if (STATS && hit) n_pts++;
          }
        }
      }
      // no barrier: one wavefront owns the tile and its LDS operations execute in program order
    }
  }
  wave_sync();

  // resolve (cpp:586-619): box sum with x-offset outer / y-offset inner, true division, truncation
  const int sr = (int)P.sr;
  const int pw = tsw / sr, ph = tsh / sr;
  const int px0 = tx * (int)P.tile_px, py0 = ty * (int)P.tile_py;
  const float denom = (float)((size_t)P.sr * (size_t)P.sr);
  const bool pow2_sr = (P.sr & (P.sr - 1u)) == 0u;
  const float inv_denom = 1.0f / denom;
  if (!touched) {                                        // an untouched tile: every pixel (255, 255, 255, 255), every sample 255.0f
    for (int k = lane; k < pw * ph; k += WAVE) rgba_out[(size_t)(py0 + k / pw) * P.w + (px0 + k % pw)] = 0xFFFFFFFFu;
    if (samples_out)
      for (int k = lane; k < tsw * tsh; k += WAVE) samples_out[(size_t)(sy0 + k / tsw) * P.ssw + (sx0 + k % tsw)] = white;
    if (STATS && lane == 0) atomicAdd(&stats[ST_BIN_ENTRIES], n_bins);
    return;
  }
  for (int k = lane; k < pw * ph; k += WAVE) {
    const int pxl = k % pw, pyl = k / pw;
    float r = 0, g = 0, bl = 0, a = 0;
    for (int i = 0; i < sr; ++i) {
      for (int j = 0; j < sr; ++j) {
        const float4 s = tile[(pyl * sr + j) * TS + pxl * sr + i];
        r += s.x; g += s.y; bl += s.z; a += s.w;
      }
    }
    if (pow2_sr) { r *= inv_denom; g *= inv_denom; bl *= inv_denom; a *= inv_denom; }   // (exact scaling: the same value as the division)
    else { r /= denom; g /= denom; bl /= denom; a /= denom; }
    const uint32_t R = (uint32_t)(uint8_t)(r), G = (uint32_t)(uint8_t)(g), B = (uint32_t)(uint8_t)(bl),
                   A = (uint32_t)(uint8_t)(a);
    rgba_out[(size_t)(py0 + pyl) * P.w + (px0 + pxl)] = R | (G << 8) | (B << 16) | (A << 24);
  }

  if (samples_out) {
    for (int k = lane; k < tsw * tsh; k += WAVE) {
      const int c = k % tsw, rr = k / tsw;
      samples_out[(size_t)(sy0 + rr) * P.ssw + (sx0 + c)] = tile[rr * TS + c];
    }
  }

  if (STATS) {
    if (lane == 0) {
      atomicAdd(&stats[ST_FRAGMENTS], n_frags);
      atomicAdd(&stats[ST_POINT_SAMPLES], n_pts);
      atomicAdd(&stats[ST_BIN_ENTRIES], n_bins);
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct srt_raster {
  int device = 0;
  hipStream_t stream = nullptr;
  RasterParams P{};
  bool have_target = false;
  // The frame's ordered stream, in PINNED host memory (grown geometrically): srt_raster_submit appends here and the upload is
  // one DMA transfer, no staging copy.
  srt_prim* pending = nullptr; size_t pending_n = 0, pending_cap = 0;
  std::vector<srt_prim> uploaded;   // what d_prims holds (host copy): an identical resubmission uploads and bins nothing
  bool dirty = true;                // pending differs from d_prims
  hipEvent_t upload_done = nullptr; bool upload_pending = false;   // the DMA out of `pending` may still be reading it
  // `reserved` words of `pending` that hold device-side indices (a line's ordinal, an image's ImageAux) while that DMA runs, with
  // the caller's values: put back by settle_upload() once the copy has read them, before anything looks at `pending` again
  std::vector<std::pair<size_t, uint32_t>> patched;
  srt_prim* d_prims = nullptr;
  int4* d_bbox = nullptr;
  size_t d_cap = 0;
  uint2* d_lists = nullptr; size_t lists_cap = 0;      // coarse-bin lists, packed back to back (raster_bin_pass<2>), entries of {primitive, box}; kept across frames and streams
  uint32_t* d_counts = nullptr; size_t counts_cap = 0; // per coarse bin: entries, then (second half) list offsets
  uint32_t* d_super = nullptr; size_t super_cap = 0;   // super-bin lists (<= 64 x nprims) followed by their 64 counts
  bool bins_valid = false;                             // the bin lists on the device belong to the current stream / target / tiling
  bool verified = false;                               // ... and a frame with them has reported that every buffer was large enough
  uint32_t coarse_min = 4;                             // tiles per coarse-bin side to start from (raised when the lists get too long)
  // lines (SRT_PRIM_LINE): one LineAux per record, the intery table (raster_setup)
  uint32_t nlines = 0;
  bool has_images = false;                             // the stream on the device holds SRT_PRIM_IMAGE records (the tile kernel's IMG build)
  LineAux* d_laux = nullptr; size_t laux_cap = 0;
  float* d_ltable = nullptr; size_t ltable_cap = 0;
  uint32_t* d_status = nullptr;                        // FS_* words of the frame in flight
  uint32_t* h_status = nullptr;                        // ... written here (pinned, device-visible) by the tile kernel's first block
  uint32_t* d_host_status = nullptr;                   // its device address
  uint32_t* d_rgba = nullptr;
  float4* d_samples = nullptr;
  unsigned long long* d_stats = nullptr;
  bool resolved = false;
  uint8_t* bound_out = nullptr;                        // srt_raster_bind_output: the caller's framebuffer, pinned
  // textures (srt_raster_add_texture): host copies, then one device blob with 4-byte aligned levels
  struct Tex { uint32_t nlevels; uint32_t w[SRT_MAX_MIP_LEVELS], h[SRT_MAX_MIP_LEVELS]; size_t off[SRT_MAX_MIP_LEVELS]; };
  std::vector<Tex> textures, prev_textures;
  // The texel blob lives in PINNED host memory and is kept across srt_raster_clear_textures: DrawSVG's redraw clears and re-adds
  // the same mip chains every frame (drawsvg.cpp:462-474 builds them once per SVG), so a re-added level is COMPARED with what the
  // blob already holds at that offset instead of copied, and as long as every level matches the device copy stays as it is.
  uint8_t* texel_blob = nullptr; size_t blob_n = 0, blob_cap = 0;   // blob_n: bytes of the current texture set
  size_t blob_kept = 0;                                // bytes of the previous set still in the blob behind blob_n (re-add comparison)
  size_t device_blob_n = 0;                            // d_texels holds texel_blob[0 .. device_blob_n) byte for byte
  bool tex_dirty = true;                               // the texture set changed since the image tables (ImageAux) were built
  uint64_t texel_bytes_uploaded = 0;                   // total bytes of texel uploads (srt_raster_texture_upload_bytes)
  bool aux_valid = false;                              // d_aux / d_tabs belong to the stream on the device, this target and these textures
  bool rebuilding = false;                             // between srt_raster_clear_textures and the next frame: add_texture compares with prev_textures / the blob
  bool blob_upload_pending = false;                    // a DMA transfer may still be reading the pinned blob
  uint8_t* d_texels = nullptr; size_t texels_cap = 0;
  ImageAux* d_aux = nullptr; size_t aux_cap = 0;
  float* d_tabs = nullptr; size_t tabs_cap = 0;
};

namespace {

// Tile height for this frame: 8 sample rows (more waves per CU, 24 with the image-free build) whenever the sample rate allows it.
// Frames of large primitives used to be better off with 16-high tiles - half the (primitive, tile) pairs - while every pair cost
// a turn of the ordered loop; since the pairs that cover nothing are dropped 64 at a time in the list scan that no longer holds
// (stress SVG 1024^2 x 4: 1.40 ms with 8 rows, 1.54 with 16; BASELINE configs[1]: 0.116 / 0.139).  The image does not depend on it.
uint32_t choose_tile_height(const srt_raster* r) {
  const uint32_t sr = r->P.sr;
  if (sr > 16) return TS;
  if (sr > 8) return 16;
  if (const char* e = getenv("SRT_RASTER_TSY")) { const uint32_t v = (uint32_t)atoi(e); if ((v == 8 || v == 16 || v == 32) && sr <= v) return v; }   // experiments
  return 8;
}

template <typename T>
int grow(T** buf, size_t* cap, size_t need) {
  if (*cap >= need && *buf) return SRT_OK;
  if (*buf) SRT_HIP(hipFree(*buf));
  *buf = nullptr; *cap = 0;      // a failed allocation below must not leave a capacity behind
  SRT_HIP(hipMalloc(buf, need * sizeof(T)));
  *cap = need;
  return SRT_OK;
}

// Before `pending` is read or rewritten on the host: the upload out of it has finished, its records are the caller's again.
int settle_upload(srt_raster* r) {
  if (r->upload_pending) { SRT_HIP(hipEventSynchronize(r->upload_done)); r->upload_pending = false; }
  for (auto& pr : r->patched) r->pending[pr.first].reserved = pr.second;   // the host copy keeps texture ids / the caller's zeros
  r->patched.clear();
  return SRT_OK;
}

int upload_stream(srt_raster* r) {
  { const int st = settle_upload(r); if (st != SRT_OK) return st; }
  const size_t n = r->pending_n;
  if (n > 0xFFFFFFFFull) return srt::fail(SRT_ERR_UNSUPPORTED, "more than 2^32-1 primitives in one frame");
  {
    const uint32_t tsy = choose_tile_height(r);
    if (tsy / r->P.sr * r->P.sr != r->P.tile_sy) {
      r->P.tile_py = tsy / r->P.sr;
      r->P.tile_sy = r->P.tile_py * r->P.sr;
      r->P.tiles_y = (r->P.h + r->P.tile_py - 1) / r->P.tile_py;
      r->bins_valid = false;
    }
  }
  // the same stream again (DrawSVG redraws on every event): nothing to upload, and the bin lists on the device stay valid
  bool has_image = false;
  for (size_t i = 0; i < n && !has_image; i++) has_image = r->pending[i].kind == SRT_PRIM_IMAGE;
  // the texture set as srt_raster_clear_textures / srt_raster_add_texture left it: re-adding the levels the blob already held
  // changes nothing (add_texture compared them); fewer or more textures than before do
  if (r->rebuilding) { if (r->textures.size() != r->prev_textures.size()) r->tex_dirty = true; r->rebuilding = false; }
  r->blob_kept = 0;
  // A stream without images does not care about the textures (tex_dirty stays set for the next stream that does); one with
  // images is the same frame only if its tables - target, textures - are the ones on the device.
  if ((!has_image || (r->aux_valid && !r->tex_dirty)) && n == r->uploaded.size() && r->P.nprims == (uint32_t)n &&
      (n == 0 || std::memcmp(r->pending, r->uploaded.data(), n * sizeof(srt_prim)) == 0)) {
    r->dirty = false;
    return SRT_OK;
  }
  if (n > r->d_cap) {
    if (r->d_prims) SRT_HIP(hipFree(r->d_prims));
    if (r->d_bbox) SRT_HIP(hipFree(r->d_bbox));
    r->d_prims = nullptr; r->d_bbox = nullptr; r->d_cap = 0;
    size_t cap = n + n / 2 + 64;
    SRT_HIP(hipMalloc(&r->d_prims, cap * sizeof(srt_prim)));
    SRT_HIP(hipMalloc(&r->d_bbox, cap * sizeof(int4)));
    r->d_cap = cap;
  }
  // what d_prims is about to hold, as the caller wrote it (the next frame's stream is compared with this); dropped again on every
  // failure below - without it the next frame just uploads
  try { r->uploaded.assign(r->pending, r->pending + n); } catch (...) { r->uploaded.clear(); }
  // SRT_PRIM_IMAGE records: per-image constants and loop-value tables (ImageAux); SRT_PRIM_LINE records: their ordinal.
  // The device copy of such a record carries the index of its ImageAux / LineAux in `reserved`.
  std::vector<ImageAux> aux;
  std::vector<float> tabs;
  std::vector<std::pair<size_t, uint32_t>>& patched = r->patched;   // (record, original reserved); empty here (settle_upload)
  const RasterParams& P = r->P;
  const float qnan = std::numeric_limits<float>::quiet_NaN();
  uint32_t nlines = 0;
  for (size_t i = 0; i < n; i++) {
    srt_prim& p = r->pending[i];
    if (p.kind == SRT_PRIM_LINE) {
      patched.emplace_back(i, p.reserved);
      p.reserved = nlines++;
      continue;
    }
    if (p.kind != SRT_PRIM_IMAGE) continue;
    if (p.reserved >= r->textures.size()) {
      r->uploaded.clear(); (void)settle_upload(r);
      return srt::fail(SRT_ERR_INVALID, "primitive %zu refers to texture %u, %zu textures are loaded", i, p.reserved, r->textures.size());
    }
    const srt_raster::Tex& T = r->textures[p.reserved];
    ImageAux A;
    std::memset(&A, 0, sizeof A);
    float x0 = p.v.tri[0], y0 = p.v.tri[1], x1 = p.v.tri[2], y1 = p.v.tri[3];
    const float uscale = x1 - x0, vscale = y1 - y0;                // software_renderer.cpp:553
    x0 *= (float)P.sr; x1 *= (float)P.sr; y0 *= (float)P.sr; y1 *= (float)P.sr;   // :556 (float *= size_t)
    A.x0s = x0; A.y0s = y0; A.x1s = x1; A.y1s = y1;
    {   // sample_trilinear's level (texture.cpp:179-192); float / double steps as the reference's expressions resolve
      const double ax = (double)((float)T.w[0] / uscale), bx = (double)((float)T.h[0] / uscale);
      const double ay = (double)((float)T.w[0] / vscale), by = (double)((float)T.h[0] / vscale);
      const float lsx = (float)(std::pow(ax, 2) + std::pow(bx, 2));
      const float lsy = (float)(std::pow(ay, 2) + std::pow(by, 2));
      float level = log2f(std::sqrt(lsx < lsy ? lsy : lsx));       // std::max(a, b) = (a < b) ? b : a
      if (level < 0) level = 0.0f;
      if (level >= (float)T.nlevels) { A.mode = 0; }
      else {
        const int lo = (int)std::floor(level), hi = lo + 1;
        if (hi >= (int)T.nlevels) { A.mode = 1; A.low = (int)T.nlevels - 1; }
        else { A.mode = 2; A.low = lo; A.frac = level - (float)(int)level; }
      }
    }
    for (int k = 0; k < 2; k++) {
      const int lv = A.low + k < (int)T.nlevels ? A.low + k : A.low;
      A.off[k] = (uint32_t)T.off[lv]; A.w[k] = T.w[lv]; A.h[k] = T.h[lv];
    }
    // the float loops of rasterize_image (:557-559), recorded per target column / row
    A.xtab = (uint32_t)tabs.size();
    tabs.resize(tabs.size() + 2 * (size_t)P.ssw, qnan);
    A.ytab = (uint32_t)tabs.size();
    tabs.resize(tabs.size() + 2 * (size_t)P.ssh, qnan);
    int rx = 0, ry = 0;
    if (tabs.size() <= 0xFFFFFFFFull) {
      A.bx0 = A.by0 = 1; A.bx1 = A.by1 = 0;
      auto walk = [&](float lo, float hi, uint32_t extent, uint32_t tab, int32_t& b0, int32_t& b1) -> int {
        bool any = false;
        uint64_t guard = 0;
        for (float x = lo; x <= hi; x++) {
          if (x + 1 == x || ++guard > (1ull << 26)) return -1;       // the reference's loop would never end
          if (!(x > -2147483648.0f && x < 2147483648.0f)) continue;  // (int)x is INT_MIN on x86: rejected by fill_sample
          const int sx = (int)x;
          if (sx < 0 || (uint32_t)sx >= extent) continue;
          float* e = &tabs[tab + sx];
          if (e[0] != e[0]) e[0] = x;
          else if (e[extent] != e[extent]) e[extent] = x;
          else return -2;
          if (!any) { b0 = b1 = sx; any = true; }
          b0 = sx < b0 ? sx : b0; b1 = sx > b1 ? sx : b1;
        }
        return any ? 1 : 0;
      };
      rx = walk(x0, x1, P.ssw, A.xtab, A.bx0, A.bx1);
      ry = walk(y0, y1, P.ssh, A.ytab, A.by0, A.by1);
    }
    if (tabs.size() > 0xFFFFFFFFull || rx < 0 || ry < 0) {
      r->uploaded.clear(); (void)settle_upload(r);
      if (tabs.size() > 0xFFFFFFFFull) return srt::fail(SRT_ERR_UNSUPPORTED, "image tables exceed 2^32 entries");
      return srt::fail(SRT_ERR_UNSUPPORTED, "image primitive %zu: extent (%g, %g)-(%g, %g) samples is outside what the reference's float loops can walk",
                       i, (double)x0, (double)y0, (double)x1, (double)y1);
    }
    if (rx == 0 || ry == 0) { A.bx0 = A.by0 = 1; A.bx1 = A.by1 = 0; }
    patched.emplace_back(i, p.reserved);
    p.reserved = (uint32_t)aux.size();
    aux.push_back(A);
  }
  // (an intery table never holds more than one entry per line and pixel column / row of the target)
  const bool lines_fit = (uint64_t)nlines * std::max(P.w, P.h) < (1ull << 32);
  // one DMA transfer out of the pinned buffer; nobody waits for it here - the patched `reserved` words stay as they are until the
  // host next touches `pending` (settle_upload; the wait per frame for this copy was 9 us of BASELINE configs[1]'s 0.32 ms redraw)
  hipError_t up = hipSuccess;
  if (n && lines_fit) {
    up = hipMemcpyAsync(r->d_prims, r->pending, n * sizeof(srt_prim), hipMemcpyHostToDevice, r->stream);
    if (up == hipSuccess) { up = hipEventRecord(r->upload_done, r->stream); r->upload_pending = up == hipSuccess; }
    if (up != hipSuccess) (void)hipStreamSynchronize(r->stream);       // (an event that could not be recorded: wait the plain way)
  }
  if (up != hipSuccess || !lines_fit) { r->uploaded.clear(); (void)settle_upload(r); }
  SRT_HIP(up);
  if (!lines_fit) return srt::fail(SRT_ERR_UNSUPPORTED, "%u lines on a %u x %u target: their tables could exceed 2^32 entries", nlines, P.w, P.h);
  r->nlines = nlines;
  r->has_images = !aux.empty();
  if (nlines) {
    int st = grow(&r->d_laux, &r->laux_cap, (size_t)nlines + nlines / 2);
    if (st != SRT_OK) return st;
  }
  if (!aux.empty()) {
    int st;
    if ((st = grow(&r->d_aux, &r->aux_cap, aux.size())) != SRT_OK || (st = grow(&r->d_tabs, &r->tabs_cap, tabs.size())) != SRT_OK) return st;
    SRT_HIP(hipMemcpy(r->d_aux, aux.data(), aux.size() * sizeof(ImageAux), hipMemcpyHostToDevice));
    SRT_HIP(hipMemcpy(r->d_tabs, tabs.data(), tabs.size() * sizeof(float), hipMemcpyHostToDevice));
    // the texels: only what the device copy does not hold yet (nothing at all when the frame re-added the textures of the last one),
    // one DMA transfer out of the pinned blob, not waited for (add_texture waits before it rewrites the blob)
    if (r->texels_cap < r->blob_n || !r->d_texels) {
      if ((st = grow(&r->d_texels, &r->texels_cap, r->blob_n ? r->blob_n + r->blob_n / 4 : 4)) != SRT_OK) return st;
      r->device_blob_n = 0;
    }
    if (r->device_blob_n < r->blob_n) {
      SRT_HIP(hipMemcpyAsync(r->d_texels + r->device_blob_n, r->texel_blob + r->device_blob_n, r->blob_n - r->device_blob_n, hipMemcpyHostToDevice, r->stream));
      r->texel_bytes_uploaded += r->blob_n - r->device_blob_n;
      r->device_blob_n = r->blob_n;
      r->blob_upload_pending = true;
    }
    r->tex_dirty = false;                          // (only here: a stream without images leaves the flag for the next one with images)
  }
  r->aux_valid = !aux.empty();
  r->P.nprims = (uint32_t)n;
  r->dirty = false;
  r->bins_valid = false;                         // another stream: setup and binning run on its first frame
  r->verified = false;
  return SRT_OK;
}

// The binning grid for `c` tiles per coarse-bin side.
void set_grid(RasterParams& P, uint32_t c) {
  P.coarse_tiles = c;
  P.coarse_x = (P.tiles_x + c - 1) / c;
  P.coarse_y = (P.tiles_y + c - 1) / c;
  P.super_bx = (P.coarse_x + 7) / 8; P.super_by = (P.coarse_y + 7) / 8;
  P.super_x = (P.coarse_x + P.super_bx - 1) / P.super_bx; P.super_y = (P.coarse_y + P.super_by - 1) / P.super_by;
  P.super_stride = P.nprims ? P.nprims : 1;
  P.packed_ok = (c * P.tile_s <= 256u && c * P.tile_sy <= 256u) ? 1u : 0u;
}

// Enqueue one frame of the current stream on `s`: raster_setup + the two ordered binning passes when the lists on the device
// do not belong to this stream / target yet, then the tile kernel; the frame's status words follow into pinned host memory.
// Nothing here waits for the device.  The line tables and the packed bin lists live in storage that is kept across frames and
// sized by what earlier frames needed (a first guess for the very first one): a frame whose needs exceed it leaves the
// affected lists / tables empty and says so in its status - check_frame() then grows the storage and the caller repeats the frame.
int launch_frame(srt_raster* r, hipStream_t s, bool dump_samples, bool stats) {
  RasterParams& P = r->P;
  if (stats) SRT_HIP(hipMemsetAsync(r->d_stats, 0, ST_COUNT * sizeof(unsigned long long), s));
  if (!r->bins_valid || stats) {
    uint32_t c = r->coarse_min;
    for (;; c *= 2) {                                                             // (keeps the bin count bounded on huge targets)
      set_grid(P, c);
      if ((uint64_t)P.coarse_x * P.coarse_y > 8192 && c < 65536) continue;
      break;
    }
    const size_t nb = (size_t)P.coarse_x * P.coarse_y, ns = (size_t)P.super_x * P.super_y;
    int st;
    if ((st = grow(&r->d_counts, &r->counts_cap, 2 * nb)) != SRT_OK) return st;
    if ((st = grow(&r->d_super, &r->super_cap, ns * P.super_stride + 64)) != SRT_OK) return st;
    if (!r->d_lists) {       // first guess: a primitive in most bins of ONE super-bin (what small primitives come to); check_frame() corrects it
      const size_t guess = (size_t)P.nprims * P.super_bx * P.super_by * 3 / 4 + 65536;
      if ((st = grow(&r->d_lists, &r->lists_cap, std::min<size_t>(guess, 1ull << 28))) != SRT_OK) return st;
    }
    if (r->nlines && !r->d_ltable) {
      if ((st = grow(&r->d_ltable, &r->ltable_cap, (size_t)r->nlines * 64 + 4096)) != SRT_OK) return st;
    }
    if (P.nprims) {
      const int bs = 256;
      raster_setup<<<dim3((P.nprims + bs - 1) / bs), dim3(bs), 0, s>>>(P, r->d_prims, r->d_aux, r->d_bbox, stats ? r->d_stats : nullptr,
                                                                      r->d_laux, r->d_ltable, (uint32_t)std::min<size_t>(r->ltable_cap, 0xFFFFFFFFull), r->d_status);
    }
    // Ordered binning (raster_bin_pass): coarse bins of c x c tiles under <= 8 x 8 super-bins
    uint32_t* d_super_counts = r->d_super + ns * P.super_stride;
    raster_bin_pass<1><<<dim3((unsigned)ns), dim3(1024), 0, s>>>(P, r->d_bbox, nullptr, nullptr, r->d_super, d_super_counts, nullptr, 0u, r->d_status, r->d_prims);
    raster_bin_pass<2><<<dim3((unsigned)nb), dim3(256), 0, s>>>(P, r->d_bbox, r->d_super, d_super_counts, reinterpret_cast<uint32_t*>(r->d_lists), r->d_counts, r->d_counts + nb,
                                                                (uint32_t)std::min<size_t>(r->lists_cap, 0xFFFFFFFFull), r->d_status, r->d_prims);
    r->bins_valid = true;
  }
  const uint32_t ntiles = P.tiles_x * P.tiles_y;
  float4* so = dump_samples ? r->d_samples : nullptr;
#define SRT_TILES2(STATS_, TSY_, IMG_, ST_)                                                                                 \
  raster_tiles<STATS_, TSY_, IMG_><<<dim3((ntiles + (TSY_ == 32 ? 1 : kTileWavesPerBlock) - 1) / (TSY_ == 32 ? 1 : kTileWavesPerBlock)), dim3(WAVE * (TSY_ == 32 ? 1 : kTileWavesPerBlock)), 0, s>>>(P, r->d_prims, r->d_bbox, reinterpret_cast<const uint32_t*>(r->d_lists), r->d_counts, \
                                                                 r->d_counts + (size_t)P.coarse_x * P.coarse_y, r->d_aux,       \
                                                                 r->d_tabs, r->d_texels, r->d_laux, r->d_ltable, r->d_rgba, so, ST_, r->d_status, r->d_host_status)
#define SRT_TILES(STATS_, TSY_, ST_) do { if (r->has_images) SRT_TILES2(STATS_, TSY_, true, ST_); else SRT_TILES2(STATS_, TSY_, false, ST_); } while (0)
  const int tsy = P.tile_sy > 16 ? 32 : (P.tile_sy > 8 ? 16 : 8);
  if (stats) { if (tsy == 32) SRT_TILES(true, 32, r->d_stats); else if (tsy == 16) SRT_TILES(true, 16, r->d_stats); else SRT_TILES(true, 8, r->d_stats); }
  else { if (tsy == 32) SRT_TILES(false, 32, nullptr); else if (tsy == 16) SRT_TILES(false, 16, nullptr); else SRT_TILES(false, 8, nullptr); }
#undef SRT_TILES
#undef SRT_TILES2
  SRT_HIP(hipGetLastError());
  r->resolved = true;
  return SRT_OK;
}

// After the stream `s` has been synchronised: did the last setup / binning fit its storage?  Returns 1 when the frame has to
// be repeated (storage grown, bins invalidated), 0 when the frame is good, < 0 on refusal / failure.
int check_frame(srt_raster* r) {
  if (r->verified) return 0;
  const volatile uint32_t* hs = r->h_status;
  const uint32_t table_need = hs[FS_TABLE_NEED], list_need = hs[FS_LIST_NEED], flags = hs[FS_FLAGS];
  if (flags & kFlagLineUnwalkable)
    return srt::fail(SRT_ERR_UNSUPPORTED, "a line's main loop runs over coordinates beyond 2^24, where the reference's `++x` on a float no longer advances");
  bool again = false;
  if (table_need > r->ltable_cap) {
    const int st = grow(&r->d_ltable, &r->ltable_cap, (size_t)table_need + table_need / 4 + 1024);
    if (st != SRT_OK) return st;
    again = true;
  }
  if (list_need != 0u) {                           // (reported only by lists that did not fit)
    if (list_need == 0xFFFFFFFFu || list_need > (256u << 20)) {        // more than 1 GiB of lists: coarser bins
      if (r->coarse_min >= 65536) return srt::fail(SRT_ERR_UNSUPPORTED, "the frame's bin lists need more than 2^32 entries");
      r->coarse_min *= 2;
    } else {
      const int st = grow(&r->d_lists, &r->lists_cap, (size_t)list_need + list_need / 8 + 1024);
      if (st != SRT_OK) return st;
    }
    again = true;
  }
  if (again) { r->bins_valid = false; return 1; }
  r->verified = true;
  return 0;
}

// How often a frame can legitimately be repeated: check_frame() either doubles coarse_min (4 -> 65536: at most 14 times), or grows
// the lists / the line table to exactly what the frame reported for the grid it ran with (once per grid each).
constexpr int kMaxFrameAttempts = 14 * 2 + 4;

// One frame, repeated while its storage has to grow (at most a few times, and only on the first frame of a stream that needs
// more than every frame before it).  `sync_all`: the caller needs the frame complete on return; otherwise the function only
// waits when the stream is new (its needs are unknown until a frame has reported them).
int run_frame(srt_raster* r, hipStream_t s, bool dump_samples, bool stats, bool sync_all) {
  for (int attempt = 0; attempt < kMaxFrameAttempts; attempt++) {
    int st = launch_frame(r, s, dump_samples, stats);
    if (st != SRT_OK) return st;
    if (r->verified && !sync_all) return SRT_OK;
    if (!r->verified) {
      SRT_HIP(hipStreamSynchronize(s));
      st = check_frame(r);
      if (st < 0) return st;
      if (st == 1) continue;
    }
    return SRT_OK;
  }
  return srt::fail(SRT_ERR_STATE, "the frame's buffers kept growing");
}

}  // namespace

extern "C" {

int srt_raster_create(int device, srt_raster** out) {
  if (!out) return srt::fail(SRT_ERR_INVALID, "srt_raster_create: out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return srt::fail(SRT_ERR_NO_DEVICE, "no HIP device available (%s); this path has no CPU fallback",
                     e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= count) return srt::fail(SRT_ERR_INVALID, "device %d out of range [0,%d)", device, count);
  SRT_HIP(hipSetDevice(device));
  srt_raster* r = new (std::nothrow) srt_raster();
  if (!r) return srt::fail(SRT_ERR_INVALID, "out of host memory");
  r->device = device;
  if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) {
    delete r;
    return srt::fail(SRT_ERR_HIP, "hipStreamCreate failed");
  }
  if (hipMalloc(&r->d_stats, ST_COUNT * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&r->d_status, FS_COUNT * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(r->d_status, 0, FS_COUNT * sizeof(uint32_t)) != hipSuccess ||
      hipHostMalloc((void**)&r->h_status, FS_COUNT * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&r->d_host_status, r->h_status, 0) != hipSuccess) {
    (void)hipFree(r->d_stats); (void)hipFree(r->d_status);
    (void)hipStreamDestroy(r->stream);
    delete r;
    return srt::fail(SRT_ERR_HIP, "hipMalloc(stats) failed");
  }
  std::memset(r->h_status, 0, FS_COUNT * sizeof(uint32_t));
  // hipMemset is ordered on the NULL stream; the context's kernels run on a non-blocking stream that does not wait for it.  Without
  // this wait the first frame could read what the allocation held before (seen on a box whose memory had been used: a stale
  // "line cannot be walked" flag refused a stream without lines).
  (void)hipDeviceSynchronize();
  if (hipEventCreateWithFlags(&r->upload_done, hipEventDisableTiming) != hipSuccess) {
    (void)hipFree(r->d_stats); (void)hipFree(r->d_status); (void)hipHostFree(r->h_status);
    (void)hipStreamDestroy(r->stream);
    delete r;
    return srt::fail(SRT_ERR_HIP, "hipEventCreate failed");
  }
  *out = r;
  return SRT_OK;
}

int srt_raster_destroy(srt_raster* r) {
  if (!r) return SRT_OK;
  (void)hipSetDevice(r->device);
  (void)hipStreamSynchronize(r->stream);
  if (r->bound_out) (void)hipHostUnregister(r->bound_out);
  (void)hipFree(r->d_prims);
  (void)hipFree(r->d_bbox);
  (void)hipFree(r->d_lists);
  (void)hipFree(r->d_super);
  (void)hipFree(r->d_counts);
  (void)hipFree(r->d_rgba);
  (void)hipFree(r->d_samples);
  (void)hipFree(r->d_stats);
  (void)hipFree(r->d_status);
  (void)hipFree(r->d_texels);
  (void)hipFree(r->d_aux);
  (void)hipFree(r->d_tabs);
  (void)hipFree(r->d_laux);
  (void)hipFree(r->d_ltable);
  if (r->h_status) (void)hipHostFree(r->h_status);
  if (r->pending) (void)hipHostFree(r->pending);
  if (r->texel_blob) (void)hipHostFree(r->texel_blob);
  if (r->upload_done) (void)hipEventDestroy(r->upload_done);
  (void)hipStreamDestroy(r->stream);
  (void)hipGetLastError();
  delete r;
  return SRT_OK;
}

int srt_raster_add_texture(srt_raster* r, uint32_t nlevels, const uint32_t* widths, const uint32_t* heights,
                           const uint8_t* const* level_texels, uint32_t* id_out) {
  if (!r || !widths || !heights || !level_texels) return srt::fail(SRT_ERR_INVALID, "srt_raster_add_texture: NULL argument");
  if (nlevels == 0 || nlevels > SRT_MAX_MIP_LEVELS)
    return srt::fail(SRT_ERR_INVALID, "a texture has 1..%d mip levels (got %u)", SRT_MAX_MIP_LEVELS, nlevels);
  srt_raster::Tex T;
  std::memset(&T, 0, sizeof T);
  T.nlevels = nlevels;
  size_t total = r->blob_n;
  for (uint32_t k = 0; k < nlevels; k++) {
    if (!widths[k] || !heights[k] || !level_texels[k]) return srt::fail(SRT_ERR_INVALID, "texture level %u is empty", k);
    T.w[k] = widths[k]; T.h[k] = heights[k]; T.off[k] = total;
    total += 4 * (size_t)widths[k] * heights[k];
  }
  if (total > 0xFFFFFFFFull) return srt::fail(SRT_ERR_UNSUPPORTED, "more than 4 GiB of texels");
  // Level by level: what the blob still holds of the previous texture set at this offset (srt_raster_clear_textures keeps it) is
  // adopted when it is the same bytes - a compare instead of a copy, and the device copy stays valid; the first level that differs
  // ends that: from there on the levels are copied in and uploaded with the next frame that draws an image.
  for (uint32_t k = 0; k < nlevels; k++) {
    const size_t bytes = 4 * (size_t)widths[k] * heights[k];
    if (r->blob_kept >= bytes && std::memcmp(r->texel_blob + r->blob_n, level_texels[k], bytes) == 0) {
      r->blob_n += bytes; r->blob_kept -= bytes;
      continue;
    }
    r->blob_kept = 0;
    r->tex_dirty = true;
    if (r->device_blob_n > r->blob_n) r->device_blob_n = r->blob_n;
    if (r->blob_upload_pending) {                   // (the transfer of an earlier frame may still be reading the blob)
      SRT_HIP(hipSetDevice(r->device));
      SRT_HIP(hipStreamSynchronize(r->stream));
      r->blob_upload_pending = false;
    }
    if (r->blob_n + bytes > r->blob_cap) {
      SRT_HIP(hipSetDevice(r->device));
      const size_t cap = std::max(total, (r->blob_n + bytes) * 2);
      uint8_t* grown = nullptr;
      if (hipHostMalloc((void**)&grown, cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return srt::fail(SRT_ERR_INVALID, "out of pinned host memory for %zu bytes of texels", cap);
      }
      if (r->blob_n) std::memcpy(grown, r->texel_blob, r->blob_n);
      if (r->texel_blob) (void)hipHostFree(r->texel_blob);
      r->texel_blob = grown; r->blob_cap = cap;
    }
    std::memcpy(r->texel_blob + r->blob_n, level_texels[k], bytes);
    r->blob_n += bytes;
  }
  // (same bytes under other dimensions are another texture)
  const size_t idx = r->textures.size();
  if (r->rebuilding && (idx >= r->prev_textures.size() || std::memcmp(&r->prev_textures[idx], &T, sizeof T) != 0)) r->tex_dirty = true;
  if (!r->rebuilding) r->tex_dirty = true;        // a texture added to the set in use
  r->textures.push_back(T);
  r->dirty = true;
  if (id_out) *id_out = (uint32_t)r->textures.size() - 1;
  return SRT_OK;
}

int srt_raster_clear_textures(srt_raster* r) {
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_clear_textures: NULL context");
  if (r->textures.empty() && !r->rebuilding) return SRT_OK;
  // DrawSVG's redraw clears and re-adds the textures of every frame: the set is only forgotten here, the texels stay in the blob
  // (and on the device) until the next frame shows whether they are still the ones in use
  if (!r->rebuilding) { r->prev_textures.swap(r->textures); r->rebuilding = true; }
  r->textures.clear();
  r->blob_kept += r->blob_n;
  r->blob_n = 0;
  r->dirty = true;
  return SRT_OK;
}

int srt_raster_texture_upload_bytes(srt_raster* r, uint64_t* total) {
  if (!r || !total) return srt::fail(SRT_ERR_INVALID, "srt_raster_texture_upload_bytes: NULL argument");
  *total = r->texel_bytes_uploaded;
  return SRT_OK;
}

int srt_raster_set_target(srt_raster* r, uint32_t width, uint32_t height, uint32_t sample_rate) {
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_set_target: NULL context");
  if (width == 0 || height == 0) return srt::fail(SRT_ERR_INVALID, "target must be at least 1x1 (got %ux%u)", width, height);
  if (sample_rate == 0) return srt::fail(SRT_ERR_INVALID, "sample_rate must be >= 1");
  if (sample_rate > (uint32_t)TS)
    return srt::fail(SRT_ERR_UNSUPPORTED, "sample_rate %u > %d is not supported by the tile kernel", sample_rate, TS);
  if ((uint64_t)width * sample_rate > (1u << 24) || (uint64_t)height * sample_rate > (1u << 24))
    return srt::fail(SRT_ERR_UNSUPPORTED, "sample grid larger than 2^24 per side");
  SRT_HIP(hipSetDevice(r->device));
  SRT_HIP(hipStreamSynchronize(r->stream));
  RasterParams& P = r->P;
  const bool realloc_px = !r->have_target || P.w != width || P.h != height;
  P.w = width; P.h = height; P.sr = sample_rate;
  P.ssw = width * sample_rate; P.ssh = height * sample_rate;
  P.tile_px = TS / sample_rate;
  P.tile_s = P.tile_px * sample_rate;
  const uint32_t tsy = sample_rate <= 8 ? 8 : (sample_rate <= 16 ? 16 : TS);   // (re-decided per frame: choose_tile_height)
  P.tile_py = tsy / sample_rate;
  P.tile_sy = P.tile_py * sample_rate;
  P.tiles_x = (width + P.tile_px - 1) / P.tile_px;
  P.tiles_y = (height + P.tile_py - 1) / P.tile_py;
  if (realloc_px) {
    if (r->d_rgba) SRT_HIP(hipFree(r->d_rgba));
    r->d_rgba = nullptr; r->have_target = false;   // until the allocation below has succeeded
    SRT_HIP(hipMalloc(&r->d_rgba, (size_t)width * height * 4));
  }
  if (r->d_samples) { SRT_HIP(hipFree(r->d_samples)); r->d_samples = nullptr; }
  r->have_target = true;
  r->resolved = false;
  r->bins_valid = false;                         // another tiling: setup (the lines' tables follow the target) and binning run again
  r->verified = false;
  r->coarse_min = 4;
  r->dirty = true;                               // (image records carry per-target tables)
  r->aux_valid = false;
  return SRT_OK;
}

int srt_raster_clear(srt_raster* r) {
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_clear: NULL context");
  { const int st = settle_upload(r); if (st != SRT_OK) return st; }   // (the pinned stream buffer is about to be rewritten)
  r->pending_n = 0;
  r->dirty = true;
  r->resolved = false;
  return SRT_OK;
}

int srt_raster_submit(srt_raster* r, const srt_prim* prims, size_t n) {
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_submit: NULL context");
  if (n && !prims) return srt::fail(SRT_ERR_INVALID, "srt_raster_submit: prims is NULL");
  if (!r->have_target) return srt::fail(SRT_ERR_STATE, "srt_raster_submit before srt_raster_set_target");
  for (size_t i = 0; i < n; i++)
    if (prims[i].kind != SRT_PRIM_TRIANGLE && prims[i].kind != SRT_PRIM_POINT && prims[i].kind != SRT_PRIM_IMAGE && prims[i].kind != SRT_PRIM_LINE)
      return srt::fail(SRT_ERR_INVALID, "primitive %zu has unknown kind %u", i, prims[i].kind);
  { const int st = settle_upload(r); if (st != SRT_OK) return st; }
  if (r->pending_n + n > r->pending_cap) {
    SRT_HIP(hipSetDevice(r->device));
    SRT_HIP(hipStreamSynchronize(r->stream));      // (an upload may still be reading the old buffer)
    size_t cap = (r->pending_n + n) * 2 + 1024;
    srt_prim* grown = nullptr;
    if (hipHostMalloc((void**)&grown, cap * sizeof(srt_prim), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return srt::fail(SRT_ERR_INVALID, "out of pinned host memory for %zu primitives", cap);
    }
    if (r->pending_n) std::memcpy(grown, r->pending, r->pending_n * sizeof(srt_prim));
    if (r->pending) (void)hipHostFree(r->pending);
    r->pending = grown; r->pending_cap = cap;
  }
  if (n) std::memcpy(r->pending + r->pending_n, prims, n * sizeof(srt_prim));
  r->pending_n += n;
  r->dirty = true;
  return SRT_OK;
}

int srt_raster_bind_output(srt_raster* r, uint8_t* host_rgba8, size_t bytes) {
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_bind_output: NULL context");
  SRT_HIP(hipSetDevice(r->device));
  SRT_HIP(hipStreamSynchronize(r->stream));
  if (r->bound_out) {
    if (hipHostUnregister(r->bound_out) != hipSuccess) (void)hipGetLastError();   // (the memory may be gone already: nothing to undo then)
    r->bound_out = nullptr;
  }
  if (!host_rgba8 || !bytes) return SRT_OK;
  if (hipHostRegister(host_rgba8, bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();                       // not fatal: srt_raster_resolve works with pageable memory as well
    return SRT_OK;
  }
  r->bound_out = host_rgba8;
  return SRT_OK;
}

int srt_raster_resolve_device(srt_raster* r, void* stream, const uint8_t** d_rgba8_out) {
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_resolve_device: NULL context");
  if (!r->have_target) return srt::fail(SRT_ERR_STATE, "resolve before srt_raster_set_target");
  SRT_HIP(hipSetDevice(r->device));
  hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream; NULL is the HIP default stream
  if (r->dirty) {
    int st = upload_stream(r);
    if (st != SRT_OK) return st;
    if (s != r->stream) SRT_HIP(hipStreamSynchronize(r->stream));  // upload went on the context stream
  }
  int st = run_frame(r, s, false, false, false);
  if (st != SRT_OK) return st;
  if (d_rgba8_out) *d_rgba8_out = (const uint8_t*)r->d_rgba;
  return SRT_OK;
}

int srt_raster_resolve(srt_raster* r, uint8_t* rgba8_out) {
  if (!rgba8_out) return srt::fail(SRT_ERR_INVALID, "srt_raster_resolve: output buffer is NULL");
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_resolve: NULL context");
  if (!r->have_target) return srt::fail(SRT_ERR_STATE, "resolve before srt_raster_set_target");
  SRT_HIP(hipSetDevice(r->device));
  if (r->dirty) { int st = upload_stream(r); if (st != SRT_OK) return st; }
  // frame and read-back are enqueued together; one wait.  (A frame whose storage has to grow is repeated: run_frame.)
  for (int attempt = 0; attempt < kMaxFrameAttempts; attempt++) {
    int st = launch_frame(r, r->stream, false, false);
    if (st != SRT_OK) return st;
    SRT_HIP(hipMemcpyAsync(rgba8_out, r->d_rgba, (size_t)r->P.w * r->P.h * 4, hipMemcpyDeviceToHost, r->stream));
    SRT_HIP(hipStreamSynchronize(r->stream));
    st = check_frame(r);
    if (st < 0) return st;
    if (st == 0) return SRT_OK;
  }
  return srt::fail(SRT_ERR_STATE, "the frame's buffers kept growing");
}

int srt_raster_read_samples(srt_raster* r, float* samples_out) {
  if (!r || !samples_out) return srt::fail(SRT_ERR_INVALID, "srt_raster_read_samples: NULL argument");
  if (!r->have_target) return srt::fail(SRT_ERR_STATE, "read_samples before srt_raster_set_target");
  SRT_HIP(hipSetDevice(r->device));
  const size_t bytes = (size_t)r->P.ssw * r->P.ssh * sizeof(float4);
  if (!r->d_samples) SRT_HIP(hipMalloc(&r->d_samples, bytes));
  if (r->dirty) { int st = upload_stream(r); if (st != SRT_OK) return st; }
  int st = run_frame(r, r->stream, true, false, true);
  if (st != SRT_OK) return st;
  SRT_HIP(hipMemcpyAsync(samples_out, r->d_samples, bytes, hipMemcpyDeviceToHost, r->stream));
  SRT_HIP(hipStreamSynchronize(r->stream));
  return SRT_OK;
}

int srt_raster_stats(srt_raster* r, srt_raster_stats_t* out) {
  if (!r || !out) return srt::fail(SRT_ERR_INVALID, "srt_raster_stats: NULL argument");
  if (!r->have_target) return srt::fail(SRT_ERR_STATE, "stats before srt_raster_set_target");
  SRT_HIP(hipSetDevice(r->device));
  if (r->dirty) { int st = upload_stream(r); if (st != SRT_OK) return st; }
  r->verified = false;                           // (the stats pass runs setup and binning again: its status is checked)
  int st = run_frame(r, r->stream, false, true, true);
  if (st != SRT_OK) return st;
  unsigned long long h[ST_COUNT];
  SRT_HIP(hipMemcpyAsync(h, r->d_stats, sizeof h, hipMemcpyDeviceToHost, r->stream));
  SRT_HIP(hipStreamSynchronize(r->stream));
  out->sample_tests = h[ST_TESTS_REF];
  out->sample_tests_in_target = h[ST_TESTS_TARGET];
  out->fragments = h[ST_FRAGMENTS];
  out->point_samples = h[ST_POINT_SAMPLES];
  out->bin_entries = h[ST_BIN_ENTRIES];
  out->list_bytes = (uint64_t)(r->super_cap + r->counts_cap) * sizeof(uint32_t) + (uint64_t)r->lists_cap * sizeof(uint2);
  return SRT_OK;
}

int srt_raster_invalidate(srt_raster* r) {
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_invalidate: NULL context");
  r->bins_valid = false;       // (`verified` stays: the storage has proven large enough for this stream)
  return SRT_OK;
}

int srt_raster_sync(srt_raster* r) {
  if (!r) return srt::fail(SRT_ERR_INVALID, "srt_raster_sync: NULL context");
  SRT_HIP(hipSetDevice(r->device));
  SRT_HIP(hipStreamSynchronize(r->stream));
  return SRT_OK;
}

}  // extern "C"

/*
 * raster_oracle.c — CPU restatement of the DrawSVG rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP rasterizer is compared with; it is
 * called from tests/, from __graft_entry__.smoke() and from bench.py's cpu_baseline leg, and from
 * nowhere else.  The product (soft-rendering-toolsets_amd/) never links or loads it.
 *
 * Parity status: PINNED.  tests/test_raster_oracle.py checks this restatement bit-for-bit (RGBA8 and
 * the float supersample buffer) against fixtures under tests/golden/ that were produced by compiling
 * the reference's own software_renderer.cpp (oracle/ref_harness/raster_ref.cpp, oracle/Makefile.ref)
 * and, when /root/reference is present, against that build directly.
 *
 * Each function cites the reference lines it restates; paths are relative to
 * /root/reference/Assignments/DrawSVG/src/.  Plain C99, scalar loops, one thread.  Build with
 * -ffp-contract=off and without -march=native so that no FMA is formed (the reference's x86-64
 * build has none).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "srt_raster.h" /* srt_prim (the stream record layout is shared with the C ABI) */

typedef struct {
    uint32_t w, h, sr;   /* target_w, target_h, sample_rate */
    uint32_t ssw, ssh;   /* supersample_w, supersample_h */
    float* ss;           /* super_sample_buffer: 4 floats per sample, row-major */
    uint64_t tests;      /* inside_triangle calls (what the reference performs, unclipped) */
    uint64_t tests_in;   /* ... of which inside the sample grid */
    uint64_t frags;      /* covered in-bounds fill_sample calls from rasterize_triangle */
    uint64_t pts;        /* in-bounds fill_sample calls from rasterize_point */
    uint64_t img;        /* in-bounds fill_sample calls from rasterize_image */
} oracle_raster;

/* std::min / std::max as libstdc++ defines them (argument order matters for NaN / -0). */
static float min_f(float a, float b) { return (b < a) ? b : a; }
static float max_f(float a, float b) { return (a < b) ? b : a; }

/* CMU462::clamp, CMU462/include/CMU462/misc.h:69 */
static float clamp_f(float x, float lo, float hi) { return min_f(max_f(x, lo), hi); }

/* fill_sample, software_renderer.cpp:634-658.  Returns 1 if the sample was in bounds. */
static int fill_sample(oracle_raster* o, int sx, int sy, const float c[4]) {
    if (sx < 0 || (uint32_t)sx >= o->ssw) return 0;
    if (sy < 0 || (uint32_t)sy >= o->ssh) return 0;
    float* s = o->ss + 4 * ((size_t)sx + (size_t)sy * o->ssw);
    s[0] = clamp_f((c[0] + (1 - c[3]) * (s[0] / 255.0f)) * 255.0f, 0.0f, 255.0f);
    s[1] = clamp_f((c[1] + (1 - c[3]) * (s[1] / 255.0f)) * 255.0f, 0.0f, 255.0f);
    s[2] = clamp_f((c[2] + (1 - c[3]) * (s[2] / 255.0f)) * 255.0f, 0.0f, 255.0f);
    s[3] = clamp_f((1.0f - ((1.0f - c[3]) * (1 - (s[3] / 255.0f)))) * 255.0f, 0.0f, 255.0f);
    return 1;
}

/* inside_triangle, software_renderer.cpp:519-538.  tri = {x0,y0,x1,y1,x2,y2} widened to double. */
static int inside_triangle(double px, double py, const double t[6]) {
    double e01x = t[2] - t[0], e01y = t[3] - t[1];
    double e12x = t[4] - t[2], e12y = t[5] - t[3];
    double e20x = t[0] - t[4], e20y = t[1] - t[5];
    double p0x = px - t[0], p0y = py - t[1];
    double p1x = px - t[2], p1y = py - t[3];
    double p2x = px - t[4], p2y = py - t[5];
    float cross1 = (float)(e01x * p0y - e01y * p0x);
    float cross2 = (float)(e12x * p1y - e12y * p1x);
    float cross3 = (float)(e20x * p2y - e20y * p2x);
    int ccw = (cross1 * cross2 >= 0) && (cross2 * cross3 >= 0) && (cross1 * cross3 >= 0);
    int cw = (cross1 * cross2 <= 0) && (cross2 * cross3 <= 0) && (cross1 * cross3 <= 0);
    return ccw || cw;
}

/* rasterize_triangle, software_renderer.cpp:456-516. */
static void rasterize_triangle(oracle_raster* o, const float v[6], const float color[4]) {
    double t[6];
    for (int i = 0; i < 6; i++) t[i] = (double)v[i];
    float xmin = floorf(min_f(v[0], min_f(v[2], v[4])));
    float ymin = floorf(min_f(v[1], min_f(v[3], v[5])));
    float xmax = ceilf(max_f(v[0], max_f(v[2], v[4])));
    float ymax = ceilf(max_f(v[1], max_f(v[3], v[5])));
    xmin *= (float)o->sr; xmax *= (float)o->sr; ymin *= (float)o->sr; ymax *= (float)o->sr;
    if (!(xmin <= xmax) || !(ymin <= ymax)) return; /* NaN: the reference's loops do not run */
    {
        double nx = (double)xmax - (double)xmin + 1.0, ny = (double)ymax - (double)ymin + 1.0;
        if (nx * ny < 1.8e19) o->tests += (uint64_t)(nx * ny);
    }
    /* The reference walks x = xmin..xmax, y = ymin..ymax (inclusive, step 1.0, integer valued) and lets
     * fill_sample reject out-of-grid samples; restricting the walk to the grid visits the same
     * in-grid samples in the same order (x outer, y inner). */
    double x_lo = xmin < 0.0f ? 0.0 : (double)xmin, x_hi = (double)xmax;
    double y_lo = ymin < 0.0f ? 0.0 : (double)ymin, y_hi = (double)ymax;
    if (x_hi > (double)(o->ssw - 1)) x_hi = (double)(o->ssw - 1);
    if (y_hi > (double)(o->ssh - 1)) y_hi = (double)(o->ssh - 1);
    for (double x = x_lo; x <= x_hi; x++) {
        for (double y = y_lo; y <= y_hi; y++) {
            o->tests_in++;
            if (inside_triangle(x / (double)o->sr, y / (double)o->sr, t)) {
                o->frags += (uint64_t)fill_sample(o, (int)x, (int)y, color);
            }
        }
    }
}

/* double -> int as the x86-64 reference build does it: out-of-range and NaN give INT_MIN. */
static int to_int(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int)v;
}

/* rasterize_point, software_renderer.cpp:272-301. */
static void rasterize_point(oracle_raster* o, double x, double y, const float color[4]) {
    for (int i = 0; i < (int)o->sr; i++) {
        for (int j = 0; j < (int)o->sr; j++) {
            o->pts += (uint64_t)fill_sample(o, to_int(x * (double)o->sr + i), to_int(y * (double)o->sr + j), color);
        }
    }
}

/* ipart / fpart / rfpart, software_renderer.cpp:355-363 (float in, float out) */
static float wu_ipart(float x) { return floorf(x); }
static float wu_fpart(float x) { return x - floorf(x); }
static float wu_rfpart(float x) { return 1 - wu_fpart(x); }

/* rasterize_line (software_renderer.cpp:303-318) = rasterize_line_xiaolinwu, :365-454: the stroke alpha is REPLACED by the
 * Wu coverage; end points first, then the main loop, whose bound subtracts sample_rate.  Returns -1 where the reference's
 * `for (float x = ...; x <= last; ++x)` would never end (|x| >= 2^24: ++x no longer advances) - the product refuses those. */
static int rasterize_line(oracle_raster* o, float x0, float y0, float x1, float y1, const float color_in[4]) {
    float color[4] = {color_in[0], color_in[1], color_in[2], color_in[3]};
    const int steep = fabsf(x1 - x0) < fabsf(y1 - y0);
    float t;
    if (steep) { t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
    if (x0 > x1) { t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
    const float dx = x1 - x0, dy = y1 - y0;
    const float gradient = (dx == 0.0f) ? 1.0f : dy / dx;
    /* first end point */
    float xend = roundf(x0);
    float yend = y0 + gradient * (xend - x0);
    float xgap = wu_rfpart(x0 + 0.5f);
    const float xpxl1 = xend, ypxl1 = wu_ipart(yend);
    color[3] = wu_rfpart(yend) * xgap;
    if (steep) rasterize_point(o, ypxl1, xpxl1, color); else rasterize_point(o, xpxl1, ypxl1, color);
    color[3] = wu_fpart(yend) * xgap;
    if (steep) rasterize_point(o, ypxl1 + 1, xpxl1, color); else rasterize_point(o, xpxl1, ypxl1 + 1, color);
    float intery = yend + gradient;
    /* second end point */
    xend = roundf(x1);
    yend = y1 + gradient * (xend - x1);
    xgap = wu_fpart(x1 + 0.5f);
    const float xpxl2 = xend, ypxl2 = wu_ipart(yend);
    color[3] = wu_rfpart(yend) * xgap;
    if (steep) rasterize_point(o, ypxl2, xpxl2, color); else rasterize_point(o, xpxl2, ypxl2, color);
    color[3] = wu_fpart(yend) * xgap;
    if (steep) rasterize_point(o, ypxl2 + 1, xpxl2, color); else rasterize_point(o, xpxl2, ypxl2 + 1, color);
    /* main loop */
    const float first = xpxl1 + 1, last = xpxl2 - (float)o->sr;
    if (first <= last && !(first > -16777216.0f && last < 16777216.0f)) return -1;
    for (float x = first; x <= last; ++x) {
        color[3] = wu_rfpart(intery);
        if (steep) rasterize_point(o, wu_ipart(intery), x, color); else rasterize_point(o, x, wu_ipart(intery), color);
        color[3] = wu_fpart(intery);
        if (steep) rasterize_point(o, wu_ipart(intery) + 1, x, color); else rasterize_point(o, x, wu_ipart(intery) + 1, color);
        intery += gradient;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Images: rasterize_image (software_renderer.cpp:540-570) + Sampler2DImp (texture.cpp:53-193).
 * A texture is a mip chain of RGBA8 levels.  Reads past the end of a level's texel vector (the reference
 * indexes column `width` and row `height` at the right / bottom half-texel border, texture.cpp:158-166) are
 * undefined there; here - and in the HIP kernel, and in the reference build the goldens come from, whose
 * vectors get zeroed slack - they read zeros.  A column index of `width` inside the buffer wraps to the next
 * row exactly as in the reference.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t nlevels;
    uint32_t width[SRT_MAX_MIP_LEVELS], height[SRT_MAX_MIP_LEVELS];
    const uint8_t* texels[SRT_MAX_MIP_LEVELS];
} oracle_texture;

/* GetColorFromTexture, texture.cpp:19-25 (x, y are the float arguments converted to int by the call) */
static void get_texel(const oracle_texture* t, int level, int x, int y, float c[4]) {
    size_t w = t->width[level], n = 4 * (size_t)t->width[level] * t->height[level];
    size_t idx = 4 * ((size_t)x + (size_t)y * w);
    for (int k = 0; k < 4; k++) c[k] = (idx + k < n ? t->texels[level][idx + k] : 0) / 255.0f;
}

/* lerpColor<float>, texture.cpp:14-17: (1 - ratio) * start + ratio * ends */
static void lerp_color(float ratio, const float a[4], const float b[4], float out[4]) {
    for (int k = 0; k < 4; k++) out[k] = (1 - ratio) * a[k] + ratio * b[k];
}

/* sample_bilinear, texture.cpp:145-169 */
static void sample_bilinear(const oracle_texture* t, float u, float v, int level, float out[4]) {
    if (level < 0 || level >= (int)t->nlevels) { out[0] = 1; out[1] = 0; out[2] = 1; out[3] = 1; return; }
    float wf = (float)t->width[level], hf = (float)t->height[level];
    float su = clamp_f(u, 0.0f, 0.99999f) * wf;
    float sv = clamp_f(v, 0.0f, 0.99999f) * hf;
    float u0 = floorf(su) + 0.5f, v0 = floorf(sv) + 0.5f, u1, v1, tmp;
    if (su - (int)su < 0.5f) { u1 = clamp_f(u0 - 1, 0.0f, wf); tmp = u1; u1 = u0; u0 = tmp; }
    else { u1 = clamp_f(u0 + 1, 0.0f, wf); }
    if (sv - (int)sv < 0.5f) { v1 = clamp_f(v0 - 1, 0.0f, hf); tmp = v1; v1 = v0; v0 = tmp; }
    else { v1 = clamp_f(v0 + 1, 0.0f, hf); }
    float c00[4], c10[4], c01[4], c11[4], h1[4], h2[4];
    get_texel(t, level, (int)u0, (int)v0, c00);
    get_texel(t, level, (int)u1, (int)v0, c10);
    get_texel(t, level, (int)u0, (int)v1, c01);
    get_texel(t, level, (int)u1, (int)v1, c11);
    lerp_color((su - u0) / (u1 - u0), c00, c10, h1);
    lerp_color((su - u0) / (u1 - u0), c01, c11, h2);
    lerp_color((sv - v0) / (v1 - v0), h1, h2, out);
}

/* The mip level arithmetic of sample_trilinear (texture.cpp:171-193), which depends on the image only:
 * mode 0 magenta, 1 bilinear at `low`, 2 lerp(frac, bilinear(low), bilinear(low + 1)). */
void srt_oracle_trilinear_level(uint32_t tex_w, uint32_t tex_h, uint32_t nlevels, float u_scale, float v_scale,
                                int* mode, int* low, float* frac) {
    double ax = (double)((float)tex_w / u_scale), bx = (double)((float)tex_h / u_scale);
    double ay = (double)((float)tex_w / v_scale), by = (double)((float)tex_h / v_scale);
    float lsx = (float)(pow(ax, 2) + pow(bx, 2));
    float lsy = (float)(pow(ay, 2) + pow(by, 2));
    float level = log2f(sqrtf(max_f(lsx, lsy)));
    if (level < 0) level = 0.0f;
    *low = 0; *frac = 0.0f;
    if (level >= (float)nlevels) { *mode = 0; return; }   /* also taken for +inf; NaN falls through as in the reference */
    int lo = (int)floorf(level), hi = lo + 1;
    if (hi >= (int)nlevels) { *mode = 1; *low = (int)nlevels - 1; return; }
    *mode = 2; *low = lo; *frac = level - (int)level;
}

/* sample_trilinear, texture.cpp:171-193 */
static void sample_trilinear(const oracle_texture* t, float u, float v, float u_scale, float v_scale, float out[4]) {
    int mode, low; float frac;
    srt_oracle_trilinear_level(t->width[0], t->height[0], t->nlevels, u_scale, v_scale, &mode, &low, &frac);
    if (mode == 0) { out[0] = 1; out[1] = 0; out[2] = 1; out[3] = 1; return; }
    if (mode == 1) { sample_bilinear(t, u, v, low, out); return; }
    float a[4], b[4];
    sample_bilinear(t, u, v, low, a);
    sample_bilinear(t, u, v, low + 1, b);
    lerp_color(frac, a, b, out);
}

/* rasterize_image, software_renderer.cpp:540-570.  Returns -1 if the float loops would not terminate. */
static int rasterize_image(oracle_raster* o, float x0, float y0, float x1, float y1, const oracle_texture* t) {
    float uscale = x1 - x0, vscale = y1 - y0;
    x0 *= (float)o->sr; x1 *= (float)o->sr; y0 *= (float)o->sr; y1 *= (float)o->sr;
    uint64_t guard = 0;
    for (float x = x0; x <= x1; x++) {
        if (++guard > (1u << 26) || x + 1 == x) return -1;
        for (float y = y0; y <= y1; y++) {
            if (y + 1 == y) return -1;
            int sx = to_int((double)x), sy = to_int((double)y);
            if (sx < 0 || (uint32_t)sx >= o->ssw || sy < 0 || (uint32_t)sy >= o->ssh) continue; /* fill_sample rejects */
            float u = (float)((x + 0.5 - x0) / (x1 - x0));
            float v = (float)((y + 0.5 - y0) / (y1 - y0));
            float c[4];
            sample_trilinear(t, u, v, uscale, vscale, c);
            o->img += (uint64_t)fill_sample(o, sx, sy, c);
        }
    }
    return 0;
}

/* Sampler2DImp::generate_mips(tex, 0), texture.cpp:53-120: level sizes and the 2x2 box filter.
 * in: level 0 (w0 x h0 RGBA8).  out_w/out_h: SRT_MAX_MIP_LEVELS entries; out_blob: every level back to back
 * (level 0 first), capacity blob_cap bytes.  Returns the number of levels, or -1. */
int srt_oracle_generate_mips(const uint8_t* level0, uint32_t w0, uint32_t h0, uint32_t* out_w, uint32_t* out_h,
                             uint8_t* out_blob, uint64_t blob_cap) {
    if (!level0 || !w0 || !h0) return -1;
    int base = (int)(w0 > h0 ? w0 : h0);
    int sub = (int)(log2f((float)base));
    if (sub > SRT_MAX_MIP_LEVELS - 1) sub = SRT_MAX_MIP_LEVELS - 1;
    uint64_t off = 0;
    int w = (int)w0, h = (int)h0;
    out_w[0] = w0; out_h[0] = h0;
    if (4ull * w0 * h0 > blob_cap) return -1;
    memcpy(out_blob, level0, 4ull * w0 * h0);
    const uint8_t* prev = out_blob;
    int pw = w, ph = h;
    off = 4ull * w0 * h0;
    for (int i = 1; i <= sub; i++) {
        w = w / 2 > 1 ? w / 2 : 1;
        h = h / 2 > 1 ? h / 2 : 1;
        out_w[i] = (uint32_t)w; out_h[i] = (uint32_t)h;
        if (off + 4ull * w * h > blob_cap) return -1;
        uint8_t* cur = out_blob + off;
        size_t pn = 4 * (size_t)pw * ph;
        for (int x = 0; x < w; x++) {
            for (int y = 0; y < h; y++) {
                float sum[4] = {0, 0, 0, 0};
                for (int m = 0; m < 2; m++) {
                    for (int n = 0; n < 2; n++) {
                        size_t idx = 4 * ((size_t)(2 * x + m) + (size_t)(2 * y + n) * pw);
                        for (int k = 0; k < 4; k++) sum[k] += (idx + k < pn ? prev[idx + k] : 0) / 255.0f;
                    }
                }
                for (int k = 0; k < 4; k++) {
                    sum[k] *= 0.25f;
                    cur[4 * (x + y * w) + k] = (uint8_t)(255.f * max_f(0.0f, min_f(1.0f, sum[k])));  /* float_to_uint8 */
                }
            }
        }
        prev = cur; pw = w; ph = h;
        off += 4ull * w * h;
    }
    return sub + 1;
}

/* resolve, software_renderer.cpp:573-622. */
static void resolve(const oracle_raster* o, uint8_t* target) {
    const size_t sr = o->sr;
    for (size_t x = 0; x < o->ssw; x += sr) {
        for (size_t y = 0; y < o->ssh; y += sr) {
            float r = 0, g = 0, b = 0, a = 0;
            for (size_t i = 0; i < sr; ++i) {
                for (size_t j = 0; j < sr; ++j) {
                    size_t pos = 4 * (x + i + (y + j) * o->ssw);
                    r += o->ss[pos];
                    g += o->ss[pos + 1];
                    b += o->ss[pos + 2];
                    a += o->ss[pos + 3];
                }
            }
            r /= (float)(sr * sr);
            g /= (float)(sr * sr);
            b /= (float)(sr * sr);
            a /= (float)(sr * sr);
            size_t pix = 4 * ((x / sr) + (y / sr) * o->w);
            target[pix] = (uint8_t)(r);
            target[pix + 1] = (uint8_t)(g);
            target[pix + 2] = (uint8_t)(b);
            target[pix + 3] = (uint8_t)(a);
        }
    }
}

/*
 * One frame: clear_target (software_renderer.h:93-98) -> the ordered rasterize_* calls -> resolve.
 * Textures (for SRT_PRIM_IMAGE records, whose `reserved` field is the texture index): ntex mip chains described
 * by tex_nlevels[ntex] and, concatenated over the textures, level_w / level_h / level_off (byte offsets into blob).
 * counts (optional) receives {tests, tests_in_target, fragments, point_samples}.
 * samples_out (optional) receives the float supersample buffer as it stands before resolve.
 * Returns 0, or -1 on bad arguments / allocation failure.
 */
int srt_oracle_raster_frame_tex(const srt_prim* prims, size_t n, uint32_t w, uint32_t h, uint32_t sr, uint32_t ntex,
                                const uint32_t* tex_nlevels, const uint32_t* level_w, const uint32_t* level_h,
                                const uint64_t* level_off, const uint8_t* blob, uint8_t* rgba_out, float* samples_out,
                                uint64_t counts[4]) {
    if (!w || !h || !sr || !rgba_out || (n && !prims)) return -1;
    oracle_raster o;
    memset(&o, 0, sizeof o);
    o.w = w; o.h = h; o.sr = sr;
    o.ssw = w * sr; o.ssh = h * sr;
    size_t nfloats = 4 * (size_t)o.ssw * o.ssh;
    o.ss = (float*)malloc(nfloats * sizeof(float));
    oracle_texture* tex = ntex ? (oracle_texture*)calloc(ntex, sizeof(oracle_texture)) : NULL;
    if (!o.ss || (ntex && !tex)) { free(o.ss); free(tex); return -1; }
    for (uint32_t t = 0, l = 0; t < ntex; t++) {
        if (tex_nlevels[t] == 0 || tex_nlevels[t] > SRT_MAX_MIP_LEVELS) { free(o.ss); free(tex); return -1; }
        tex[t].nlevels = tex_nlevels[t];
        for (uint32_t k = 0; k < tex_nlevels[t]; k++, l++) {
            tex[t].width[k] = level_w[l]; tex[t].height[k] = level_h[l]; tex[t].texels[k] = blob + level_off[l];
        }
    }
    memset(rgba_out, 255, 4 * (size_t)w * h);
    for (size_t i = 0; i < nfloats; i++) o.ss[i] = 255.0f;

    int rc = 0;
    for (size_t i = 0; i < n && rc == 0; i++) {
        const srt_prim* p = &prims[i];
        if (p->kind == SRT_PRIM_TRIANGLE) rasterize_triangle(&o, p->v.tri, p->rgba);
        else if (p->kind == SRT_PRIM_POINT) rasterize_point(&o, p->v.point[0], p->v.point[1], p->rgba);
        else if (p->kind == SRT_PRIM_LINE) rc = rasterize_line(&o, p->v.tri[0], p->v.tri[1], p->v.tri[2], p->v.tri[3], p->rgba);
        else if (p->kind == SRT_PRIM_IMAGE && p->reserved < ntex)
            rc = rasterize_image(&o, p->v.tri[0], p->v.tri[1], p->v.tri[2], p->v.tri[3], &tex[p->reserved]);
        else rc = -1;
    }
    if (rc == 0) {
        if (samples_out) memcpy(samples_out, o.ss, nfloats * sizeof(float));
        resolve(&o, rgba_out);
        if (counts) { counts[0] = o.tests; counts[1] = o.tests_in; counts[2] = o.frags; counts[3] = o.pts; }
    }
    free(o.ss);
    free(tex);
    return rc;
}

int srt_oracle_raster_frame(const srt_prim* prims, size_t n, uint32_t w, uint32_t h, uint32_t sr,
                            uint8_t* rgba_out, float* samples_out, uint64_t counts[4]) {
    return srt_oracle_raster_frame_tex(prims, n, w, h, sr, 0, NULL, NULL, NULL, NULL, NULL, rgba_out, samples_out, counts);
}

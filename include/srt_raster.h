/*
 * srt_raster.h — C ABI of the MI355X (gfx950) rasterizer hot path.
 *
 * Drop-in boundary for the DrawSVG software rasterizer of the reference
 * (paths below are relative to /root/reference/Assignments/DrawSVG/):
 *
 *   srt_raster_set_target   <- SoftwareRendererImp::set_render_target  src/software_renderer.cpp:73-92
 *                              + SoftwareRendererImp::set_sample_rate  src/software_renderer.cpp:55-70
 *   srt_raster_clear        <- SoftwareRendererImp::clear_target       src/software_renderer.h:93-98
 *   srt_raster_submit       <- the ordered calls draw_svg/draw_element make into
 *                              rasterize_triangle  src/software_renderer.cpp:456-516 (+ inside_triangle :519-538,
 *                                                   fill_sample :634-658)
 *                              rasterize_line      src/software_renderer.cpp:303-318 -> rasterize_line_xiaolinwu :365-454
 *                                                   (expanded into its rasterize_point calls ON THE DEVICE)
 *                              rasterize_point     src/software_renderer.cpp:272-301
 *                              rasterize_image     src/software_renderer.cpp:540-570 (+ Sampler2DImp, src/texture.cpp:145-193)
 *   srt_raster_add_texture  <- the CMU462::Texture of an <image> element after Sampler2D::generate_mips
 *                              (DrawSVG::regenerate_mipmap, src/drawsvg.cpp:462-474)
 *   srt_raster_resolve      <- SoftwareRendererImp::resolve            src/software_renderer.cpp:573-622
 *
 * The host element walk (draw_svg / draw_element / transform stack, src/software_renderer.cpp:17-52,94-265)
 * stays on the host: it produces the ORDERED primitive stream that is handed to srt_raster_submit - one
 * record per rasterize_triangle / rasterize_line / rasterize_point / rasterize_image call. Painter's
 * order is part of the contract: primitives are blended in stream order.
 *
 * Conventions: every entry point returns 0 on success and a negative srt_status on
 * failure (the path tracer's render calls also SRT_CANCELLED = 1 after srt_pt_cancel); srt_last_error() returns a thread-local message. No exceptions cross the ABI.
 * The caller owns every host buffer. One context may be used by one thread at a time.
 * There is NO CPU fallback: without a HIP device srt_raster_create fails.
 */
#ifndef SRT_RASTER_H
#define SRT_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum srt_status {
    SRT_CANCELLED = 1,        /* not an error: srt_pt_cancel cut the call short, its output was not written (srt_pt.h) */
    SRT_OK = 0,
    SRT_ERR_INVALID = -1,     /* bad argument */
    SRT_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    SRT_ERR_HIP = -3,         /* a HIP runtime call failed */
    SRT_ERR_UNSUPPORTED = -4, /* valid in the reference, not supported by this path */
    SRT_ERR_STATE = -5        /* call order violated (e.g. submit before set_target) */
} srt_status;

/* Primitive kinds of the ordered stream. */
enum {
    SRT_PRIM_TRIANGLE = 1, /* rasterize_triangle(x0,y0,x1,y1,x2,y2,color) — args already narrowed to float */
    SRT_PRIM_POINT = 2,    /* rasterize_point(x,y,color) — args are double, sr*sr block fill */
    SRT_PRIM_IMAGE = 3,    /* rasterize_image(x0,y0,x1,y1,tex) — tri[0..3] = the four float parameters, `reserved` =
                              texture id from srt_raster_add_texture; sampled with Sampler2DImp::sample_trilinear */
    SRT_PRIM_LINE = 4      /* rasterize_line(x0,y0,x1,y1,color) = rasterize_line_xiaolinwu — tri[0..3] = the four float
                              parameters; rgba[3] is ignored (the reference REPLACES the stroke alpha by the Wu coverage).
                              The device performs the reference's rasterize_point calls in their order: first end point
                              (2), second end point (2), then the main loop, whose upper bound subtracts the sample rate
                              (:434,445).  Lines whose main loop the reference could not walk (|x| >= 2^24, where its
                              `++x` on a float stops advancing) are refused with SRT_ERR_UNSUPPORTED at resolve. */
};

#define SRT_MAX_MIP_LEVELS 14 /* kMaxMipLevels, D/src/texture.h:9 */

/* One 48-byte record of the ordered stream. Screen-space coordinates (pixels). */
typedef struct srt_prim {
    uint32_t kind;
    uint32_t reserved; /* 0; for SRT_PRIM_IMAGE the texture id */
    union {
        float tri[6];     /* x0 y0 x1 y1 x2 y2 — the six float parameters of rasterize_triangle (images, lines: the first four) */
        double point[2];  /* x y — the two double parameters of rasterize_point */
    } v;
    float rgba[4]; /* CMU462::Color r g b a, NOT premultiplied (fill_sample semantics) */
} srt_prim;

typedef struct srt_raster srt_raster; /* opaque */

/* Per-frame work counters (optional, filled by srt_raster_stats). */
typedef struct srt_raster_stats_t {
    uint64_t sample_tests;           /* inside_triangle evaluations the reference performs (bbox area, unclipped) */
    uint64_t sample_tests_in_target; /* the subset that lies inside the sample grid (what the kernel evaluates) */
    uint64_t fragments;      /* covered, in-bounds fill_sample calls reached from rasterize_triangle */
    uint64_t point_samples;  /* in-bounds fill_sample calls reached from rasterize_point (directly or through a line) */
    uint64_t bin_entries;    /* (primitive, tile) pairs processed */
    uint64_t list_bytes;     /* device memory held by the two levels of ordered bin lists for this stream / target */
} srt_raster_stats_t;

const char* srt_last_error(void);

/* Create / destroy a context on HIP device `device`. */
int srt_raster_create(int device, srt_raster** out);
int srt_raster_destroy(srt_raster* r);

/* Size of the render target in pixels and the supersample rate (samples per pixel side, >= 1).
 * (Re)allocates the device-side state. Equivalent to set_render_target + set_sample_rate. */
int srt_raster_set_target(srt_raster* r, uint32_t width, uint32_t height, uint32_t sample_rate);

/* Start a new frame: every sample becomes (255,255,255,255) and the pending stream is dropped. */
int srt_raster_clear(srt_raster* r);

/* Append `n` primitives (host memory) to the frame's ordered stream. May be called repeatedly. */
int srt_raster_submit(srt_raster* r, const srt_prim* prims, size_t n);

/* Textures for SRT_PRIM_IMAGE records (CMU462::Texture, D/src/texture.h:24-28): a mip chain of 1..14 RGBA8
 * levels as Sampler2D::generate_mips left it (level k is widths[k] x heights[k], row-major, 4 bytes per texel;
 * the library copies the texels).  The id (0, 1, ... in call order) goes into the record's `reserved` field.
 * Textures stay loaded across frames until srt_raster_clear_textures.  Residency: clearing only forgets the SET; the texels stay
 * in the library's pinned copy and on the device, and a frame that re-adds the same levels in the same order (what DrawSVG's
 * redraw does every frame, drawsvg.cpp:462-474 having built the chains once) costs one compare per level and uploads nothing -
 * a stream of unchanged image records on an unchanged target then takes the identical-stream shortcut like any other.
 * Sampling is the reference's
 * Sampler2DImp::sample_trilinear (texture.cpp:171-193); texels it indexes past the end of a level (undefined in
 * the reference: column `width` / row `height` at the right / bottom border) read as zero. */
int srt_raster_add_texture(srt_raster* r, uint32_t nlevels, const uint32_t* widths, const uint32_t* heights,
                           const uint8_t* const* level_texels, uint32_t* id_out);
int srt_raster_clear_textures(srt_raster* r);
/* Diagnostic: bytes of texels uploaded to the device by this context so far (tests: a redraw of unchanged textures adds 0). */
int srt_raster_texture_upload_bytes(srt_raster* r, uint64_t* total);

/* Rasterize the pending stream in order, box-filter resolve, and write width*height RGBA8
 * (row 0 = top, row-major) into host memory `rgba8_out`. Synchronous: the buffer is complete on return. */
int srt_raster_resolve(srt_raster* r, uint8_t* rgba8_out);

/* Optional: tell the library that `host_rgba8` (bytes >= width*height*4) is the buffer srt_raster_resolve will be
 * handed from now on - DrawSVG lends one framebuffer per window size (set_render_target, drawsvg.cpp:107-114).  The
 * pages are pinned (hipHostRegister), so the read-back of a frame is one DMA transfer instead of a staged pageable
 * copy.  srt_raster_resolve works with any pointer; a bound one is only faster.  The registration is released by the next
 * bind (of another buffer or of NULL) or by srt_raster_destroy.  Preferably do that before the memory is freed; DrawSVG cannot
 * (DrawSVG::resize resizes its framebuffer vector BEFORE it calls set_render_target, drawsvg.cpp:111-114), and that order is
 * tolerated: releasing a registration removes the runtime's record of the pinned pages and never dereferences the range, and
 * srt_raster_resolve never uses a bound pointer other than the one it is handed.  What must not happen is a resolve INTO a
 * buffer that has been freed - which the reference's own renderer would not survive either. */
int srt_raster_bind_output(srt_raster* r, uint8_t* host_rgba8, size_t bytes);

/* Same as srt_raster_resolve but leaves the RGBA8 image in device memory (pointer valid until the next
 * set_target/destroy) and does not synchronize the stream; used by the resident-input benchmark.
 * `stream` is the hipStream_t to enqueue on (NULL = the HIP default stream, which is also PyTorch's default). */
int srt_raster_resolve_device(srt_raster* r, void* stream, const uint8_t** d_rgba8_out);

/* Optional: read back the supersample buffer (float RGBA, (width*sr)*(height*sr)*4 floats, row-major)
 * of the last resolve; parity tests compare it bit-for-bit with super_sample_buffer. */
int srt_raster_read_samples(srt_raster* r, float* samples_out);

/* Optional: counters of the last resolve (costs an extra device pass). */
int srt_raster_stats(srt_raster* r, srt_raster_stats_t* out);

/* The library keeps what it derived from the current stream on the device - bounding boxes, line tables, the ordered bin
 * lists - and a frame of an unchanged stream on an unchanged target runs the tile kernel alone (DrawSVG redraws on every
 * event).  srt_raster_invalidate makes the next frame derive them again: benchmarks use it to time a FULL frame
 * (setup + binning + tiles) of a resident stream. */
int srt_raster_invalidate(srt_raster* r);

/* Block until all work of the context has finished. */
int srt_raster_sync(srt_raster* r);

#ifdef __cplusplus
}
#endif
#endif /* SRT_RASTER_H */

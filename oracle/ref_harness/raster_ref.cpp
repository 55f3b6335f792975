// raster_ref.cpp — harness around the REFERENCE's own rasterizer, compiled from the sources where
// they lie under /root/reference (never copied into this repo; see oracle/Makefile, target `ref`).
//
// TEST INFRASTRUCTURE ONLY.  Output goes to oracle/_ref/libref_raster.so (git-ignored).  It is used
//   * by tests/golden/make_raster_golden.py to produce the committed fixtures, and
//   * by tests/test_raster_oracle.py (when /root/reference exists) to pin oracle/raster_oracle.c.
//
// What runs here is CMU462::SoftwareRendererImp (Assignments/DrawSVG/src/software_renderer.cpp) and
// the reference's SVG parser / triangulator / ViewportImp, unmodified.  `#define private public`
// is applied to this translation unit only, to reach rasterize_triangle / rasterize_point /
// super_sample_buffer; it does not change the reference's object code.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define private public
#include "software_renderer.h"
#undef private
#include "svg.h"
#include "viewport.h"

#include "svg_stream.h"  // the host half of our drop-in (no device dependency)
#include "srt_raster.h"

using namespace CMU462;

namespace {

// The framing DrawSVG applies at start-up: auto_adjust (drawsvg.cpp:476-483) and the
// norm_to_screen matrix of DrawSVG::resize (drawsvg.cpp:119-123), composed as in redraw (:440-444).
Matrix3x3 initial_svg_2_screen(const SVG& svg, size_t width, size_t height) {
  float w = svg.width, h = svg.height;
  float span = 1.2 * std::max(w, h) / 2;
  ViewportImp vp;
  vp.set_viewbox(w / 2, h / 2, span);
  Matrix3x3 norm_to_screen = Matrix3x3::identity();
  float scale = std::min(width, height);
  norm_to_screen(0, 0) = scale; norm_to_screen(0, 2) = (width - scale) / 2;
  norm_to_screen(1, 1) = scale; norm_to_screen(1, 2) = (height - scale) / 2;
  return norm_to_screen * vp.get_svg_2_norm();
}

}  // namespace

extern "C" {

// Render an SVG file with the reference renderer.  rgba_out: w*h*4 bytes.  samples_out (nullable):
// (w*sr)*(h*sr)*4 floats, the reference's super_sample_buffer after draw_svg.
int ref_raster_render_svg(const char* path, uint32_t w, uint32_t h, uint32_t sr, uint8_t* rgba_out,
                          float* samples_out) {
  SVG* svg = new SVG();  // leaked on purpose: Texture/Sampler destructors are not all defined in the reference
  if (SVGParser::load(path, svg) < 0) return -1;
  SoftwareRendererImp* ren = new SoftwareRendererImp();
  std::vector<unsigned char> fb(4 * (size_t)w * h);
  ren->set_render_target(fb.data(), w, h);
  ren->set_sample_rate(sr);
  ren->SoftwareRenderer::clear_target();  // DrawSVG::clear (drawsvg.cpp:274-281) calls the base version
  ren->set_svg_2_screen(initial_svg_2_screen(*svg, w, h));
  ren->draw_svg(*svg);
  std::memcpy(rgba_out, fb.data(), fb.size());
  if (samples_out)
    std::memcpy(samples_out, ren->super_sample_buffer.data(), ren->super_sample_buffer.size() * sizeof(float));
  return 0;
}

// The ordered primitive stream our host walk (SvgStreamBuilder::build) emits for the same
// SVG / framing.  Returns the number of primitives (writes at most cap of them), <0 on error.
long ref_raster_svg_stream(const char* path, uint32_t w, uint32_t h, uint32_t sr, srt_prim* out, size_t cap) {
  SVG* svg = new SVG();
  if (SVGParser::load(path, svg) < 0) return -1;
  SvgStreamBuilder builder;
  const std::vector<srt_prim>& s = builder.build(*svg, initial_svg_2_screen(*svg, w, h), sr);
  for (size_t i = 0; i < s.size() && i < cap; i++) out[i] = s[i];
  return (long)s.size();
}

// Feed a primitive stream straight into the reference's private rasterize_triangle / rasterize_point,
// then resolve.  This is the call sequence draw_svg makes, minus the SVG walk.
int ref_raster_prims(const srt_prim* prims, size_t n, uint32_t w, uint32_t h, uint32_t sr, uint8_t* rgba_out,
                     float* samples_out) {
  SoftwareRendererImp* ren = new SoftwareRendererImp();
  std::vector<unsigned char> fb(4 * (size_t)w * h);
  ren->set_render_target(fb.data(), w, h);
  ren->set_sample_rate(sr);
  ren->clear_target();  // Imp version: memset target + fill sample buffer with 255.0f
  for (size_t i = 0; i < n; i++) {
    const srt_prim& p = prims[i];
    Color c(p.rgba[0], p.rgba[1], p.rgba[2], p.rgba[3]);
    if (p.kind == SRT_PRIM_TRIANGLE)
      ren->rasterize_triangle(p.v.tri[0], p.v.tri[1], p.v.tri[2], p.v.tri[3], p.v.tri[4], p.v.tri[5], c);
    else if (p.kind == SRT_PRIM_POINT)
      ren->rasterize_point(p.v.point[0], p.v.point[1], c);
    else
      return -1;
  }
  if (samples_out)
    std::memcpy(samples_out, ren->super_sample_buffer.data(), ren->super_sample_buffer.size() * sizeof(float));
  ren->resolve();
  std::memcpy(rgba_out, fb.data(), fb.size());
  delete ren;
  return 0;
}

}  // extern "C"

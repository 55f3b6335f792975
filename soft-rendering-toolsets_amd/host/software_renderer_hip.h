// SoftwareRendererHIP — drop-in CMU462::SoftwareRenderer whose triangle fill,
// point fill and resolve run on an MI355X through the C ABI in include/srt_raster.h.
//
// This header is compiled INSIDE the reference tree (it includes the reference's own
// software_renderer.h); see INTEGRATION.md for the one-line swap at
// Assignments/DrawSVG/src/drawsvg.cpp:55.  It keeps the public surface of
// SoftwareRendererImp (Assignments/DrawSVG/src/software_renderer.h:78-98):
// draw_svg / set_sample_rate / set_render_target / clear_target.
//
// Split of work:
//   host  : element walk, transform stack, ear-clip triangulation (the reference's own
//           triangulate()), Xiaolin-Wu line decomposition into rasterize_point blocks
//   device: rasterize_triangle + inside_triangle + fill_sample + rasterize_point + resolve
#ifndef SRT_SOFTWARE_RENDERER_HIP_H
#define SRT_SOFTWARE_RENDERER_HIP_H

#include <vector>

#include "software_renderer.h"  // the reference's header (CMU462::SoftwareRenderer)
#include "srt_raster.h"

namespace CMU462 {

class SoftwareRendererHIP : public SoftwareRenderer {
 public:
  // connect_device=false builds the ordered stream only (used by the fixture generator,
  // which has no GPU); any draw_svg call then fails loudly.
  explicit SoftwareRendererHIP(int device = 0, bool connect_device = true);
  ~SoftwareRendererHIP();

  void draw_svg(SVG& svg);
  void set_sample_rate(size_t sample_rate);
  void set_render_target(unsigned char* target_buffer, size_t width, size_t height);

  // Mirrors SoftwareRendererImp::clear_target (software_renderer.h:93-98).
  void clear_target();

  // Host half of draw_svg: walks the SVG and returns the ordered primitive stream that
  // draw_svg hands to srt_raster_submit. Exposed so the stream can be captured as a fixture.
  const std::vector<srt_prim>& build_stream(SVG& svg);

  // Number of <image> elements skipped by the last build_stream (unsupported on this path).
  size_t skipped_images() const { return skipped_images_; }

 private:
  void walk(SVGElement* element);
  void emit_triangle(float x0, float y0, float x1, float y1, float x2, float y2, const Color& c);
  void emit_point(double x, double y, const Color& c);
  void emit_line(float x0, float y0, float x1, float y1, Color c);

  srt_raster* ctx_;
  std::vector<srt_prim> stream_;
  size_t skipped_images_;
};

}  // namespace CMU462

#endif

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
//           triangulate()) (svg_stream.cpp)
//   device: rasterize_triangle + inside_triangle + fill_sample + rasterize_line_xiaolinwu + rasterize_point + rasterize_image
//           (Sampler2DImp::sample_trilinear over the application's mip chains) + resolve
#ifndef SRT_SOFTWARE_RENDERER_HIP_H
#define SRT_SOFTWARE_RENDERER_HIP_H

#include <string>
#include <vector>

#include "software_renderer.h"  // the reference's header (CMU462::SoftwareRenderer)
#include "srt_raster.h"
#include "svg_stream.h"

namespace CMU462 {

class SoftwareRendererHIP : public SoftwareRenderer {
 public:
  explicit SoftwareRendererHIP(int device = 0);
  ~SoftwareRendererHIP();

  void draw_svg(SVG& svg);
  void set_sample_rate(size_t sample_rate);
  void set_render_target(unsigned char* target_buffer, size_t width, size_t height);

  // Mirrors SoftwareRendererImp::clear_target (software_renderer.h:93-98).
  void clear_target();

  // Frames draw_svg dropped because the device path refused their content (see frame_refused in the .cpp); the reason of the last one.
  size_t refused_frames() const { return refused_frames_; }
  const std::string& last_refusal() const { return last_refusal_; }

 private:
  bool frame_refused(int status, const char* what);
  srt_raster* ctx_;
  size_t refused_frames_;
  std::string last_refusal_;
  SvgStreamBuilder builder_;  // host half of draw_svg
};

}  // namespace CMU462

#endif

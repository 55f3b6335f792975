// SvgStreamBuilder — the HOST half of SoftwareRendererImp::draw_svg: walks an SVG and emits the ordered
// primitive stream (include/srt_raster.h: srt_prim) that the device rasterizes.  No device dependency:
// this file needs only the reference's headers and srt_raster.h's record layout.
//
// Compiled INSIDE the reference tree (Assignments/DrawSVG/src); see INTEGRATION.md.
#ifndef SRT_SVG_STREAM_H
#define SRT_SVG_STREAM_H

#include <map>
#include <vector>

#include "svg_renderer.h"  // the reference's SVGRenderer (transform stack helpers)
#include "srt_raster.h"

namespace CMU462 {

class SvgStreamBuilder : public SVGRenderer {
 public:
  SvgStreamBuilder() : sample_rate_(1), reference_transforms_(false), last_size_(0) {}

  // true: every point goes through SVGRenderer::transform itself, corner by corner, as in the reference's draw_* functions
  // (the checker's setting; the default computes the same doubles inline and once per polygon point - see svg_stream.cpp)
  void set_reference_transforms(bool on) { reference_transforms_ = on; }

  // SVGRenderer interface: same as build().
  void draw_svg(SVG& svg) { build(svg, transformation, sample_rate_); }

  // Walk `svg` with top-level transform svg_2_screen.  (sample_rate no longer enters: lines are single records, their
  // Xiaolin-Wu expansion - whose loop bound depends on it, software_renderer.cpp:434,445 - happens on the device.)
  const std::vector<srt_prim>& build(SVG& svg, const Matrix3x3& svg_2_screen, size_t sample_rate);

  const std::vector<srt_prim>& stream() const { return stream_; }
  // Textures of the <image> elements met by the last build, in walk order: an SRT_PRIM_IMAGE record's `reserved`
  // field indexes this list.  The mip chains are the application's (DrawSVG::regenerate_mipmap ->
  // Sampler2D::generate_mips); the caller uploads them with srt_raster_add_texture before submitting the stream.
  const std::vector<const Texture*>& textures() const { return textures_; }

 private:
  void walk(SVGElement* element);
  void emit_triangle(float x0, float y0, float x1, float y1, float x2, float y2, const Color& c);
  void emit_point(double x, double y, const Color& c);
  void emit_line(float x0, float y0, float x1, float y1, Color c);
  void emit_image(float x0, float y0, float x1, float y1, const Texture& tex);

  // triangulate() results per polygon element, valid while the polygon's points are what they were (see walk())
  // (index: per corner of tris, which of the polygon's points it is; empty if some corner is none of them)
  struct CachedTriangulation { bool valid; std::vector<Vector2D> points, tris; std::vector<uint32_t> index; CachedTriangulation() : valid(false) {} };
  const CachedTriangulation& triangulation_of(const Polygon& pg);
  std::map<const Polygon*, CachedTriangulation> tri_cache_;
  struct PointMap;
  const std::vector<Vector2D>& transformed(const std::vector<Vector2D>& points);
  std::vector<Vector2D> scratch_;
  bool reference_transforms_;
  size_t last_size_;

  size_t sample_rate_;
  std::vector<srt_prim> stream_;
  std::vector<const Texture*> textures_;
};

}  // namespace CMU462

#endif

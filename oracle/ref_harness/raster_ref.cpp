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
#include "texture.h"
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

// Sampler2DImp::sample_bilinear indexes one column / one row past a level at the right / bottom border
// (texture.cpp:158-166): undefined in the reference.  The harness gives every texel vector zeroed slack behind
// its end (capacity, not size), so that those reads see zeros here, as they do in the oracle and on the GPU.
void pad_level(MipLevel& l) {
  const size_t n = l.texels.size(), slack = 4 * (l.width + 2);
  l.texels.reserve(n + slack);
  std::memset(l.texels.data() + n, 0, slack);
}

void collect_images(SVGElement* e, std::vector<Image*>& out) {
  if (e->type == IMAGE) out.push_back(static_cast<Image*>(e));
  if (e->type == GROUP) {
    Group& g = static_cast<Group&>(*e);
    for (size_t i = 0; i < g.elements.size(); ++i) collect_images(g.elements[i], out);
  }
}

// What DrawSVG does before the first redraw: a trilinear sampler on the renderer (drawsvg.cpp:58-62) and
// generate_mips for every image texture (regenerate_mipmap, drawsvg.cpp:462-474).
void prepare_textures(SVG& svg, SoftwareRendererImp* ren) {
  Sampler2DImp* sampler = new Sampler2DImp();  // never deleted: ~Sampler2D is declared but not defined
  ren->set_tex_sampler(sampler);
  std::vector<Image*> images;
  for (size_t i = 0; i < svg.elements.size(); ++i) collect_images(svg.elements[i], images);
  for (Image* im : images) {
    sampler->generate_mips(im->tex, 0);
    for (MipLevel& l : im->tex.mipmap) pad_level(l);
  }
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
  prepare_textures(*svg, ren);
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

// The host walk's own shortcuts (points transformed inline and once per polygon point, svg_stream.cpp) against the same walk
// with every corner sent through the reference's SVGRenderer::transform: 0 if the two streams are byte-identical, else
// 1 + the index of the first record that differs (or 1 + the shorter length).  Two views: the initial framing and a skewed,
// shifted one (`variant` 1) so that the matrix entries are not round numbers.
long ref_raster_svg_stream_ab(const char* path, uint32_t w, uint32_t h, uint32_t sr, uint32_t variant) {
  SVG* svg = new SVG();
  if (SVGParser::load(path, svg) < 0) return -1;
  Matrix3x3 m = initial_svg_2_screen(*svg, w, h);
  if (variant) { m(0, 0) *= 1.0 / 3.0; m(0, 1) += 0.1234567; m(1, 0) -= 0.0714285; m(0, 2) += 17.3; m(1, 2) -= 5.7; m(2, 0) += 1e-4; m(2, 2) = 0.93; }
  SvgStreamBuilder fast, slow;
  slow.set_reference_transforms(true);
  for (int pass = 0; pass < 2; pass++) {   // (the second pass runs on warm triangulation caches)
    const std::vector<srt_prim>& a = fast.build(*svg, m, sr);
    const std::vector<srt_prim>& b = slow.build(*svg, m, sr);
    const size_t n = a.size() < b.size() ? a.size() : b.size();
    for (size_t i = 0; i < n; i++)
      if (std::memcmp(&a[i], &b[i], sizeof(srt_prim)) != 0) return (long)i + 1;
    if (a.size() != b.size()) return (long)n + 1;
  }
  return 0;
}

// The textures the stream of the same SVG refers to (SvgStreamBuilder::textures() order), with the mip chains the
// reference's Sampler2DImp::generate_mips builds.  tex_nlevels[<= max_tex]; level_w / level_h / level_off per level,
// concatenated over the textures; blob receives the texels.  Returns the number of textures, <0 on error / overflow.
long ref_raster_svg_textures(const char* path, uint32_t w, uint32_t h, uint32_t sr, uint32_t max_tex, uint32_t* tex_nlevels,
                             uint32_t* level_w, uint32_t* level_h, uint64_t* level_off, uint8_t* blob, uint64_t blob_cap) {
  SVG* svg = new SVG();
  if (SVGParser::load(path, svg) < 0) return -1;
  SoftwareRendererImp* ren = new SoftwareRendererImp();
  prepare_textures(*svg, ren);
  SvgStreamBuilder builder;
  builder.build(*svg, initial_svg_2_screen(*svg, w, h), sr);
  const std::vector<const Texture*>& tex = builder.textures();
  if (tex.size() > max_tex) return -2;
  uint64_t off = 0;
  size_t l = 0;
  for (size_t t = 0; t < tex.size(); t++) {
    tex_nlevels[t] = (uint32_t)tex[t]->mipmap.size();
    for (const MipLevel& lv : tex[t]->mipmap) {
      if (off + lv.texels.size() > blob_cap) return -3;
      level_w[l] = (uint32_t)lv.width; level_h[l] = (uint32_t)lv.height; level_off[l] = off;
      std::memcpy(blob + off, lv.texels.data(), lv.texels.size());
      off += lv.texels.size();
      l++;
    }
  }
  return (long)tex.size();
}

// Feed a primitive stream straight into the reference's private rasterize_triangle / rasterize_line / rasterize_point / rasterize_image,
// then resolve.  This is the call sequence draw_svg makes, minus the SVG walk.
int ref_raster_prims_tex(const srt_prim* prims, size_t n, uint32_t w, uint32_t h, uint32_t sr, uint32_t ntex,
                         const uint32_t* tex_nlevels, const uint32_t* level_w, const uint32_t* level_h,
                         const uint64_t* level_off, const uint8_t* blob, uint8_t* rgba_out, float* samples_out) {
  std::vector<Texture> textures(ntex);
  for (uint32_t t = 0, l = 0; t < ntex; t++) {
    textures[t].mipmap.resize(tex_nlevels[t]);
    for (uint32_t k = 0; k < tex_nlevels[t]; k++, l++) {
      MipLevel& lv = textures[t].mipmap[k];
      lv.width = level_w[l]; lv.height = level_h[l];
      lv.texels.assign(blob + level_off[l], blob + level_off[l] + 4 * (size_t)level_w[l] * level_h[l]);
      pad_level(lv);
    }
    textures[t].width = textures[t].mipmap[0].width;
    textures[t].height = textures[t].mipmap[0].height;
  }
  SoftwareRendererImp* ren = new SoftwareRendererImp();
  ren->set_tex_sampler(new Sampler2DImp());
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
    else if (p.kind == SRT_PRIM_LINE)
      ren->rasterize_line(p.v.tri[0], p.v.tri[1], p.v.tri[2], p.v.tri[3], c);   // = rasterize_line_xiaolinwu (cpp:303-318)
    else if (p.kind == SRT_PRIM_IMAGE && p.reserved < ntex)
      ren->rasterize_image(p.v.tri[0], p.v.tri[1], p.v.tri[2], p.v.tri[3], textures[p.reserved]);
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

int ref_raster_prims(const srt_prim* prims, size_t n, uint32_t w, uint32_t h, uint32_t sr, uint8_t* rgba_out,
                     float* samples_out) {
  return ref_raster_prims_tex(prims, n, w, h, sr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, rgba_out, samples_out);
}

// Sampler2DImp::generate_mips on a bare level-0 image: pins the oracle's restatement.  Same output format as
// ref_raster_svg_textures for one texture.  Returns the number of levels.
long ref_raster_generate_mips(const uint8_t* level0, uint32_t w0, uint32_t h0, uint32_t* level_w, uint32_t* level_h,
                              uint8_t* blob, uint64_t blob_cap) {
  Texture tex;
  tex.width = w0; tex.height = h0;
  tex.mipmap.resize(1);
  tex.mipmap[0].width = w0; tex.mipmap[0].height = h0;
  tex.mipmap[0].texels.assign(level0, level0 + 4 * (size_t)w0 * h0);
  pad_level(tex.mipmap[0]);
  Sampler2DImp* sampler = new Sampler2DImp();
  sampler->generate_mips(tex, 0);
  uint64_t off = 0;
  for (size_t k = 0; k < tex.mipmap.size(); k++) {
    const MipLevel& lv = tex.mipmap[k];
    if (off + lv.texels.size() > blob_cap) return -1;
    level_w[k] = (uint32_t)lv.width; level_h[k] = (uint32_t)lv.height;
    std::memcpy(blob + off, lv.texels.data(), lv.texels.size());
    off += lv.texels.size();
  }
  return (long)tex.mipmap.size();
}

}  // extern "C"

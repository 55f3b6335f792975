// See software_renderer_hip.h.  The class glue between CMU462::SoftwareRenderer's surface and the C ABI;
// the SVG walk itself lives in svg_stream.cpp.
#include "software_renderer_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace CMU462 {

namespace {

void die_on(int status, const char* what) {
  if (status != SRT_OK) {
    std::fprintf(stderr, "[SoftwareRendererHIP] %s failed (%d): %s\n", what, status, srt_last_error());
    std::abort();  // the reference's surface has void returns and no error channel
  }
}

}  // namespace

// A frame the device path REFUSES (SRT_ERR_UNSUPPORTED: content whose result the reference itself never produces - a line whose main
// loop runs past 2^24, where `++x` on a float stops advancing and the reference hangs - or that exceeds a table limit) is dropped:
// the reason goes to stderr (once per distinct reason), the target is left as clear_target() leaves it - white - and the
// application keeps running; the next frame starts from srt_raster_clear like any other.  Nothing is drawn by another path.  Every other failure (no
// device, a HIP error, a misuse of the ABI) still ends the process: the reference's surface has no error channel.
bool SoftwareRendererHIP::frame_refused(int status, const char* what) {
  if (status != SRT_ERR_UNSUPPORTED) { die_on(status, what); return false; }
  const char* why = srt_last_error();
  if (last_refusal_ != why) {
    last_refusal_ = why;
    std::fprintf(stderr, "[SoftwareRendererHIP] frame dropped, %s refused it: %s\n", what, why);
  }
  refused_frames_++;
  if (render_target) std::memset(render_target, 255, 4 * target_w * target_h);
  return true;
}

SoftwareRendererHIP::SoftwareRendererHIP(int device) : SoftwareRenderer(), ctx_(nullptr), refused_frames_(0) {
  render_target = nullptr;
  target_w = target_h = 0;
  die_on(srt_raster_create(device, &ctx_), "srt_raster_create");  // aborts without a HIP device: no CPU path
}

SoftwareRendererHIP::~SoftwareRendererHIP() {
  if (ctx_) {
    srt_raster_bind_output(ctx_, nullptr, 0);   // unpin DrawSVG's framebuffer while it is still alive
    srt_raster_destroy(ctx_);
  }
}

void SoftwareRendererHIP::set_sample_rate(size_t rate) {
  if (this->sample_rate == rate) return;  // software_renderer.cpp:61-62
  this->sample_rate = rate;
  if (ctx_ && target_w && target_h)
    die_on(srt_raster_set_target(ctx_, (uint32_t)target_w, (uint32_t)target_h, (uint32_t)rate),
           "srt_raster_set_target");
}

void SoftwareRendererHIP::set_render_target(unsigned char* target_buffer, size_t width, size_t height) {
  render_target = target_buffer;
  target_w = width;
  target_h = height;
  if (ctx_) {
    die_on(srt_raster_set_target(ctx_, (uint32_t)width, (uint32_t)height, (uint32_t)sample_rate),
           "srt_raster_set_target");
    // DrawSVG lends ONE framebuffer per window size (drawsvg.cpp:107-114): pin it, so that every frame's read-back is one DMA
    // transfer.  (Binding drops the previous buffer's registration; DrawSVG::resize has already resized the vector by now.)
    die_on(srt_raster_bind_output(ctx_, target_buffer, 4 * width * height), "srt_raster_bind_output");
  }
}

void SoftwareRendererHIP::clear_target() {
  if (render_target) std::memset(render_target, 255, 4 * target_w * target_h);
  if (ctx_) die_on(srt_raster_clear(ctx_), "srt_raster_clear");
}

void SoftwareRendererHIP::draw_svg(SVG& svg) {
  if (!ctx_) {
    std::fprintf(stderr, "[SoftwareRendererHIP] draw_svg without a device context; there is no CPU path\n");
    std::abort();
  }
  // SoftwareRendererImp::draw_svg starts with clear_target() (cpp:19); the render target's part of it - a 4 MiB memset at
  // 1024^2 - is skipped here because srt_raster_resolve below writes every byte of it
  die_on(srt_raster_clear(ctx_), "srt_raster_clear");
  const std::vector<srt_prim>& stream = builder_.build(svg, svg_2_screen, sample_rate);
  // <image> textures: the mip chains DrawSVG::regenerate_mipmap built with the application's sampler
  die_on(srt_raster_clear_textures(ctx_), "srt_raster_clear_textures");
  for (const Texture* tex : builder_.textures()) {
    uint32_t w[SRT_MAX_MIP_LEVELS], h[SRT_MAX_MIP_LEVELS], id = 0;
    const uint8_t* lv[SRT_MAX_MIP_LEVELS];
    const size_t n = tex->mipmap.size() < (size_t)SRT_MAX_MIP_LEVELS ? tex->mipmap.size() : (size_t)SRT_MAX_MIP_LEVELS;
    for (size_t k = 0; k < n; k++) {
      w[k] = (uint32_t)tex->mipmap[k].width; h[k] = (uint32_t)tex->mipmap[k].height; lv[k] = tex->mipmap[k].texels.data();
    }
    die_on(srt_raster_add_texture(ctx_, (uint32_t)n, w, h, lv, &id), "srt_raster_add_texture");
  }
  if (frame_refused(srt_raster_submit(ctx_, stream.data(), stream.size()), "srt_raster_submit")) return;
  if (frame_refused(srt_raster_resolve(ctx_, render_target), "srt_raster_resolve")) return;
}

}  // namespace CMU462

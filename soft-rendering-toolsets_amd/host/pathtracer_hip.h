// PT::Pathtracer — drop-in replacement of the reference's class declaration
// (Assignments/Scotty3D/src/rays/pathtracer.h) whose do_trace runs on MI355X GPUs through the C ABI of
// include/srt_pt.h.  The PUBLIC surface is the reference's, signature for signature (pathtracer.h:24-40), so
// Gui::Widget_Render (gui/widgets.h:131, gui/widgets.cpp:788-968) compiles and behaves unchanged:
// begin_render is asynchronous, progress()/in_progress() are polled, get_output()/get_output_texture() hand out
// the running-mean accumulator, cancel() stops the launches in flight within milliseconds (srt_pt_cancel), "Add Samples"
// keeps the accumulator, and the rays log_ray's coin selects reach gui.log_ray.
//
// To integrate: build this header/implementation INSTEAD of rays/pathtracer.{h,cpp} and student/pathtracer.cpp
// (INTEGRATION.md).  There is no CPU path behind it: without a HIP device the constructor dies like the
// reference's die() does on misuse.
#pragma once

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../lib/mathlib.h"
#include "../scene/scene.h"
#include "../util/camera.h"
#include "../util/hdr_image.h"
#include "../util/thread_pool.h"

// kept because other GUI headers rely on rays/pathtracer.h pulling these in (e.g. gui/simulate.h uses PT::Object)
#include "bsdf.h"
#include "env_light.h"
#include "light.h"
#include "object.h"

#include "srt_pt.h"
#include "pathtracer_core.h"

namespace Gui {
class Widget_Render;
}

namespace PT {

class Pathtracer {
public:
    Pathtracer(Gui::Widget_Render& gui, Vec2 screen_dim);
    ~Pathtracer();

    void set_params(size_t w, size_t h, size_t pixel_samples, size_t depth, bool use_bvh);
    void set_samples(size_t samples);

    const HDR_Image& get_output();
    const GL::Tex2D& get_output_texture(float exposure);
    // The accumulator's display bytes, HDR_Image::tonemap_to (util/hdr_image.cpp:161-187) evaluated on the GPU
    // (srt_pt_tonemap); exposure <= 0 keeps the last one, as HDR_Image::tonemap does.  widgets.cpp:719,863,967 call
    // pathtracer.get_output().tonemap_to(data, exposure); with the drop-in they call pathtracer.tonemap_to(data, exposure).
    void tonemap_to(std::vector<unsigned char>& data, float exposure = 0.0f);
    size_t visualize_bvh(GL::Lines& lines, GL::Lines& active, size_t level);

    void begin_render(Scene& scene, const Camera& camera, bool add_samples = false);
    void cancel();
    bool in_progress() const;
    float progress() const;
    std::pair<float, float> completion_time() const;

    // Not in the reference: RNG seed of the next render (the reference is unseeded).
    void set_seed(unsigned long long s) { core.set_seed(s); }

private:
    static void deliver_logged_rays(void* self, const srt_pt_logged_ray* rays, size_t n);   // Pathtracer::log_ray -> gui.log_ray
    void build_scene(Scene& scene);              // rays/pathtracer.cpp:66-176 -> srt_pt_scene_* on every device's context
    void feed_scene(srt_pt* ctx, Scene& scene);  // the object / light / particle walk for one context

    Gui::Widget_Render& gui;
    // Everything that does not need Scene: a context group - device 0 by default, more GPUs through SRT_PT_DEVICES=<count> (image tiles + one RCCL gather per epoch), the
    // epoch scheme, the worker, the running mean, cancel / progress, the display epilogue (pathtracer_core.h).
    srt_host::RenderCore core;

    HDR_Image accumulator;           // what get_output() hands out: refreshed from the core's running mean
    std::vector<float> accumulator_copy;
    GL::Tex2D output_tex;            // display texture fed by tonemap_to
    std::vector<unsigned char> tonemap_out;
    bool scene_use_bvh = true;
    // BVH<Object> node arrays of the last build_scene, kept for visualize_bvh (no context call while a render is in flight)
    std::vector<float> bvh_boxes;
    std::vector<uint32_t> bvh_links;
};

} // namespace PT

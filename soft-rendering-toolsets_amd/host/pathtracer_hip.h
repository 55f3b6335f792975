// PT::Pathtracer — drop-in replacement of the reference's class declaration
// (Assignments/Scotty3D/src/rays/pathtracer.h) whose do_trace runs on MI355X GPUs through the C ABI of
// include/srt_pt.h.  The PUBLIC surface is the reference's, signature for signature (pathtracer.h:24-40), so
// Gui::Widget_Render (gui/widgets.h:131, gui/widgets.cpp:788-968) compiles and behaves unchanged:
// begin_render is asynchronous, progress()/in_progress() are polled, get_output()/get_output_texture() hand out
// the running-mean accumulator, cancel() stops between epochs, "Add Samples" keeps the accumulator.
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

    // Not in the reference: RNG seed of the next render (the reference is unseeded) and the GPU to use.
    void set_seed(unsigned long long s) { seed = s; }

private:
    void build_scene(Scene& scene);   // rays/pathtracer.cpp:66-176 -> srt_pt_scene_*
    void accumulate(const float* epoch);  // rays/pathtracer.cpp:195-207
    void worker(size_t samples_per_epoch, size_t first_sample);

    Gui::Widget_Render& gui;
    srt_pt* ctx = nullptr;
    std::thread render_thread;
    std::atomic<bool> cancel_flag{false};

    HDR_Image accumulator;
    GL::Tex2D output_tex;            // display texture fed by tonemap_to
    float display_exposure = 1.0f;   // HDR_Image::exposure
    std::vector<float> tonemap_in;
    std::vector<unsigned char> tonemap_out;
    std::mutex accumulator_mut;
    size_t total_epochs = 0, accumulator_samples = 0;
    std::atomic<size_t> completed_epochs{0};
    size_t samples_done = 0;   // sample index the next epoch starts at ("Add Samples" continues it)

    std::chrono::steady_clock::time_point t_build0, t_render0;
    std::atomic<long long> build_ns{0}, render_ns{0};

    bool scene_use_bvh = true;
    size_t out_w = 0, out_h = 0, n_samples = 0, max_depth = 0;
    unsigned long long seed = 0;
    std::vector<float> epoch_buf;
};

} // namespace PT

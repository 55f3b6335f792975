// RenderCore — everything of the drop-in PT::Pathtracer that does not need the reference's Scene / GUI types: the
// epoch scheme of Pathtracer::begin_render (rays/pathtracer.cpp:250-280), the worker that stands in for the thread pool's
// do_trace tasks (:209-231 -> srt_pt_group_render_epoch), the running-mean accumulate (:195-207), cancel / progress /
// completion_time (:282-294), "Add Samples", and the display epilogue (HDR_Image::tonemap_to on the GPU).
//
// It owns a group of device contexts (srt_pt_create_multi): with more than one visible GPU the image tiles of every epoch
// are spread over them and gathered with one RCCL collective; with one GPU the group has one member.  The scene walk stays
// with the class that knows the reference's Scene (pathtracer_hip.cpp) and feeds every member through the C ABI.
//
// Threads (include/srt_pt.h: a context is used from one host thread at a time): the worker is the only thread that touches
// the render contexts while a render is in flight; scene / camera / parameter calls happen between renders (begin() joins
// the previous worker first).  The GUI thread's display path never touches them: tonemap() copies the accumulator under
// its mutex and runs on a context of its own, so it waits neither for an epoch in flight nor with the accumulator locked.
// No reference header is included here: this file and pathtracer_core.cpp build and run on their own
// (tests/host_emu/pt_core_driver.cpp).
#pragma once

#include <atomic>
#include <chrono>
#include <cstdint>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "srt_pt.h"

namespace srt_host {

class RenderCore {
public:
    // devices == nullptr: device 0 only; SRT_PT_DEVICES=n opts in to the first n visible HIP devices (the distinct-device gather).  Aborts through `fatal` when there is
    // no device: the path has no CPU fallback.
    RenderCore(const int* devices, int n, void (*fatal)(const char* what, int status, const char* message));
    ~RenderCore();
    RenderCore(const RenderCore&) = delete;
    RenderCore& operator=(const RenderCore&) = delete;

    int ranks() const { return (int)members.size(); }
    srt_pt* context(int rank) const { return members[rank]; }      // for the owner's scene walk (no render in flight)

    void set_params(size_t w, size_t h, size_t pixel_samples, size_t depth);
    void set_samples(size_t samples) { n_samples = samples; }
    void set_seed(uint64_t s) { seed = s; }
    void set_threads(size_t n) { n_threads = n; }                  // the reference's hardware_concurrency(), which sets the epoch size

    // begin_render minus the scene walk.  add_samples keeps the accumulator and continues the sample index.
    void begin(const float iview[16], float vert_fov_deg, float aspect_ratio, bool add_samples);
    void cancel();
    bool in_progress() const { return completed_epochs.load() < total_epochs; }
    float progress() const { return (float)completed_epochs.load() / (float)total_epochs; }
    void wait();                                                    // join the worker (headless use, tests)
    std::pair<float, float> completion_time() const { return {(float)(build_ns.load() * 1e-9), (float)(render_ns.load() * 1e-9)}; }
    void note_build_time(long long ns) { build_ns = ns; }

    size_t width() const { return out_w; }
    size_t height() const { return out_h; }
    size_t epochs_accumulated() const { return accumulator_samples; }
    // the running mean so far, w*h*3 floats, row 0 = bottom (HDR_Image order); copied under the accumulator's mutex
    void copy_accumulator(std::vector<float>& out);
    // HDR_Image::tonemap_to (util/hdr_image.cpp:161-187) of the accumulator on the GPU: w*h*4 bytes, rows flipped for
    // display; exposure <= 0 keeps the previous one, as HDR_Image::tonemap does
    void tonemap(std::vector<unsigned char>& data, float exposure);

private:
    void worker(size_t samples_per_epoch, size_t first_sample);
    void accumulate(const float* epoch);
    void check(int status, const char* what) const;

    void (*fatal)(const char*, int, const char*);
    srt_pt_group* group = nullptr;
    std::vector<srt_pt*> members;
    srt_pt* display_ctx = nullptr;           // tone mapping only: its own stream, never the render contexts
    std::thread render_thread;
    std::atomic<bool> cancel_flag{false};

    std::vector<float> accumulator;          // w*h*3
    std::mutex accumulator_mut;
    size_t total_epochs = 0, accumulator_samples = 0;
    std::atomic<size_t> completed_epochs{0};
    size_t samples_done = 0;                 // sample index the next render starts at ("Add Samples" continues it)
    float display_exposure = 1.0f;           // HDR_Image::exposure
    std::vector<float> tonemap_in;

    std::chrono::steady_clock::time_point t_render0;
    std::atomic<long long> build_ns{0}, render_ns{0};
    size_t out_w = 0, out_h = 0, n_samples = 0, max_depth = 0, n_threads = 0;
    uint64_t seed = 0;
    std::vector<float> epoch_buf;
};

}  // namespace srt_host

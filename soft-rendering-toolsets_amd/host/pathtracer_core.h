// RenderCore — everything of the drop-in PT::Pathtracer that does not need the reference's Scene / GUI types: the
// epoch scheme of Pathtracer::begin_render (rays/pathtracer.cpp:250-280), the worker that stands in for the thread pool's
// do_trace tasks (:209-231), the running-mean accumulate (:195-207), cancel / progress / completion_time (:282-294), "Add
// Samples", the ray log (:191-193) and the display epilogue (HDR_Image::tonemap_to on the GPU).
//
// How a render runs (round 4).  The reference's epochs are small - samples_per_epoch = max(1, n / (threads * 10)), ONE sample for
// a 256-spp render on a 32-thread host - and a launch per epoch, a 12 MiB read-back and a host-side running mean over 3 M floats
// per epoch left the GPU idle most of the time.  Now the worker renders LAUNCHES of up to 64 samples per pixel
// (srt_pt_group_render_samples), two in flight on two lanes so that one launch's tail is filled by the next, and after each
// launch has completed un-cancelled enqueues the fold (srt_pt_group_fold) that replays do_trace's epoch means and accumulate's
// running mean over the launch's samples, epoch by epoch, into an accumulator that LIVES ON THE DEVICES - every rank folds its
// own tiles, nothing is exchanged per epoch.  The image comes to the host only when somebody asks (copy_accumulator / tonemap:
// one gather, one un-tiling, one copy).  Bit-identical to an epoch-by-epoch render; progress() advances a launch at a time.
//
// It owns a group of device contexts (srt_pt_create_multi): with more than one GPU (opt-in) the image tiles are spread over
// them; with one GPU the group has one member.  The scene walk stays with the class that knows the reference's Scene
// (pathtracer_hip.cpp) and feeds every member through the C ABI.
//
// Threads: the worker is the only thread that renders; scene / camera / parameter calls happen between renders (begin() joins
// the previous worker first).  The GUI thread's display path (copy_accumulator, tonemap) only enqueues on the group's display
// lane and is serialised against the worker's folds by the accumulator's mutex; cancel() may come from any thread.
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
    // Pathtracer::log_ray's sink: called on the render thread after every launch with the rays the 0.0005 coin selected, in log
    // order (pixel, sample, bounce).  capacity = rays kept per launch and rank (0: the log is off).
    void set_ray_log(void (*sink)(void* user, const srt_pt_logged_ray* rays, size_t n), void* user, uint32_t capacity = 1u << 16);

    size_t width() const { return out_w; }
    size_t height() const { return out_h; }
    size_t epochs_accumulated() const { return accumulator_samples.load(); }
    // the running mean so far, w*h*3 floats, row 0 = bottom (HDR_Image order); copied under the accumulator's mutex
    void copy_accumulator(std::vector<float>& out);
    // HDR_Image::tonemap_to (util/hdr_image.cpp:161-187) of the accumulator on the GPU: w*h*4 bytes, rows flipped for
    // display; exposure <= 0 keeps the previous one, as HDR_Image::tonemap does
    void tonemap(std::vector<unsigned char>& data, float exposure);

private:
    void worker(size_t samples_per_epoch, size_t first_sample, size_t first_epochs);
    void check(int status, const char* what) const;

    void (*fatal)(const char*, int, const char*);
    srt_pt_group* group = nullptr;
    std::vector<srt_pt*> members;
    std::thread render_thread;
    std::atomic<bool> cancel_flag{false};

    std::mutex accumulator_mut;              // the device accumulator: the worker's folds against the display path's reads
    size_t total_epochs = 0;
    std::atomic<size_t> accumulator_samples{0};   // epochs folded so far (Pathtracer::accumulator_samples)
    std::atomic<size_t> completed_epochs{0};
    void (*ray_sink)(void*, const srt_pt_logged_ray*, size_t) = nullptr;
    void* ray_sink_user = nullptr;
    std::vector<srt_pt_logged_ray> ray_buf;
    unsigned char* d_rgba = nullptr; size_t rgba_bytes = 0;   // tone-mapped bytes on rank 0's device
    size_t samples_done = 0;                 // sample index the next render starts at ("Add Samples" continues it)
    float display_exposure = 1.0f;           // HDR_Image::exposure

    std::chrono::steady_clock::time_point t_render0;
    std::atomic<long long> build_ns{0}, render_ns{0};
    size_t out_w = 0, out_h = 0, n_samples = 0, max_depth = 0, n_threads = 0;
    uint64_t seed = 0;
    int device0 = 0;
};

}  // namespace srt_host

// See pathtracer_core.h.
#include "pathtracer_core.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime_api.h>

namespace srt_host {

void RenderCore::check(int status, const char* what) const {
    if(status != SRT_OK) fatal(what, status, srt_last_error());
}

RenderCore::RenderCore(const int* devices, int n, void (*fatal_)(const char*, int, const char*)) : fatal(fatal_) {
    std::vector<int> all;
    if(!devices) {
        // Default: ONE device (device 0).  More devices are opt-in - the constructor's list or SRT_PT_DEVICES=<count> - because that
        // routes every epoch through srt_pt_group's distinct-device gather (ncclCommInitAll + grouped ncclGather), which no
        // hardware run has shown bit-equal to the single context yet (Multi-GPU).
        int count = 0;
        if(hipGetDeviceCount(&count) != hipSuccess || count <= 0) fatal("hipGetDeviceCount", SRT_ERR_NO_DEVICE, "no HIP device: this path has no CPU fallback");
        int use = 1;
        if(const char* lim = getenv("SRT_PT_DEVICES")) use = std::max(1, std::min(count, atoi(lim)));
        for(int d = 0; d < use; d++) all.push_back(d);
        devices = all.data();
        n = use;
    }
    check(srt_pt_create_multi(devices, n, &group), "srt_pt_create_multi");
    for(int r = 0; r < srt_pt_group_size(group); r++) {
        members.push_back(srt_pt_group_context(group, r));
        // the BSDF-sampled direct ray whose term the reference adds and subtracts again is not traced where that is provably
        // result-neutral (srt_pt_set_elision): bit-identical image, Cornell-type scenes ~25 % faster
        check(srt_pt_set_elision(members.back(), 1), "srt_pt_set_elision");
    }
    check(srt_pt_create(devices[0], &display_ctx), "srt_pt_create (display)");
    n_threads = std::thread::hardware_concurrency();
}

RenderCore::~RenderCore() {
    cancel();
    srt_pt_destroy(display_ctx);
    srt_pt_group_destroy(group);
}

void RenderCore::set_params(size_t w, size_t h, size_t samples, size_t depth) {
    cancel();
    out_w = w; out_h = h; n_samples = samples; max_depth = depth;
    {
        std::lock_guard<std::mutex> lock(accumulator_mut);
        accumulator.assign(3 * w * h, 0.0f);
    }
    epoch_buf.assign(3 * w * h, 0.0f);
    check(srt_pt_group_set_params(group, (uint32_t)w, (uint32_t)h, (uint32_t)depth), "srt_pt_group_set_params");
}

// rays/pathtracer.cpp:195-207: s += (n - s) * (1.0f / accumulator_samples), in epoch completion order (one worker: issue order)
void RenderCore::accumulate(const float* epoch) {
    std::lock_guard<std::mutex> lock(accumulator_mut);
    accumulator_samples++;
    const float inv = 1.0f / accumulator_samples;
    for(size_t i = 0; i < accumulator.size(); i++) accumulator[i] += (epoch[i] - accumulator[i]) * inv;
}

void RenderCore::worker(size_t samples_per_epoch, size_t first_sample) {
    for(size_t s = 0; s < n_samples; s += samples_per_epoch) {
        if(cancel_flag.load()) return;
        const size_t samples = (s + samples_per_epoch) > n_samples ? n_samples - s : samples_per_epoch;
        check(srt_pt_group_render_epoch(group, seed, (uint32_t)(first_sample + s), (uint32_t)samples, epoch_buf.data()),
              "srt_pt_group_render_epoch");
        accumulate(epoch_buf.data());
        const size_t completed = completed_epochs++;
        if(completed + 1 == total_epochs)
            render_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_render0).count();
    }
}

void RenderCore::begin(const float iview[16], float vert_fov_deg, float aspect_ratio, bool add_samples) {
    const size_t threads = n_threads ? n_threads : 1;
    const size_t samples_per_epoch = std::max(size_t(1), n_samples / (threads * 10));     // rays/pathtracer.cpp:252-253
    cancel();
    total_epochs = n_samples / samples_per_epoch + !!(n_samples % samples_per_epoch);
    if(!add_samples) {
        std::lock_guard<std::mutex> lock(accumulator_mut);
        std::fill(accumulator.begin(), accumulator.end(), 0.0f);
        accumulator_samples = 0;
        samples_done = 0;
    }
    t_render0 = std::chrono::steady_clock::now();
    for(srt_pt* m : members) check(srt_pt_set_camera(m, iview, vert_fov_deg, aspect_ratio), "srt_pt_set_camera");
    const size_t first = samples_done;
    samples_done += n_samples;
    render_thread = std::thread([this, samples_per_epoch, first]() { worker(samples_per_epoch, first); });
}

void RenderCore::wait() {
    if(render_thread.joinable()) render_thread.join();
}

void RenderCore::cancel() {
    cancel_flag = true;
    if(render_thread.joinable()) render_thread.join();
    completed_epochs = 0;
    total_epochs = 0;
    cancel_flag = false;
}

void RenderCore::copy_accumulator(std::vector<float>& out) {
    std::lock_guard<std::mutex> lock(accumulator_mut);
    out = accumulator;
}

void RenderCore::tonemap(std::vector<unsigned char>& data, float exposure) {
    if(exposure > 0.0f) display_exposure = exposure;
    copy_accumulator(tonemap_in);                      // the lock is held for the copy only
    if(data.size() != out_w * out_h * 4) data.resize(out_w * out_h * 4);
    if(out_w == 0 || out_h == 0) return;
    check(srt_pt_tonemap(display_ctx, tonemap_in.data(), (uint32_t)out_w, (uint32_t)out_h, display_exposure, data.data()), "srt_pt_tonemap");
}

}  // namespace srt_host

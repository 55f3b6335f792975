// See pathtracer_core.h.
#include "pathtracer_core.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>

#include <hip/hip_runtime_api.h>

namespace srt_host {

void RenderCore::check(int status, const char* what) const {
    if(status != SRT_OK) fatal(what, status, srt_last_error());
}

RenderCore::RenderCore(const int* devices, int n, void (*fatal_)(const char*, int, const char*)) : fatal(fatal_) {
    std::vector<int> all;
    if(!devices) {
        // Default: ONE device (device 0).  More devices are opt-in - the constructor's list or SRT_PT_DEVICES=<count> - because that
        // routes the image through srt_pt_group's distinct-device gather (ncclCommInitAll + grouped ncclGather), which no
        // hardware run has shown bit-equal to the single context yet (Multi-GPU).
        int count = 0;
        if(hipGetDeviceCount(&count) != hipSuccess || count <= 0) fatal("hipGetDeviceCount", SRT_ERR_NO_DEVICE, "no HIP device: this path has no CPU fallback");
        int use = 1;
        if(const char* lim = getenv("SRT_PT_DEVICES")) use = std::max(1, std::min(count, atoi(lim)));
        for(int d = 0; d < use; d++) all.push_back(d);
        devices = all.data();
        n = use;
    }
    device0 = devices[0];
    check(srt_pt_create_multi(devices, n, &group), "srt_pt_create_multi");
    for(int r = 0; r < srt_pt_group_size(group); r++) {
        members.push_back(srt_pt_group_context(group, r));
        // the BSDF-sampled direct ray whose term the reference adds and subtracts again is not traced where that is provably
        // result-neutral (srt_pt_set_elision): bit-identical image, Cornell-type scenes ~25 % faster
        check(srt_pt_set_elision(members.back(), 1), "srt_pt_set_elision");
    }
    n_threads = std::thread::hardware_concurrency();
}

RenderCore::~RenderCore() {
    cancel();
    if(d_rgba) { (void)hipSetDevice(device0); (void)hipFree(d_rgba); }
    srt_pt_group_destroy(group);
}

void RenderCore::set_ray_log(void (*sink)(void*, const srt_pt_logged_ray*, size_t), void* user, uint32_t capacity) {
    cancel();
    ray_sink = sink; ray_sink_user = user;
    check(srt_pt_group_set_ray_log(group, sink ? capacity : 0u), "srt_pt_group_set_ray_log");
}

void RenderCore::set_params(size_t w, size_t h, size_t samples, size_t depth) {
    cancel();
    out_w = w; out_h = h; n_samples = samples; max_depth = depth;
    std::lock_guard<std::mutex> lock(accumulator_mut);
    check(srt_pt_group_set_params(group, (uint32_t)w, (uint32_t)h, (uint32_t)depth), "srt_pt_group_set_params");
    check(srt_pt_group_reset_accumulator(group), "srt_pt_group_reset_accumulator");      // accumulator.resize(out_w, out_h): zeros
    accumulator_samples = 0;
}

// The stand-in for the thread pool's do_trace tasks.  Launches of up to `most` samples per pixel, two in flight; a launch that
// has completed un-cancelled is folded - do_trace's epoch means, accumulate's running mean (rays/pathtracer.cpp:195-231), epoch
// by epoch - into the accumulator on the devices.  A cancelled launch is dropped like the reference's partial epochs.
void RenderCore::worker(size_t samples_per_epoch, size_t first_sample, size_t first_epochs) {
    struct Launch { size_t pos, n; int lane; };
    uint32_t most = 1;
    check(srt_pt_group_max_samples_per_launch(group, &most), "srt_pt_group_max_samples_per_launch");
    std::deque<Launch> inflight;
    size_t next = 0;
    int lane = 0;
    for(;;) {
        while(inflight.size() < 2 && next < n_samples && !cancel_flag.load()) {
            const size_t n = std::min<size_t>(most, n_samples - next);
            const int st = srt_pt_group_render_samples(group, lane, seed, (uint32_t)(first_sample + next), (uint32_t)n);
            if(st == SRT_CANCELLED) return;
            check(st, "srt_pt_group_render_samples");
            inflight.push_back({next, n, lane});
            next += n;
            lane ^= 1;
        }
        if(inflight.empty()) return;
        const Launch l = inflight.front();
        inflight.pop_front();
        check(srt_pt_group_wait_lane(group, l.lane), "srt_pt_group_wait_lane");
        if(cancel_flag.load() || srt_pt_group_cancel_requested(group)) return;               // the launch may have been cut short: dropped
        const size_t through = l.pos + l.n;                                                    // samples of the render folded after this launch
        const size_t epochs_through = through == n_samples ? total_epochs : through / samples_per_epoch;
        {
            std::lock_guard<std::mutex> lock(accumulator_mut);
            check(srt_pt_group_fold(group, l.lane, (uint32_t)samples_per_epoch, (uint32_t)l.pos, (uint32_t)n_samples, (uint32_t)first_epochs),
                  "srt_pt_group_fold");
            accumulator_samples = first_epochs + epochs_through;
        }
        if(ray_sink) {                                                                         // Pathtracer::log_ray -> gui.log_ray
            size_t waiting = 0, got = 0;
            uint64_t dropped = 0;
            check(srt_pt_group_read_ray_log(group, l.lane, nullptr, 0, &waiting, nullptr), "srt_pt_group_read_ray_log");
            if(waiting) {
                ray_buf.resize(waiting);
                check(srt_pt_group_read_ray_log(group, l.lane, ray_buf.data(), waiting, &got, &dropped), "srt_pt_group_read_ray_log");
                if(got) ray_sink(ray_sink_user, ray_buf.data(), got);
            }
        }
        completed_epochs = epochs_through;
        if(epochs_through == total_epochs)
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
        check(srt_pt_group_reset_accumulator(group), "srt_pt_group_reset_accumulator");
        accumulator_samples = 0;
        samples_done = 0;
    }
    t_render0 = std::chrono::steady_clock::now();
    for(srt_pt* m : members) check(srt_pt_set_camera(m, iview, vert_fov_deg, aspect_ratio), "srt_pt_set_camera");
    const size_t first = samples_done, first_epochs = accumulator_samples.load();
    samples_done += n_samples;
    render_thread = std::thread([this, samples_per_epoch, first, first_epochs]() { worker(samples_per_epoch, first, first_epochs); });
}

void RenderCore::wait() {
    if(render_thread.joinable()) render_thread.join();
}

// Pathtracer::cancel (rays/pathtracer.cpp:282-290): the flag reaches the kernels (srt_pt_cancel), the launches in flight end
// within milliseconds and are dropped, what was folded stays.
void RenderCore::cancel() {
    cancel_flag = true;
    if(render_thread.joinable()) {
        srt_pt_group_cancel(group);
        render_thread.join();
        check(srt_pt_group_clear_cancel(group), "srt_pt_group_clear_cancel");
    }
    completed_epochs = 0;
    total_epochs = 0;
    cancel_flag = false;
}

void RenderCore::copy_accumulator(std::vector<float>& out) {
    out.resize(3 * out_w * out_h);
    if(out.empty()) return;
    std::lock_guard<std::mutex> lock(accumulator_mut);   // (held until the copy is down: a fold behind it waits on the device anyway)
    float* d_image = nullptr; void* s = nullptr;
    check(srt_pt_group_accumulator_image(group, &d_image, &s), "srt_pt_group_accumulator_image");
    if(hipSetDevice(device0) != hipSuccess ||
       hipMemcpyAsync(out.data(), d_image, out.size() * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)s) != hipSuccess ||
       hipStreamSynchronize((hipStream_t)s) != hipSuccess)
        fatal("copy_accumulator", SRT_ERR_HIP, hipGetErrorString(hipGetLastError()));
}

void RenderCore::tonemap(std::vector<unsigned char>& data, float exposure) {
    if(exposure > 0.0f) display_exposure = exposure;
    if(data.size() != out_w * out_h * 4) data.resize(out_w * out_h * 4);
    if(out_w == 0 || out_h == 0) return;
    std::lock_guard<std::mutex> lock(accumulator_mut);
    if(hipSetDevice(device0) != hipSuccess) fatal("tonemap", SRT_ERR_HIP, "hipSetDevice");
    if(rgba_bytes != data.size()) {
        if(d_rgba) (void)hipFree(d_rgba);
        d_rgba = nullptr; rgba_bytes = 0;
        if(hipMalloc((void**)&d_rgba, data.size()) != hipSuccess) fatal("tonemap", SRT_ERR_HIP, "hipMalloc");
        rgba_bytes = data.size();
    }
    float* d_image = nullptr; void* s = nullptr;
    check(srt_pt_group_accumulator_image(group, &d_image, &s), "srt_pt_group_accumulator_image");
    // HDR_Image::tonemap_to as a device epilogue behind the gather: 4 bytes per pixel come down instead of 12
    check(srt_pt_tonemap_device(members[0], s, d_image, (uint32_t)out_w, (uint32_t)out_h, display_exposure, d_rgba), "srt_pt_tonemap_device");
    if(hipMemcpyAsync(data.data(), d_rgba, data.size(), hipMemcpyDeviceToHost, (hipStream_t)s) != hipSuccess ||
       hipStreamSynchronize((hipStream_t)s) != hipSuccess)
        fatal("tonemap", SRT_ERR_HIP, hipGetErrorString(hipGetLastError()));
}

}  // namespace srt_host

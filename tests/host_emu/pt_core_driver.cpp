// TEST-ONLY: a C surface over srt_host::RenderCore (soft-rendering-toolsets_amd/host/pathtracer_core.{h,cpp}) - the
// Scene-independent part of the drop-in PT::Pathtracer - so that tests/test_dropin_gpu.py can run its epoch scheme, worker,
// running mean, cancel, "Add Samples" and display epilogue against the C ABI for real.  Built by tests/_harness.py with g++
// and linked to the product library; the scene is fed through the members' contexts with the ordinary C ABI calls.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pathtracer_core.h"

namespace {
void fatal(const char* what, int status, const char* message) {
  std::fprintf(stderr, "[pt_core_driver] %s failed (%d): %s\n", what, status, message);
  std::abort();
}
}  // namespace

extern "C" {

void* core_create(const int* devices, int n) { return new srt_host::RenderCore(devices, n, fatal); }
void core_destroy(void* h) { delete (srt_host::RenderCore*)h; }
int core_ranks(void* h) { return ((srt_host::RenderCore*)h)->ranks(); }
void* core_context(void* h, int rank) { return ((srt_host::RenderCore*)h)->context(rank); }
void core_set_params(void* h, size_t w, size_t hh, size_t samples, size_t depth) { ((srt_host::RenderCore*)h)->set_params(w, hh, samples, depth); }
void core_set_samples(void* h, size_t samples) { ((srt_host::RenderCore*)h)->set_samples(samples); }
void core_set_seed(void* h, unsigned long long seed) { ((srt_host::RenderCore*)h)->set_seed(seed); }
void core_set_threads(void* h, size_t n) { ((srt_host::RenderCore*)h)->set_threads(n); }
void core_begin(void* h, const float* iview, float fov, float ar, int add_samples) { ((srt_host::RenderCore*)h)->begin(iview, fov, ar, add_samples != 0); }
void core_wait(void* h) { ((srt_host::RenderCore*)h)->wait(); }
void core_cancel(void* h) { ((srt_host::RenderCore*)h)->cancel(); }
int core_in_progress(void* h) { return ((srt_host::RenderCore*)h)->in_progress() ? 1 : 0; }
float core_progress(void* h) { return ((srt_host::RenderCore*)h)->progress(); }
size_t core_epochs_accumulated(void* h) { return ((srt_host::RenderCore*)h)->epochs_accumulated(); }
void core_copy_accumulator(void* h, float* out) {
  std::vector<float> v;
  ((srt_host::RenderCore*)h)->copy_accumulator(v);
  std::memcpy(out, v.data(), v.size() * sizeof(float));
}
void core_tonemap(void* h, unsigned char* out, float exposure) {
  std::vector<unsigned char> v;
  ((srt_host::RenderCore*)h)->tonemap(v, exposure);
  std::memcpy(out, v.data(), v.size());
}

// Pathtracer::log_ray's sink: the driver keeps what the core delivers (40 bytes per ray, srt_pt_logged_ray) for the test to fetch
static std::vector<srt_pt_logged_ray> g_logged;
static void keep_rays(void*, const srt_pt_logged_ray* rays, size_t n) { g_logged.insert(g_logged.end(), rays, rays + n); }
void core_enable_ray_log(void* h, unsigned capacity) { g_logged.clear(); ((srt_host::RenderCore*)h)->set_ray_log(keep_rays, nullptr, capacity); }
size_t core_logged_rays(void* out, size_t cap) {
  const size_t n = g_logged.size() < cap ? g_logged.size() : cap;
  if (out && n) std::memcpy(out, g_logged.data(), n * sizeof(srt_pt_logged_ray));
  return g_logged.size();
}

}  // extern "C"

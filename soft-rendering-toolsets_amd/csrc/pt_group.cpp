// Image-tile sharding of one render over the GPUs of a node inside ONE process: the multi-device form of the path
// tracer's C ABI (include/srt_pt.h, srt_pt_create_multi).
//
// The reference's only parallel axis is the epoch fan-out of Pathtracer::begin_render over a thread pool
// (rays/pathtracer.cpp:250-280); here an epoch is cut into 32 x 32 image tiles dealt round-robin to the devices
// (srt_pt_set_tiling), every device renders its tiles with its own context and stream - the scene is replicated, a few MB -
// and the tile radiance meets on device 0 through ONE RCCL gather per epoch (ncclGather over xGMI, grouped over the
// process's communicators), followed by the un-tiling kernel.  No reduction: the tiles are disjoint.
// RCCL is loaded on first use (dlopen), so single-GPU users never touch it.  Logical ranks that share a device - the way
// the N > 1 path is exercised on a one-GPU box - cannot form an RCCL communicator; their tiles are gathered with
// device-to-device copies instead (also the fallback when librccl is absent; SRT_PT_GATHER=copy|rccl forces one).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <set>
#include <vector>

#include <hip/hip_runtime.h>

#include "srt_common.h"
#include "srt_pt.h"

namespace {

// the part of <rccl/rccl.h> that is used, bound at run time
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclFloat = 7;   // ncclFloat32
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Gather)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    Gather = (decltype(Gather))dlsym(lib, "ncclGather");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Gather && GetErrorString;
  }
};

}  // namespace

struct srt_pt_group {
  std::vector<int> devices;
  std::vector<srt_pt*> ctx;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> events;
  std::vector<float*> d_tiles;          // per rank, on its device: tiles_per_rank * floats_per_tile
  size_t tile_floats = 0;               // per rank
  float* d_gather = nullptr;            // device 0: n * tile_floats
  float* d_image = nullptr;             // device 0: w * h * 3
  size_t image_floats = 0;
  bool use_rccl = false;
  Rccl rccl;
  std::vector<ncclComm_t> comms;
  uint32_t tile_w = 32, tile_h = 32;
  // srt_pt_group_gather_time: event pairs on rank 0's stream around the gather + un-tiling of each epoch
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed, spare;
};

namespace {

int free_buffers(srt_pt_group* g) {
  for (size_t r = 0; r < g->d_tiles.size(); r++)
    if (g->d_tiles[r]) { (void)hipSetDevice(g->devices[r]); (void)hipFree(g->d_tiles[r]); g->d_tiles[r] = nullptr; }
  (void)hipSetDevice(g->devices[0]);
  if (g->d_gather) { (void)hipFree(g->d_gather); g->d_gather = nullptr; }
  if (g->d_image) { (void)hipFree(g->d_image); g->d_image = nullptr; }
  g->tile_floats = g->image_floats = 0;
  return SRT_OK;
}

// (Re)sizes the exchange buffers after srt_pt_set_params on the members.
int ensure_buffers(srt_pt_group* g, uint32_t w, uint32_t h) {
  uint32_t local = 0, per_rank = 0, fpt = 0;
  int st = srt_pt_tile_info(g->ctx[0], &local, &per_rank, &fpt);
  if (st != SRT_OK) return st;
  const size_t tf = (size_t)per_rank * fpt, imf = (size_t)w * h * 3;
  if (tf == g->tile_floats && imf == g->image_floats) return SRT_OK;
  free_buffers(g);
  const size_t n = g->ctx.size();
  for (size_t r = 0; r < n; r++) {
    SRT_HIP(hipSetDevice(g->devices[r]));
    SRT_HIP(hipMalloc(&g->d_tiles[r], (tf ? tf : 1) * sizeof(float)));
  }
  SRT_HIP(hipSetDevice(g->devices[0]));
  SRT_HIP(hipMalloc(&g->d_gather, (tf ? tf : 1) * n * sizeof(float)));
  SRT_HIP(hipMalloc(&g->d_image, (imf ? imf : 1) * sizeof(float)));
  g->tile_floats = tf; g->image_floats = imf;
  return SRT_OK;
}

}  // namespace

extern "C" {

int srt_pt_create_multi(const int* devices, int n, srt_pt_group** out) {
  if (!out) return srt::fail(SRT_ERR_INVALID, "srt_pt_create_multi: out is NULL");
  *out = nullptr;
  if (!devices || n < 1 || n > 64) return srt::fail(SRT_ERR_INVALID, "srt_pt_create_multi: need 1..64 devices (got %d)", n);
  srt_pt_group* g = new (std::nothrow) srt_pt_group();
  if (!g) return srt::fail(SRT_ERR_INVALID, "out of host memory");
  g->devices.assign(devices, devices + n);
  g->ctx.assign(n, nullptr); g->streams.assign(n, nullptr); g->events.assign(n, nullptr); g->d_tiles.assign(n, nullptr);
  int st = SRT_OK;
  for (int r = 0; r < n && st == SRT_OK; r++) {
    st = srt_pt_create(devices[r], &g->ctx[r]);
    if (st == SRT_OK) st = srt_pt_set_tiling(g->ctx[r], g->tile_w, g->tile_h, (uint32_t)r, (uint32_t)n);
    if (st == SRT_OK && (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreateWithFlags(&g->streams[r], hipStreamNonBlocking) != hipSuccess ||
                         hipEventCreateWithFlags(&g->events[r], hipEventDisableTiming) != hipSuccess))
      st = srt::fail(SRT_ERR_HIP, "srt_pt_create_multi: stream setup failed on device %d", devices[r]);
  }
  if (st == SRT_OK && n > 0) {
    // one RCCL communicator per rank when every rank has a device of its own; shared devices gather by copies
    const std::set<int> distinct(g->devices.begin(), g->devices.end());
    const char* mode = getenv("SRT_PT_GATHER");
    const bool want_rccl = mode ? std::strcmp(mode, "rccl") == 0 : (n > 1 && (int)distinct.size() == n);
    if (want_rccl && (int)distinct.size() == n) {
      if (g->rccl.load()) {
        g->comms.assign(n, nullptr);
        const ncclResult_t rc = g->rccl.CommInitAll(g->comms.data(), n, g->devices.data());
        if (rc != 0) st = srt::fail(SRT_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n, g->rccl.GetErrorString(rc));
        else g->use_rccl = true;
      } else if (mode) {
        st = srt::fail(SRT_ERR_UNSUPPORTED, "SRT_PT_GATHER=rccl but librccl could not be loaded: %s", dlerror());
      }
    }
  }
  if (st != SRT_OK) { srt_pt_group_destroy(g); return st; }
  *out = g;
  return SRT_OK;
}

int srt_pt_group_destroy(srt_pt_group* g) {
  if (!g) return SRT_OK;
  for (size_t r = 0; r < g->ctx.size(); r++)
    if (g->streams[r]) { (void)hipSetDevice(g->devices[r]); (void)hipStreamSynchronize(g->streams[r]); }
  if (g->use_rccl)
    for (ncclComm_t c : g->comms)
      if (c) (void)g->rccl.CommDestroy(c);
  free_buffers(g);
  for (auto* v : {&g->timed, &g->spare})
    for (auto& ev : *v) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (size_t r = 0; r < g->ctx.size(); r++) {
    (void)hipSetDevice(g->devices[r]);
    if (g->events[r]) (void)hipEventDestroy(g->events[r]);
    if (g->streams[r]) (void)hipStreamDestroy(g->streams[r]);
    if (g->ctx[r]) (void)srt_pt_destroy(g->ctx[r]);
  }
  delete g;
  return SRT_OK;
}

int srt_pt_group_size(srt_pt_group* g) { return g ? (int)g->ctx.size() : 0; }

srt_pt* srt_pt_group_context(srt_pt_group* g, int rank) {
  if (!g || rank < 0 || rank >= (int)g->ctx.size()) { srt::fail(SRT_ERR_INVALID, "srt_pt_group_context: rank %d out of range", rank); return nullptr; }
  return g->ctx[rank];
}

int srt_pt_group_uses_rccl(srt_pt_group* g) { return g && g->use_rccl ? 1 : 0; }

int srt_pt_group_set_params(srt_pt_group* g, uint32_t width, uint32_t height, uint32_t max_depth) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_set_params: NULL group");
  for (srt_pt* c : g->ctx) {
    const int st = srt_pt_set_params(c, width, height, max_depth);
    if (st != SRT_OK) return st;
  }
  return ensure_buffers(g, width, height);
}

int srt_pt_group_render_epoch_device(srt_pt_group* g, uint64_t seed, uint32_t sample_base, uint32_t samples, float** d_image_out, void** stream_out) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_render_epoch: NULL group");
  if (!g->image_floats) return srt::fail(SRT_ERR_STATE, "srt_pt_group_render_epoch before srt_pt_group_set_params");
  const size_t n = g->ctx.size();
  int st;
  for (size_t r = 0; r < n; r++) {       // every rank renders its tiles; the launches only enqueue
    SRT_HIP(hipSetDevice(g->devices[r]));
    if ((st = srt_pt_render_epoch_device(g->ctx[r], (void*)g->streams[r], seed, sample_base, samples, g->d_tiles[r])) != SRT_OK) return st;
  }
  const float* gathered = g->d_tiles[0];
  if (g->timing) {                       // (the bracket opens once rank 0's own tiles are rendered: what follows is exchange + un-tiling)
    std::pair<hipEvent_t, hipEvent_t> ev;
    SRT_HIP(hipSetDevice(g->devices[0]));
    if (!g->spare.empty()) { ev = g->spare.back(); g->spare.pop_back(); }
    else { SRT_HIP(hipEventCreate(&ev.first)); SRT_HIP(hipEventCreate(&ev.second)); }
    g->timed.push_back(ev);
    SRT_HIP(hipEventRecord(ev.first, g->streams[0]));
  }
  if (g->use_rccl) {                     // ONE collective per epoch: tile radiance -> rank 0
    // (no early return between GroupStart and GroupEnd: a failing rank ends the loop, the group is always closed)
    ncclResult_t rc = g->rccl.GroupStart();
    hipError_t he = hipSuccess;
    for (size_t r = 0; r < n && rc == 0 && he == hipSuccess; r++) {
      he = hipSetDevice(g->devices[r]);
      if (he == hipSuccess) rc = g->rccl.Gather(g->d_tiles[r], r == 0 ? g->d_gather : nullptr, g->tile_floats, kNcclFloat, 0, g->comms[r], g->streams[r]);
    }
    const ncclResult_t rc2 = g->rccl.GroupEnd();
    if (he != hipSuccess) return srt::fail(SRT_ERR_HIP, "hipSetDevice inside the gather group failed: %s", hipGetErrorString(he));
    if (rc != 0 || rc2 != 0) return srt::fail(SRT_ERR_HIP, "ncclGather failed: %s", g->rccl.GetErrorString(rc != 0 ? rc : rc2));
    gathered = g->d_gather;
  } else if (n > 1) {                    // ranks that share a device (or no RCCL): device-to-device copies behind events
    for (size_t r = 0; r < n; r++) {
      SRT_HIP(hipSetDevice(g->devices[r]));
      SRT_HIP(hipEventRecord(g->events[r], g->streams[r]));
    }
    SRT_HIP(hipSetDevice(g->devices[0]));
    for (size_t r = 0; r < n; r++) {
      SRT_HIP(hipStreamWaitEvent(g->streams[0], g->events[r], 0));
      if (g->devices[r] == g->devices[0])
        SRT_HIP(hipMemcpyAsync(g->d_gather + r * g->tile_floats, g->d_tiles[r], g->tile_floats * sizeof(float), hipMemcpyDeviceToDevice, g->streams[0]));
      else
        SRT_HIP(hipMemcpyPeerAsync(g->d_gather + r * g->tile_floats, g->devices[0], g->d_tiles[r], g->devices[r], g->tile_floats * sizeof(float), g->streams[0]));
    }
    gathered = g->d_gather;
  }
  SRT_HIP(hipSetDevice(g->devices[0]));
  if ((st = srt_pt_untile_device(g->ctx[0], (void*)g->streams[0], gathered, g->d_image)) != SRT_OK) return st;
  if (g->timing) SRT_HIP(hipEventRecord(g->timed.back().second, g->streams[0]));
  if (n > 1 && !g->use_rccl) {
    // the next epoch must not overwrite a rank's tiles before rank 0 has copied them
    SRT_HIP(hipEventRecord(g->events[0], g->streams[0]));
    for (size_t r = 1; r < n; r++) { SRT_HIP(hipSetDevice(g->devices[r])); SRT_HIP(hipStreamWaitEvent(g->streams[r], g->events[0], 0)); }
    SRT_HIP(hipSetDevice(g->devices[0]));
  }
  if (d_image_out) *d_image_out = g->d_image;
  if (stream_out) *stream_out = (void*)g->streams[0];
  return SRT_OK;
}

int srt_pt_group_gather_time(srt_pt_group* g, int enable, double* total_ms, uint64_t* epochs) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_gather_time: NULL group");
  SRT_HIP(hipSetDevice(g->devices[0]));
  double sum = 0.0;
  uint64_t n = 0;
  for (auto& ev : g->timed) {
    SRT_HIP(hipEventSynchronize(ev.second));
    float ms = 0.f;
    SRT_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
    sum += ms; n++;
    g->spare.push_back(ev);
  }
  g->timed.clear();
  g->timing = enable != 0;
  if (total_ms) *total_ms = sum;
  if (epochs) *epochs = n;
  return SRT_OK;
}

int srt_pt_group_render_epoch(srt_pt_group* g, uint64_t seed, uint32_t sample_base, uint32_t samples, float* rgb_out) {
  if (!rgb_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_render_epoch: output is NULL");
  float* d_image = nullptr; void* s = nullptr;
  const int st = srt_pt_group_render_epoch_device(g, seed, sample_base, samples, &d_image, &s);
  if (st != SRT_OK) return st;
  SRT_HIP(hipMemcpyAsync(rgb_out, d_image, g->image_floats * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)s));
  SRT_HIP(hipStreamSynchronize((hipStream_t)s));
  return SRT_OK;
}

}  // extern "C"

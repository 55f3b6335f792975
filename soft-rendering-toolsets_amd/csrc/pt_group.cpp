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
#include <algorithm>
#include <set>
#include <vector>

#include <hip/hip_runtime.h>

#include "srt_common.h"
#include "srt_pt.h"
#include "pt_internal.h"

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

// One set of what an epoch in flight needs: a stream and an event per rank, the ranks' tile buffers, the gathered tiles and the
// image on rank 0.  Lane 0 always exists; further lanes (srt_pt_group_render_epoch_lane) let epochs overlap.
struct GroupLane {
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> events;
  std::vector<float*> d_tiles;          // per rank, on its device: tiles_per_rank * floats_per_tile
  float* d_gather = nullptr;            // device 0: n * tile_floats
  float* d_image = nullptr;             // device 0: w * h * 3
  size_t tile_floats = 0, image_floats = 0;   // what the buffers were sized for
};
constexpr int kMaxLanes = 4;

struct srt_pt_group {
  std::vector<int> devices;
  std::vector<srt_pt*> ctx;
  std::vector<GroupLane> lanes;
  size_t tile_floats = 0;               // per rank (srt_pt_group_set_params)
  size_t image_floats = 0;
  bool use_rccl = false;
  Rccl rccl;
  std::vector<ncclComm_t> comms;
  uint32_t tile_w = 32, tile_h = 32;
  // the accumulator of a render kept on the devices (srt_pt_group_fold): per rank its tiles' state, the event behind its last
  // fold and the one behind the last read of it (srt_pt_group_accumulator_image, on the display lane)
  std::vector<float*> d_acc; std::vector<size_t> acc_floats;
  std::vector<hipEvent_t> fold_done, image_done;
  std::vector<char> fold_recorded, image_recorded;
  // srt_pt_group_gather_time: event pairs on rank 0's stream around the gather + un-tiling of each epoch
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed, spare;
};

namespace {

void free_lane_buffers(srt_pt_group* g, GroupLane& L) {
  for (size_t r = 0; r < L.d_tiles.size(); r++)
    if (L.d_tiles[r]) { (void)hipSetDevice(g->devices[r]); (void)hipFree(L.d_tiles[r]); L.d_tiles[r] = nullptr; }
  (void)hipSetDevice(g->devices[0]);
  if (L.d_gather) { (void)hipFree(L.d_gather); L.d_gather = nullptr; }
  if (L.d_image) { (void)hipFree(L.d_image); L.d_image = nullptr; }
  L.tile_floats = L.image_floats = 0;
}

// A lane's streams and events (once) and its buffers (whenever srt_pt_group_set_params changed the sizes).
int ensure_lane(srt_pt_group* g, int lane) {
  if (lane < 0 || lane >= kMaxLanes) return srt::fail(SRT_ERR_INVALID, "srt_pt_group: lane %d out of range [0, %d)", lane, kMaxLanes);
  const size_t n = g->ctx.size();
  while ((int)g->lanes.size() <= lane) {
    GroupLane L;
    L.streams.assign(n, nullptr); L.events.assign(n, nullptr); L.d_tiles.assign(n, nullptr);
    g->lanes.push_back(L);
    GroupLane& N = g->lanes.back();
    for (size_t r = 0; r < n; r++) {
      SRT_HIP(hipSetDevice(g->devices[r]));
      SRT_HIP(hipStreamCreateWithFlags(&N.streams[r], hipStreamNonBlocking));
      SRT_HIP(hipEventCreateWithFlags(&N.events[r], hipEventDisableTiming));
    }
  }
  GroupLane& L = g->lanes[lane];
  if (L.tile_floats == g->tile_floats && L.image_floats == g->image_floats && L.d_image) return SRT_OK;
  for (size_t r = 0; r < n; r++)
    if (L.streams[r]) { SRT_HIP(hipSetDevice(g->devices[r])); SRT_HIP(hipStreamSynchronize(L.streams[r])); }
  free_lane_buffers(g, L);
  const size_t tf = g->tile_floats, imf = g->image_floats;
  for (size_t r = 0; r < n; r++) {
    SRT_HIP(hipSetDevice(g->devices[r]));
    SRT_HIP(hipMalloc(&L.d_tiles[r], (tf ? tf : 1) * sizeof(float)));
  }
  SRT_HIP(hipSetDevice(g->devices[0]));
  SRT_HIP(hipMalloc(&L.d_gather, (tf ? tf : 1) * n * sizeof(float)));
  SRT_HIP(hipMalloc(&L.d_image, (imf ? imf : 1) * sizeof(float)));
  L.tile_floats = tf; L.image_floats = imf;
  return SRT_OK;
}

// The exchange step of one epoch on one lane (everything behind the members' launches).
int exchange(srt_pt_group* g, GroupLane& L) {
  const size_t n = g->ctx.size();
  const float* gathered = L.d_tiles[0];
  if (g->use_rccl) {                     // ONE collective per epoch: tile radiance -> rank 0
    // (no early return between GroupStart and GroupEnd: a failing rank ends the loop, the group is always closed)
    ncclResult_t rc = g->rccl.GroupStart();
    hipError_t he = hipSuccess;
    for (size_t r = 0; r < n && rc == 0 && he == hipSuccess; r++) {
      he = hipSetDevice(g->devices[r]);
      if (he == hipSuccess) rc = g->rccl.Gather(L.d_tiles[r], r == 0 ? L.d_gather : nullptr, g->tile_floats, kNcclFloat, 0, g->comms[r], L.streams[r]);
    }
    const ncclResult_t rc2 = g->rccl.GroupEnd();
    if (he != hipSuccess) return srt::fail(SRT_ERR_HIP, "hipSetDevice inside the gather group failed: %s", hipGetErrorString(he));
    if (rc != 0 || rc2 != 0) return srt::fail(SRT_ERR_HIP, "ncclGather failed: %s", g->rccl.GetErrorString(rc != 0 ? rc : rc2));
    gathered = L.d_gather;
  } else if (n > 1) {                    // ranks that share a device (or no RCCL): device-to-device copies behind events
    for (size_t r = 0; r < n; r++) {
      SRT_HIP(hipSetDevice(g->devices[r]));
      SRT_HIP(hipEventRecord(L.events[r], L.streams[r]));
    }
    SRT_HIP(hipSetDevice(g->devices[0]));
    for (size_t r = 0; r < n; r++) {
      SRT_HIP(hipStreamWaitEvent(L.streams[0], L.events[r], 0));
      if (g->devices[r] == g->devices[0])
        SRT_HIP(hipMemcpyAsync(L.d_gather + r * g->tile_floats, L.d_tiles[r], g->tile_floats * sizeof(float), hipMemcpyDeviceToDevice, L.streams[0]));
      else
        SRT_HIP(hipMemcpyPeerAsync(L.d_gather + r * g->tile_floats, g->devices[0], L.d_tiles[r], g->devices[r], g->tile_floats * sizeof(float), L.streams[0]));
    }
    gathered = L.d_gather;
  }
  SRT_HIP(hipSetDevice(g->devices[0]));
  return srt_pt_untile_device(g->ctx[0], (void*)L.streams[0], gathered, L.d_image);
}

constexpr int kDisplayLane = kMaxLanes - 1;   // srt_pt_group_accumulator_image's own streams and exchange buffers

// Every rank's accumulator state, sized for the current image / tiling; (re)allocated and zeroed when the size changed.
int ensure_accumulators(srt_pt_group* g) {
  const size_t n = g->ctx.size();
  if (g->d_acc.size() != n) {
    g->d_acc.assign(n, nullptr); g->acc_floats.assign(n, 0); g->fold_done.assign(n, nullptr); g->image_done.assign(n, nullptr);
    g->fold_recorded.assign(n, 0); g->image_recorded.assign(n, 0);
    for (size_t r = 0; r < n; r++) {
      SRT_HIP(hipSetDevice(g->devices[r]));
      SRT_HIP(hipEventCreateWithFlags(&g->fold_done[r], hipEventDisableTiming));
      SRT_HIP(hipEventCreateWithFlags(&g->image_done[r], hipEventDisableTiming));
    }
  }
  for (size_t r = 0; r < n; r++) {
    size_t need = 0;
    const int st = srt_pt_accumulator_floats(g->ctx[r], &need);
    if (st != SRT_OK) return st;
    if (g->d_acc[r] && g->acc_floats[r] == need) continue;
    SRT_HIP(hipSetDevice(g->devices[r]));
    SRT_HIP(hipDeviceSynchronize());
    if (g->d_acc[r]) { SRT_HIP(hipFree(g->d_acc[r])); g->d_acc[r] = nullptr; }
    SRT_HIP(hipMalloc(&g->d_acc[r], need * sizeof(float)));
    SRT_HIP(hipMemset(g->d_acc[r], 0, need * sizeof(float)));
    SRT_HIP(hipDeviceSynchronize());                     // (null-stream memset: done before a lane's non-blocking stream folds into it)
    g->acc_floats[r] = need;
    g->fold_recorded[r] = g->image_recorded[r] = 0;
  }
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
  g->ctx.assign(n, nullptr);
  int st = SRT_OK;
  for (int r = 0; r < n && st == SRT_OK; r++) {
    st = srt_pt_create(devices[r], &g->ctx[r]);
    if (st == SRT_OK) st = srt_pt_set_tiling(g->ctx[r], g->tile_w, g->tile_h, (uint32_t)r, (uint32_t)n);
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
  for (GroupLane& L : g->lanes)
    for (size_t r = 0; r < L.streams.size(); r++)
      if (L.streams[r]) { (void)hipSetDevice(g->devices[r]); (void)hipStreamSynchronize(L.streams[r]); }
  if (g->use_rccl)
    for (ncclComm_t c : g->comms)
      if (c) (void)g->rccl.CommDestroy(c);
  for (auto* v : {&g->timed, &g->spare})
    for (auto& ev : *v) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (GroupLane& L : g->lanes) {
    free_lane_buffers(g, L);
    for (size_t r = 0; r < L.streams.size(); r++) {
      (void)hipSetDevice(g->devices[r]);
      if (L.events[r]) (void)hipEventDestroy(L.events[r]);
      if (L.streams[r]) (void)hipStreamDestroy(L.streams[r]);
    }
  }
  for (size_t r = 0; r < g->d_acc.size(); r++) {
    (void)hipSetDevice(g->devices[r]);
    if (g->d_acc[r]) (void)hipFree(g->d_acc[r]);
    if (g->fold_done[r]) (void)hipEventDestroy(g->fold_done[r]);
    if (g->image_done[r]) (void)hipEventDestroy(g->image_done[r]);
  }
  for (size_t r = 0; r < g->ctx.size(); r++) {
    (void)hipSetDevice(g->devices[r]);
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
  uint32_t local = 0, per_rank = 0, fpt = 0;
  const int st = srt_pt_tile_info(g->ctx[0], &local, &per_rank, &fpt);
  if (st != SRT_OK) return st;
  g->tile_floats = (size_t)per_rank * fpt;
  g->image_floats = (size_t)width * height * 3;
  return ensure_lane(g, 0);
}

int srt_pt_group_render_epoch_lane(srt_pt_group* g, int lane, uint64_t seed, uint32_t sample_base, uint32_t samples, float** d_image_out, void** stream_out) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_render_epoch: NULL group");
  if (!g->image_floats) return srt::fail(SRT_ERR_STATE, "srt_pt_group_render_epoch before srt_pt_group_set_params");
  int st = ensure_lane(g, lane);
  if (st != SRT_OK) return st;
  GroupLane& L = g->lanes[lane];
  const size_t n = g->ctx.size();
  for (size_t r = 0; r < n; r++) {       // every rank renders its tiles; the launches only enqueue
    SRT_HIP(hipSetDevice(g->devices[r]));
    if ((st = srt_pt_render_epoch_device(g->ctx[r], (void*)L.streams[r], seed, sample_base, samples, L.d_tiles[r])) != SRT_OK) return st;   // (SRT_CANCELLED included)
  }
  // (the bracket opens once rank 0's own tiles are rendered: what follows is exchange + un-tiling.  A pair is only kept once BOTH
  //  of its events are recorded: whatever fails in between puts it back)
  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  if (g->timing) {
    SRT_HIP(hipSetDevice(g->devices[0]));
    if (!g->spare.empty()) { ev = g->spare.back(); g->spare.pop_back(); }
    else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) {
      if (ev.first) (void)hipEventDestroy(ev.first);
      return srt::fail(SRT_ERR_HIP, "srt_pt_group: hipEventCreate failed");
    }
    if (hipEventRecord(ev.first, L.streams[0]) != hipSuccess) { g->spare.push_back(ev); return srt::fail(SRT_ERR_HIP, "srt_pt_group: hipEventRecord failed"); }
  }
  st = exchange(g, L);
  if (g->timing) {
    if (st == SRT_OK && hipEventRecord(ev.second, L.streams[0]) == hipSuccess) g->timed.push_back(ev);
    else g->spare.push_back(ev);
  }
  if (st != SRT_OK) return st;
  if (n > 1 && !g->use_rccl) {
    // the next epoch of this lane must not overwrite a rank's tiles before rank 0 has copied them
    SRT_HIP(hipEventRecord(L.events[0], L.streams[0]));
    for (size_t r = 1; r < n; r++) { SRT_HIP(hipSetDevice(g->devices[r])); SRT_HIP(hipStreamWaitEvent(L.streams[r], L.events[0], 0)); }
    SRT_HIP(hipSetDevice(g->devices[0]));
  }
  if (d_image_out) *d_image_out = L.d_image;
  if (stream_out) *stream_out = (void*)L.streams[0];
  return SRT_OK;
}

int srt_pt_group_render_epoch_device(srt_pt_group* g, uint64_t seed, uint32_t sample_base, uint32_t samples, float** d_image_out, void** stream_out) {
  return srt_pt_group_render_epoch_lane(g, 0, seed, sample_base, samples, d_image_out, stream_out);
}

int srt_pt_group_gather_time(srt_pt_group* g, int enable, double* total_ms, uint64_t* epochs) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_gather_time: NULL group");
  SRT_HIP(hipSetDevice(g->devices[0]));
  double sum = 0.0;
  uint64_t n = 0;
  int st = SRT_OK;
  for (auto& ev : g->timed) {            // (every pair goes back to `spare`, also when one of them cannot be read)
    float ms = 0.f;
    if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { sum += ms; n++; }
    else { (void)hipGetLastError(); st = srt::fail(SRT_ERR_HIP, "srt_pt_group_gather_time: an event pair could not be read"); }
    g->spare.push_back(ev);
  }
  g->timed.clear();
  g->timing = enable != 0;
  if (total_ms) *total_ms = sum;
  if (epochs) *epochs = n;
  return st;
}

int srt_pt_group_render_epoch(srt_pt_group* g, uint64_t seed, uint32_t sample_base, uint32_t samples, float* rgb_out) {
  if (!rgb_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_render_epoch: output is NULL");
  float* d_image = nullptr; void* s = nullptr;
  int st = srt_pt_group_render_epoch_device(g, seed, sample_base, samples, &d_image, &s);
  if (st != SRT_OK) return st;
  SRT_HIP(hipMemcpyAsync(rgb_out, d_image, g->image_floats * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)s));
  SRT_HIP(hipStreamSynchronize((hipStream_t)s));
  // the image is host-visible now: a member whose streamed launch ended with unfinished units, or a cancel, invalidates it
  for (srt_pt* c : g->ctx) {
    if (srt_pt_cancel_requested(c)) return SRT_CANCELLED;
    if ((st = srt::pt_check_fault(c, "srt_pt_group_render_epoch")) != SRT_OK) return st;
  }
  return SRT_OK;
}

// ---- a render with the accumulator on the devices ----------------------------------------------------------------------
int srt_pt_group_reset_accumulator(srt_pt_group* g) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_reset_accumulator: NULL group");
  if (!g->image_floats) return srt::fail(SRT_ERR_STATE, "srt_pt_group_reset_accumulator before srt_pt_group_set_params");
  int st = ensure_accumulators(g);
  if (st != SRT_OK) return st;
  for (size_t r = 0; r < g->ctx.size(); r++) {
    SRT_HIP(hipSetDevice(g->devices[r]));
    SRT_HIP(hipDeviceSynchronize());                     // (between renders: nothing is in flight that matters)
    SRT_HIP(hipMemset(g->d_acc[r], 0, g->acc_floats[r] * sizeof(float)));
    SRT_HIP(hipDeviceSynchronize());
    g->fold_recorded[r] = g->image_recorded[r] = 0;
  }
  return SRT_OK;
}

int srt_pt_group_max_samples_per_launch(srt_pt_group* g, uint32_t* samples) {
  if (!g || !samples) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_max_samples_per_launch: NULL argument");
  uint32_t least = 0xFFFFFFFFu;
  for (srt_pt* c : g->ctx) {
    uint32_t m = 0;
    const int st = srt_pt_max_samples_per_launch(c, &m);
    if (st != SRT_OK) return st;
    least = std::min(least, m);
  }
  *samples = least;
  return SRT_OK;
}

int srt_pt_group_render_samples(srt_pt_group* g, int lane, uint64_t seed, uint32_t sample_base, uint32_t samples) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_render_samples: NULL group");
  if (!g->image_floats) return srt::fail(SRT_ERR_STATE, "srt_pt_group_render_samples before srt_pt_group_set_params");
  if (lane == kDisplayLane) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_render_samples: lane %d belongs to srt_pt_group_accumulator_image", lane);
  int st = ensure_lane(g, lane);
  if (st != SRT_OK) return st;
  GroupLane& L = g->lanes[lane];
  for (size_t r = 0; r < g->ctx.size(); r++) {
    SRT_HIP(hipSetDevice(g->devices[r]));
    if ((st = srt_pt_render_samples_device(g->ctx[r], (void*)L.streams[r], seed, sample_base, samples)) != SRT_OK) return st;
  }
  return SRT_OK;
}

int srt_pt_group_wait_lane(srt_pt_group* g, int lane) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_wait_lane: NULL group");
  if (lane < 0 || lane >= (int)g->lanes.size()) return SRT_OK;
  GroupLane& L = g->lanes[lane];
  for (size_t r = 0; r < g->ctx.size(); r++) {
    SRT_HIP(hipSetDevice(g->devices[r]));
    SRT_HIP(hipStreamSynchronize(L.streams[r]));
    const int st = srt::pt_check_fault(g->ctx[r], "srt_pt_group_wait_lane");
    if (st != SRT_OK) return st;
  }
  return SRT_OK;
}

int srt_pt_group_fold(srt_pt_group* g, int lane, uint32_t samples_per_epoch, uint32_t position, uint32_t total_samples, uint32_t accumulator_samples) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_fold: NULL group");
  if (lane < 0 || lane >= (int)g->lanes.size() || lane == kDisplayLane) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_fold: nothing was rendered on lane %d", lane);
  int st = ensure_accumulators(g);
  if (st != SRT_OK) return st;
  GroupLane& L = g->lanes[lane];
  for (size_t r = 0; r < g->ctx.size(); r++) {
    SRT_HIP(hipSetDevice(g->devices[r]));
    // in launch order behind the previous fold (another lane's stream) and behind whoever is reading the accumulator
    if (g->fold_recorded[r]) SRT_HIP(hipStreamWaitEvent(L.streams[r], g->fold_done[r], 0));
    if (g->image_recorded[r]) SRT_HIP(hipStreamWaitEvent(L.streams[r], g->image_done[r], 0));
    if ((st = srt_pt_fold_epochs_device(g->ctx[r], (void*)L.streams[r], samples_per_epoch, position, total_samples, accumulator_samples, g->d_acc[r])) != SRT_OK) return st;
    SRT_HIP(hipEventRecord(g->fold_done[r], L.streams[r]));
    g->fold_recorded[r] = 1;
  }
  return SRT_OK;
}

int srt_pt_group_accumulator_image(srt_pt_group* g, float** d_image_out, void** stream_out) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_accumulator_image: NULL group");
  if (!g->image_floats) return srt::fail(SRT_ERR_STATE, "srt_pt_group_accumulator_image before srt_pt_group_set_params");
  int st = ensure_lane(g, kDisplayLane);
  if (st == SRT_OK) st = ensure_accumulators(g);
  if (st != SRT_OK) return st;
  GroupLane& L = g->lanes[kDisplayLane];
  for (size_t r = 0; r < g->ctx.size(); r++) {
    SRT_HIP(hipSetDevice(g->devices[r]));
    if (g->fold_recorded[r]) SRT_HIP(hipStreamWaitEvent(L.streams[r], g->fold_done[r], 0));
    if ((st = srt_pt_accumulator_tiles_device(g->ctx[r], (void*)L.streams[r], g->d_acc[r], L.d_tiles[r])) != SRT_OK) return st;
    SRT_HIP(hipEventRecord(g->image_done[r], L.streams[r]));
    g->image_recorded[r] = 1;
  }
  if ((st = exchange(g, L)) != SRT_OK) return st;
  if (g->ctx.size() > 1 && !g->use_rccl) {
    SRT_HIP(hipEventRecord(L.events[0], L.streams[0]));
    for (size_t r = 1; r < g->ctx.size(); r++) { SRT_HIP(hipSetDevice(g->devices[r])); SRT_HIP(hipStreamWaitEvent(L.streams[r], L.events[0], 0)); }
    SRT_HIP(hipSetDevice(g->devices[0]));
  }
  if (d_image_out) *d_image_out = L.d_image;
  if (stream_out) *stream_out = (void*)L.streams[0];
  return SRT_OK;
}

int srt_pt_group_cancel_requested(srt_pt_group* g) {
  if (!g) return 0;
  for (srt_pt* c : g->ctx)
    if (srt_pt_cancel_requested(c)) return 1;
  return 0;
}

int srt_pt_group_cancel(srt_pt_group* g) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_cancel: NULL group");
  for (srt_pt* c : g->ctx) (void)srt_pt_cancel(c);
  return SRT_OK;
}

int srt_pt_group_clear_cancel(srt_pt_group* g) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_clear_cancel: NULL group");
  for (srt_pt* c : g->ctx) { const int st = srt_pt_clear_cancel(c); if (st != SRT_OK) return st; }
  return SRT_OK;
}

int srt_pt_group_set_ray_log(srt_pt_group* g, uint32_t capacity) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_set_ray_log: NULL group");
  for (srt_pt* c : g->ctx) { const int st = srt_pt_set_ray_log(c, capacity); if (st != SRT_OK) return st; }
  return SRT_OK;
}

int srt_pt_group_read_ray_log(srt_pt_group* g, int lane, srt_pt_logged_ray* out, size_t cap, size_t* n_out, uint64_t* dropped) {
  if (!g) return srt::fail(SRT_ERR_INVALID, "srt_pt_group_read_ray_log: NULL group");
  if (n_out) *n_out = 0;
  if (dropped) *dropped = 0;
  if (lane < 0 || lane >= (int)g->lanes.size()) return SRT_OK;          // a lane that never rendered has logged nothing
  GroupLane& L = g->lanes[lane];
  std::vector<srt_pt_logged_ray> all;
  size_t total_waiting = 0;
  for (size_t r = 0; r < g->ctx.size(); r++) {
    SRT_HIP(hipSetDevice(g->devices[r]));
    size_t waiting = 0;
    int st = srt_pt_read_ray_log_stream(g->ctx[r], (void*)L.streams[r], nullptr, 0, &waiting, nullptr);   // how many (nothing consumed)
    if (st != SRT_OK) return st;
    total_waiting += waiting;
    if (!waiting || !out) continue;
    const size_t at = all.size();
    all.resize(at + waiting);
    size_t got = 0;
    uint64_t d = 0;
    if ((st = srt_pt_read_ray_log_stream(g->ctx[r], (void*)L.streams[r], all.data() + at, waiting, &got, &d)) != SRT_OK) return st;
    all.resize(at + got);
    if (dropped) *dropped += d;
  }
  if (!out) { if (n_out) *n_out = total_waiting; return SRT_OK; }
  std::sort(all.begin(), all.end(), [](const srt_pt_logged_ray& a, const srt_pt_logged_ray& b) {
    if (a.pixel != b.pixel) return a.pixel < b.pixel;
    if (a.sample != b.sample) return a.sample < b.sample;
    return a.bounce < b.bounce;
  });
  const size_t n = std::min(all.size(), cap);
  if (n) std::memcpy(out, all.data(), n * sizeof(srt_pt_logged_ray));
  if (dropped) *dropped += all.size() - n;
  if (n_out) *n_out = n;
  return SRT_OK;
}

}  // extern "C"

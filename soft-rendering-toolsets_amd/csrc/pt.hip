// MI355X (gfx950) path tracer: Pathtracer::trace_pixel and everything below it as HIP kernels, plus the
// C ABI of include/srt_pt.h.  Reference paths are relative to /root/reference/Assignments/Scotty3D/src/.
//
//   trace_pixel               student/pathtracer.cpp:14-40      -> path_sample()
//   trace                     student/pathtracer.cpp:174-218    -> the bounce loop of path_sample()
//   sample_direct_lighting    student/pathtracer.cpp:78-172     -> direct block (BSDF ray + MIS ray)
//   sample_indirect_lighting  student/pathtracer.cpp:42-76      -> per-bounce record folded bottom-up
//   BVH<>::hit                student/bvh.inl:166-276           -> traverse<>() (explicit-stack form of the recursion)
//   Object::hit               rays/object.h:57-65               -> object_hit()
//   sample_area_lights / area_lights_pdf  rays/pathtracer.cpp:301-325 -> light_sample() / light_pdf()
//
// The reference recursion `L = direct + (L_next * atten) * (1/pdf)` is evaluated leaf-first; to round
// identically the kernel records (direct, atten, 1/pdf) per bounce and folds the records from the last
// bounce back to the first instead of carrying a running throughput.
#include <cstdlib>
#include <cstdio>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <map>
#include <vector>

#include "pt_device.h"
#include "pt_scene.h"
#include "srt_common.h"
#include "srt_pt.h"
#include "srt_pt_debug.h"
#include "pt_internal.h"

#include "pt_trace.h"
#include "pt_wave.h"

namespace srt {

// ---------------------------------------------------------------------------------------------------
// Kernels
// ---------------------------------------------------------------------------------------------------

// One lane per pixel of this rank's tiles; the lane walks the epoch's samples in order so the per-pixel
// sum is accumulated exactly like do_trace (rays/pathtracer.cpp:216-226).
__global__ __launch_bounds__(64) void pt_epoch_kernel(DScene S, TileMap T, uint64_t seed, uint32_t sample_base,
                                                      uint32_t samples, float* __restrict__ tiles_out,
                                                      unsigned long long* __restrict__ ray_counter, const uint32_t* host_cancel, uint32_t* dev_cancel) {
  // srt_pt_cancel: the first block looks at the host's flag, the others at its device copy - blocks that have not started yet end at once
  if (blockIdx.x == 0u && threadIdx.x == 0u && cancel_requested(host_cancel)) atomicExch(dev_cancel, 1u);
  if (cancel_raised(dev_cancel)) return;
  const uint32_t px_per_tile = T.tile_w * T.tile_h;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t local_tile = gid / px_per_tile;
  const uint32_t in_tile = gid % px_per_tile;
  if (local_tile >= T.local_tiles) return;
  // 8x8 pixel blocks per wavefront inside the tile: neighbouring pixels take similar paths
  const uint32_t blocks_x = T.tile_w / 8;
  const uint32_t blk = in_tile / 64, lane = in_tile % 64;
  const uint32_t lx = (blk % blocks_x) * 8 + (lane % 8), ly = (blk / blocks_x) * 8 + (lane / 8);
  const uint32_t tile = T.rank + local_tile * T.world;
  const uint32_t x = (tile % T.tiles_x) * T.tile_w + lx, y = (tile / T.tiles_x) * T.tile_h + ly;
  float* out = tiles_out + ((size_t)local_tile * px_per_tile + (size_t)ly * T.tile_w + lx) * 3;
  Counters cnt;
  cnt.v[C_RAYS] = 0;
  const bool inside = x < S.w && y < S.h;
  if (!inside) samples = 0;  // padding lanes of edge tiles: no samples, zero output, still join the wave reduction
  Rng rng;
  Spec acc = spec(0, 0, 0);
  uint32_t sampled = 0;
  for (uint32_t s = 0; s < samples; s++) {
    rng.key(seed, y * S.w + x, sample_base + s);
    const Spec p = path_sample<false>(S, x, y, rng, cnt);
    if (valid(p)) { acc = acc + p; sampled++; }
  }
  if (sampled > 0) acc = acc * (1.0f / sampled);
  out[0] = acc.r; out[1] = acc.g; out[2] = acc.b;
  if (ray_counter) {  // one atomic per wavefront (every lane of the wave reaches this point)
    unsigned long long r = cnt.v[C_RAYS], e = cnt.elided;
    for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off); e += __shfl_down(e, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(ray_counter, r); if (e) atomicAdd(ray_counter + 1, e); }
  }
}

// One lane per (pixel, sample) unit: the hardware's wave scheduler balances the load (scenes whose rays differ
// wildly in cost, e.g. a 100k-triangle mesh inside the Cornell box); radiance goes to the per-unit buffer and
// pt_reduce_kernel adds a pixel's samples in order.  Lanes of a wave hold consecutive samples of one pixel.
// 8 waves/SIMD (64 VGPRs, the rest spilled): the kernel waits on memory two thirds of the time, and occupancy hides
// that better than registers do (131 k-triangle scene: 325 -> 417 Mrays/s).
__global__ __launch_bounds__(64, 8) void pt_unit_kernel(DScene S, TileMap T, uint64_t seed, uint32_t sample_base, uint32_t samples,
                                                     uint32_t total_units, float* __restrict__ sample_out,
                                                     unsigned long long* __restrict__ ray_counter, const uint32_t* host_cancel, uint32_t* dev_cancel) {
  if (blockIdx.x == 0u && threadIdx.x == 0u && cancel_requested(host_cancel)) atomicExch(dev_cancel, 1u);   // (as pt_epoch_kernel)
  if (cancel_raised(dev_cancel)) return;
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  Counters cnt;
  cnt.v[C_RAYS] = 0;
  if (u < total_units) {
    uint32_t x, y;
    unit_pixel(T, u / samples, x, y);
    if (x < S.w && y < S.h) {
      Rng rng;
      rng.key(seed, y * S.w + x, sample_base + u % samples);
      const Spec p = path_sample<false>(S, x, y, rng, cnt);
      reinterpret_cast<float4*>(sample_out)[(size_t)(u % samples) * (total_units / samples) + u / samples] = make_float4(p.r, p.g, p.b, 0.0f);
    }
  }
  unsigned long long r = cnt.v[C_RAYS], e = cnt.elided;
  for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off); e += __shfl_down(e, off); }
  if ((threadIdx.x & 63) == 0 && r) { atomicAdd(ray_counter, r); if (e) atomicAdd(ray_counter + 1, e); }
}

// Explicit (x, y, sample) triples; instrumented when COUNT.
template <bool COUNT>
__global__ __launch_bounds__(64) void pt_samples_kernel(DScene S, uint64_t seed, const uint32_t* __restrict__ xs,
                                                        const uint32_t* __restrict__ ys, const uint32_t* __restrict__ ss,
                                                        uint32_t n, float* __restrict__ rgb, uint32_t* __restrict__ draws,
                                                        uint32_t* __restrict__ rays, unsigned long long* __restrict__ totals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Counters cnt;
  for (int k = 0; k < C_COUNT; k++) cnt.v[k] = 0;
  Rng rng;
  rng.key(seed, ys[i] * S.w + xs[i], ss[i]);
  const Spec p = path_sample<COUNT>(S, xs[i], ys[i], rng, cnt);
  rgb[3 * i] = p.r; rgb[3 * i + 1] = p.g; rgb[3 * i + 2] = p.b;
  if (draws) draws[i] = rng.draws;
  if (COUNT) {
    if (rays) rays[i] = cnt.v[C_RAYS];
    for (int k = 0; k < C_COUNT; k++) atomicAdd(&totals[k], (unsigned long long)cnt.v[k]);
  }
}

// FLAT: the query goes through flat_trace3 (pt_flat.h) in batch slot i % 3 instead of the nested scene_hit.
template <bool FLAT>
__global__ void pt_hit_kernel(DScene S, const float* __restrict__ org, const float* __restrict__ dir,
                              const float* __restrict__ bounds, uint32_t n, float* __restrict__ out9) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;                     // every lane stays in the kernel: flat_trace3 is a wave-wide loop
  const uint32_t j = live ? i : 0;
  Ray r;
  r.o = v3(org[3 * j], org[3 * j + 1], org[3 * j + 2]);
  r.d = v3(dir[3 * j], dir[3 * j + 1], dir[3 * j + 2]);
  r.b0 = bounds[2 * j]; r.b1 = bounds[2 * j + 1];
  Counters cnt;
  Hit h;
  if (FLAT) {
    const uint32_t slot = i % 3u;
    Hit res[3];
    flat_trace3(S, r.o, r.d, r.d, r.d, r.b0, r.b1, live && slot == 0, live && slot == 1, live && slot == 2, res[0], res[1], res[2]);
    h = slot == 0 ? res[0] : (slot == 1 ? res[1] : res[2]);
  } else {
    h = scene_hit<false>(S, r, cnt);
  }
  if (!live) return;
  float* o = out9 + 9 * i;
  for (int k = 0; k < 9; k++) o[k] = 0.0f;
  if (h.hit) {
    const Surface sf = surface_of(S, h, r);
    o[0] = 1.0f; o[1] = h.dist;
    o[2] = sf.position.x; o[3] = sf.position.y; o[4] = sf.position.z;
    o[5] = sf.normal.x; o[6] = sf.normal.y; o[7] = sf.normal.z;
    o[8] = (float)S.objects[h.obj].material;
  }
}

// Scene_Particles::Particle::update (student/particles.cpp:5-59), one lane per particle: the second caller of Object::hit
// (SURVEY.md 8(f)-4).  The particle flies at constant velocity for what is left of dt, bounces off what
// scene.hit(Ray(pos, velocity)) reports - default bounds [0, inf], the direction is the velocity, NOT normalised - and
// gravity acts on the velocity after every leg.  The reference's loop does not return when hit_time stays <= 0; a lane
// gives up after kParticleMaxLegs legs.
constexpr uint32_t kParticleMaxLegs = 4096;
__global__ __launch_bounds__(64) void pt_particles_kernel(DScene S, float* __restrict__ pos, float* __restrict__ vel, float* __restrict__ age,
                                                          uint32_t n, float dt, float radius, uint8_t* __restrict__ alive) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  V3 p = v3(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]);
  V3 velocity = v3(vel[3 * k], vel[3 * k + 1], vel[3 * k + 2]);
  const V3 acceleration = v3(0.0f, -9.8f, 0.0f);
  float remain = dt;
  Counters cnt;
  for (uint32_t leg = 0; remain > 0 && leg < kParticleMaxLegs; leg++) {
    Ray r;
    r.o = p; r.d = velocity; r.b0 = 0.0f; r.b1 = __uint_as_float(0x7f800000u);
    const Hit h = scene_hit<false>(S, r, cnt);
    V3 t_normal = v3(0, 0, 0), t_position = v3(0, 0, 0);       // a Trace without a hit is all zeros (rays/trace.h)
    if (h.hit) { const Surface sf = surface_of(S, h, r); t_normal = sf.normal; t_position = sf.position; }
    float cos_t = dot(t_normal, velocity * -1.0f) / (norm(velocity) * norm(t_normal));
    V3 surface_normal = t_normal / norm(t_normal);
    if (cos_t < 0) {
      cos_t = (float)sqrt((double)(1 - cos_t * cos_t));          // unqualified sqrt: the double overload
      surface_normal = surface_normal * -1.0f;
    }
    const float interval = fabsf(radius / cos_t);
    const float hit_time = (h.dist - interval) / norm(r.d);
    if (!h.hit || hit_time > remain || cos_t == 0) {
      p = p + velocity * remain;
      velocity = velocity + acceleration * remain;
      break;
    }
    p = t_position - (velocity * interval) / norm(velocity);
    velocity = velocity - surface_normal * (2.0f * dot(velocity, surface_normal));
    velocity = velocity + acceleration * hit_time;
    remain -= hit_time;
  }
  const float a = age[k] - dt;
  age[k] = a;
  alive[k] = a > 0 ? 1 : 0;
  pos[3 * k] = p.x; pos[3 * k + 1] = p.y; pos[3 * k + 2] = p.z;
  vel[3 * k] = velocity.x; vel[3 * k + 1] = velocity.y; vel[3 * k + 2] = velocity.z;
}

__global__ void pt_untile_kernel(TileMap T, uint32_t w, uint32_t h, uint32_t tiles_per_rank,
                                 const float* __restrict__ gathered, float* __restrict__ image) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const uint32_t x = i % w, y = i / w;
  const uint32_t tile = (y / T.tile_h) * T.tiles_x + (x / T.tile_w);
  const uint32_t rank = tile % T.world, local = tile / T.world;
  const size_t src = (((size_t)rank * tiles_per_rank + local) * (T.tile_w * T.tile_h) + (size_t)(y % T.tile_h) * T.tile_w +
                      (x % T.tile_w)) * 3;
  image[3 * (size_t)i] = gathered[src];
  image[3 * (size_t)i + 1] = gathered[src + 1];
  image[3 * (size_t)i + 2] = gathered[src + 2];
}

// Pathtracer::accumulate: s += (n - s) * (1.0f / accumulator_samples)
__global__ void pt_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ epoch, size_t n, float inv) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] += (epoch[i] - acc[i]) * inv;
}

__global__ void pt_math_kernel(const float* __restrict__ x, size_t n, float* __restrict__ c, float* __restrict__ s) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { c[i] = srt_cosf(x[i]); s[i] = srt_sinf(x[i]); }
}

__global__ void pt_acos_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = srt_acosf(x[i]);
}

__global__ void pt_atan2_kernel(const float* __restrict__ y, const float* __restrict__ x, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = srt_atan2f(y[i], x[i]);
}

// HDR_Image::tonemap_to (util/hdr_image.cpp:161-187): output row j is image row h-1-j; per channel
// 1 - exp(-c * exposure), Spectrum::to_srgb, (unsigned char)round(c * 255); alpha 255.  One lane per pixel, one packed
// 32-bit store (consecutive lanes: consecutive pixels of an output row, 12-byte loads / 4-byte stores, fully coalesced).
__global__ void pt_tonemap_kernel(const float* __restrict__ rgb, uint32_t w, uint32_t h, float exposure, uint32_t* __restrict__ rgba) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)w * h) return;
  const uint32_t j = (uint32_t)(p / w), i = (uint32_t)(p % w);
  const float* s = rgb + 3 * ((size_t)(h - j - 1) * w + i);
  uint32_t px = 0xff000000u;
#pragma unroll
  for (int c = 0; c < 3; c++) px |= srgb_byte(to_srgb(1.0f - srt_expf(-s[c] * exposure))) << (8 * c);
  rgba[p] = px;
}
__global__ void pt_exp_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = srt_expf(x[i]);
}
__global__ void pt_pow_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = srt_powf(x[i], y[i]);
}

// Diagnostic for divNx3 / sqrtN (pt_device.h): lane i takes the operands of "rays" 3i, 3i+1, 3i+2 — exactly how the
// wave kernel's batch tests call them.  io layout: planes of n3 = 3 * lanes floats: num0, num1, num2, den, x in; q0, q1, q2,
// root out.  shared_c2: column 2's numerator of a lane is num2[3i] for its three rays (the triangle test's shared t numerator).
__global__ void pt_div_sqrt_kernel(const float* __restrict__ in, size_t lanes, int shared_c2, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n3 = lanes * 3;
  const size_t k = (i < lanes ? i : lanes - 1) * 3;     // every lane of the last wave computes; the spares repeat the last lane
  float num[3][3], den[3], q[3][3], x[3], root[3];
  bool zero[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    num[r][0] = in[k + r]; num[r][1] = in[n3 + k + r];
    num[r][2] = in[2 * n3 + (shared_c2 ? k : k + r)];
    den[r] = in[3 * n3 + k + r];
    x[r] = in[4 * n3 + k + r];
    zero[r] = __float_as_uint(x[r]) == 0u;
  }
  if (shared_c2) divNx3<3, true>(num, den, q); else divNx3<3, false>(num, den, q);
  sqrtN<3>(x, zero, root);
  if (i < lanes) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
      out[k + r] = q[r][0]; out[n3 + k + r] = q[r][1]; out[2 * n3 + k + r] = q[r][2];
      out[3 * n3 + k + r] = root[r];
    }
  }
}

}  // namespace srt

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
using namespace srt;

struct srt_pt {
  int device = -1;            // -1: host-only context (scene assembly / BVH inspection, no rendering)
  hipStream_t stream = nullptr;
  std::vector<ObjectInput> inputs;
  std::vector<Material> materials;
  BuiltScene built;
  bool committed = false;
  Camera cam{};
  bool have_cam = false;
  uint32_t w = 0, h = 0, max_depth = 8;
  TileMap tiles{32, 32, 0, 0, 0, 1, 0};
  uint32_t tiles_per_rank = 0;
  // device copies
  Node* d_nodes = nullptr; Tri* d_tris = nullptr; TriNrm* d_nrm = nullptr; Object* d_objects = nullptr;
  float* d_tri_packed = nullptr;                                 // the triangle records without padding (pt_scene.h): what the cast kernel reads
  Light* d_lights = nullptr; LightTri* d_ltris = nullptr; Material* d_mats = nullptr;
  WaveInterior* d_wave = nullptr; WaveInterior* d_blas = nullptr; uint32_t* d_wave_lazy = nullptr;
  DeltaLight* d_dlights = nullptr;
  std::vector<DeltaLight> delta_lights;   // srt_pt_add_light, in call order
  uint32_t env_type = 0; float env_radiance[3] = {0, 0, 0};   // srt_pt_set_env_light
  std::vector<float> env_map; uint32_t env_w = 0, env_h = 0; float* d_env_map = nullptr;   // srt_pt_set_env_map
  float* d_tile_buf = nullptr; size_t tile_buf_floats = 0;
  float* d_image = nullptr; size_t image_floats = 0;
  int kernel_mode = 0;        // 0 auto, 1 general per-lane kernel, 2 wave-uniform persistent kernel
  // Scratch of one epoch in flight.  One set per stream the caller renders on: epochs launched on different streams
  // may overlap on the device (the next epoch's blocks fill the CUs the previous launch's tail leaves idle).
  struct EpochBuffers {
    float* d_samples = nullptr; size_t samples_floats = 0;   // per-sample radiance
    float* d_records = nullptr; size_t records_floats = 0;   // wave kernel: per-bounce records
    float* d_running = nullptr; size_t running_floats = 0;   // (sum, count) across the launches of one epoch
    unsigned long long* d_queue = nullptr;                   // wave kernel: queue head (+ section stamps)
    // streamed form (pt_stream.h): saved path state, ray queue, hits, counters
    uint32_t* d_state = nullptr; size_t state_words = 0;
    float4* d_ray_o = nullptr; size_t ray_o_n = 0;
    float4* d_ray_d = nullptr; size_t ray_d_n = 0;
    uint32_t* d_ray_id = nullptr; size_t ray_id_n = 0;
    uint2* d_hits = nullptr; size_t hits_n = 0;
    StreamCounters* d_sc = nullptr;
    unsigned long long* d_block_counters = nullptr; size_t block_counters_n = 0;
    uint32_t* d_cast_spill = nullptr; size_t cast_spill_words = 0;   // the ray-cast kernel's traversal frames beyond those in LDS
    uint32_t* d_cancel = nullptr;                                     // srt_pt_cancel as the kernels of this stream have seen it (sticky until srt_pt_clear_cancel)
    uint32_t* d_ray_log = nullptr; uint32_t ray_log_cap = 0;          // srt_pt_set_ray_log: this stream's ring (pt_trace.h: log_ray_event)
    uint32_t* d_alive_list = nullptr; size_t alive_list_n = 0;          // streamed forms: the alive slots the next generation works from
    uint32_t last_samples = 0, last_npix = 0;                         // what d_samples holds: samples per pixel and pixel slots of the last launch
  };
  std::map<hipStream_t, EpochBuffers> epoch_buffers;
  int wave_blocks = 0; size_t wave_lds = 0; int wave_mode = -1; const void* wave_kern = nullptr;
  const void* cast_kern = nullptr; uint32_t cast_lds_frames = 0;                                 // traversal frames per lane kept in LDS (the deeper ones: d_cast_spill)
  int cast_blocks = 0, cast_threads = 0; size_t cast_lds = 0; uint32_t cast_depth = 0;   // pt_cast_kernel's launch shape (0: not derived yet)
  int bvh_builder = 1; uint32_t bvh_device_min = 16384;         // srt_pt_set_bvh_builder: device build for sets of >= this many primitives
  uint32_t stream_slots = 0;                                    // srt_pt_set_stream_slots (0: default)
  unsigned long long* d_cast_stats = nullptr;                   // SRT_CAST_STATS=1: the STATS build of pt_cast_kernel adds into these
  unsigned long long* d_totals = nullptr;   // C_COUNT instrumented totals + 4 slots: rays of the epoch kernels, rays elided, streamed forms: entries queued, alive slot-generations
  uint32_t* h_fault = nullptr;              // pinned, device-visible: bit 0 = a streamed launch ended with unfinished units (sticky until reported)
  uint32_t* d_fault = nullptr;              // its device address
  uint32_t* h_cancel = nullptr;             // pinned, device-visible: srt_pt_cancel's flag (any host thread may set it)
  uint32_t* d_host_cancel = nullptr;        // its device address
  uint32_t ray_log_cap = 0;                 // srt_pt_set_ray_log: rays per stream and read; 0: Pathtracer::log_ray is not delivered
  int elide = 0;                            // srt_pt_set_elision
  unsigned long long last_counters[C_COUNT] = {0};
  uint64_t camera_samples = 0;
  // srt_pt_kernel_time: event pairs recorded around the dominant kernel's launches, on the launch stream
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;   // pending (recorded, not yet read)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> spare;
  // srt_pt_stream_times: per-kernel event pairs of the streamed form {logic, compaction, ray cast}
  bool stream_timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> stream_timed[4];   // {logic (resolve), compaction, ray cast, probe}
  uint64_t stream_generations = 0;
};

namespace {

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
  if (*dst) { SRT_HIP(hipFree(*dst)); *dst = nullptr; }
  const size_t n = src.empty() ? 1 : src.size();
  SRT_HIP(hipMalloc(dst, n * sizeof(T)));
  if (!src.empty()) SRT_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return SRT_OK;
}

int need_device(srt_pt* pt, const char* what) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "%s: NULL context", what);
  if (pt->device < 0) return srt::fail(SRT_ERR_NO_DEVICE, "%s needs a HIP device; this context is host-only and there is no CPU fallback", what);
  SRT_HIP(hipSetDevice(pt->device));
  return SRT_OK;
}

// After a synchronisation: did a streamed launch end with unfinished units (pt_stream_finish_kernel)?  Reported once.
int check_stream_fault(srt_pt* pt, const char* what) {
  if (pt->h_fault && *(volatile uint32_t*)pt->h_fault != 0u) {
    *(volatile uint32_t*)pt->h_fault = 0u;
    return srt::fail(SRT_ERR_STATE, "%s: a streamed launch ended before every work unit was finished (generation bound too small); the epoch's image is invalid", what);
  }
  return SRT_OK;
}

int need_ready(srt_pt* pt, const char* what) {
  int st = need_device(pt, what);
  if (st != SRT_OK) return st;
  if (!pt->committed) return srt::fail(SRT_ERR_STATE, "%s before srt_pt_scene_commit", what);
  if (!pt->have_cam) return srt::fail(SRT_ERR_STATE, "%s before srt_pt_set_camera", what);
  if (!pt->w || !pt->h) return srt::fail(SRT_ERR_STATE, "%s before srt_pt_set_params", what);
  return SRT_OK;
}

// srt_pt_set_elision asked for it and the BSDF-sampled direct ray of every continuous bounce is provably dead in this scene:
// no delta light, no environment light, and no light material without emission (it would be shaded as a continuous,
// non-Lambertian BSDF).  include/srt_pt.h, pt_wave.h.
bool elision_provable(const srt_pt* pt) {
  const FlatScene& F = pt->built.flat;
  if (!pt->elide || !F.delta_lights.empty() || pt->env_type != 0) return false;
  for (const Material& m : F.materials) {
    if (m.type == 3u && !(0.2126f * m.a[0] + 0.7152f * m.a[1] + 0.0722f * m.a[2] > 0.0f)) return false;
    // the dead term is emission x albedo x cos x 1/pdf with 1/pdf <= pi / cos(1): finite as long as the factors are moderate
    if (m.type == 0u || m.type == 3u)
      for (int i = 0; i < 3; i++)
        if (!(fabsf(m.a[i]) <= 1e15f)) return false;
  }
  return true;
}

DScene device_scene(const srt_pt* pt, const srt_pt::EpochBuffers* B = nullptr) {
  const FlatScene& F = pt->built.flat;
  DScene S;
  S.ray_log = B ? B->d_ray_log : nullptr; S.ray_log_cap = B ? B->ray_log_cap : 0u;
  S.nodes = pt->d_nodes; S.tris = pt->d_tris; S.tri_nrm = pt->d_nrm; S.objects = pt->d_objects;
  S.lights = pt->d_lights; S.light_tris = pt->d_ltris; S.materials = pt->d_mats;
  S.wave_tlas = pt->d_wave; S.wave_q = (uint32_t)F.wave_tlas.size(); S.blas_recs = pt->d_blas; S.wave_lazy = pt->d_wave_lazy;
  S.delta_lights = pt->d_dlights; S.ndelta = (uint32_t)F.delta_lights.size();
  S.env_map = pt->d_env_map; S.env_w = pt->env_w; S.env_h = pt->env_h;
  S.env_type = pt->env_type; S.env_radiance[0] = pt->env_radiance[0]; S.env_radiance[1] = pt->env_radiance[1]; S.env_radiance[2] = pt->env_radiance[2];
  S.nobjects = (uint32_t)F.objects.size(); S.nlights = (uint32_t)F.lights.size();
  S.tlas_nodes = F.tlas_nodes; S.use_bvh = F.use_bvh ? 1u : 0u; S.light_tri_first = F.light_tri_first;
  S.cam = pt->cam; S.w = pt->w; S.h = pt->h; S.max_depth = pt->max_depth;
  S.elide = elision_provable(pt) ? 1u : 0u;
  return S;
}

void update_tiling(srt_pt* pt) {
  TileMap& T = pt->tiles;
  if (!pt->w || !pt->h) { T.tiles_x = T.tiles_y = T.local_tiles = 0; pt->tiles_per_rank = 0; return; }
  T.tiles_x = (pt->w + T.tile_w - 1) / T.tile_w;
  T.tiles_y = (pt->h + T.tile_h - 1) / T.tile_h;
  const uint32_t ntiles = T.tiles_x * T.tiles_y;
  pt->tiles_per_rank = (ntiles + T.world - 1) / T.world;
  T.local_tiles = (ntiles > T.rank) ? (ntiles - T.rank + T.world - 1) / T.world : 0;
}

}  // namespace

namespace {

template <typename T>
int ensure(T** buf, size_t* have, size_t need) {
  if (*have >= need && *buf) return SRT_OK;
  if (*buf) { SRT_HIP(hipFree(*buf)); *buf = nullptr; *have = 0; }
  SRT_HIP(hipMalloc(buf, need * sizeof(T)));
  *have = need;
  return SRT_OK;
}

// The scratch set of stream `s` (created on first use), with what every kernel form needs: the stream's sticky cancel word and,
// when srt_pt_set_ray_log asked for one, its ray-log ring.
int stream_buffers(srt_pt* pt, hipStream_t s, srt_pt::EpochBuffers** out) {
  srt_pt::EpochBuffers& B = pt->epoch_buffers[s];
  // (hipMemsetAsync ON `s`: a plain hipMemset is ordered on the null stream only, which the callers' non-blocking streams do not wait
  //  for - a kernel on `s` could still see what the allocation held before)
  if (!B.d_cancel) {
    SRT_HIP(hipMalloc(&B.d_cancel, sizeof(uint32_t)));
    SRT_HIP(hipMemsetAsync(B.d_cancel, 0, sizeof(uint32_t), s));
  }
  if (B.ray_log_cap != pt->ray_log_cap) {
    if (B.d_ray_log) { SRT_HIP(hipStreamSynchronize(s)); SRT_HIP(hipFree(B.d_ray_log)); B.d_ray_log = nullptr; }
    B.ray_log_cap = 0;
    if (pt->ray_log_cap) {
      const size_t words = kRayLogHeader + (size_t)kRayLogWords * pt->ray_log_cap;
      SRT_HIP(hipMalloc(&B.d_ray_log, words * sizeof(uint32_t)));
      SRT_HIP(hipMemsetAsync(B.d_ray_log, 0, kRayLogHeader * sizeof(uint32_t), s));
      B.ray_log_cap = pt->ray_log_cap;
    }
  }
  *out = &B;
  return SRT_OK;
}

// Which traversal the persistent wave kernel would use for this scene and kernel mode: 0 wave-uniform sweeps,
// 1 sweeps + per-lane walk of each BVH<Triangle>, 2 flattened per-lane walk (pt_flat.h); -1: not the wave kernel.
// The flattened walk (pt_flat.h) packs a hit as 5 + 27 bits and a TLAS leaf's object count in 3 bits.
bool flat_walk_fits(const FlatScene& F) {
  bool fits = F.objects.size() >= 1 && F.objects.size() <= 31 && F.tris.size() < (1u << 27);
  for (const WaveInterior& w : F.wave_tlas)
    if ((w.l_ref < 0 && w.l_cnt > kFlatMaxLeafObjects) || (w.r_ref < 0 && w.r_cnt > kFlatMaxLeafObjects)) fits = false;
  return fits;
}

// The streamed form (pt_stream.h): any number of objects, as long as a hit still packs into one word
// (object slot << shift | triangle) and the references fit the 30-bit field of an LDS frame.
uint32_t stream_obj_shift(const FlatScene& F) {
  uint32_t bits = 1;
  while ((1ull << bits) <= F.objects.size()) bits++;      // the object field is never all ones: 0xFFFFFFFF / ..FE stay free
  return 32u - bits;
}
bool stream_fits(const FlatScene& F) {
  if (F.objects.empty() || F.objects.size() >= (1u << 24)) return false;
  if (F.tris.size() >= (1ull << stream_obj_shift(F)) || F.tris.size() >= (1u << 26) || F.blas_recs.size() >= (1u << 29)) return false;
  for (const WaveInterior& w : F.wave_tlas)
    if ((w.l_ref < 0 && w.l_cnt > kFlatMaxLeafObjects) || (w.r_ref < 0 && w.r_cnt > kFlatMaxLeafObjects)) return false;
  return true;
}

int wave_trav(const srt_pt* pt) {
  const FlatScene& F = pt->built.flat;
  const int m = pt->kernel_mode;
  if (m == 1 || m == 4) return -1;
  // delta lights (point_lighting's shadow batches) and environment lights: the sweeps' DL instantiation only
  const bool lights = !F.delta_lights.empty() || pt->env_type != 0;
  const bool blas = !F.blas_recs.empty();
  const bool sweeps_fit = F.objects.size() >= 1 && F.objects.size() <= kWaveMaxObjects && F.tris.size() < (1u << 27);
  const bool flat_fits = flat_walk_fits(F);
  if (m == 2) return sweeps_fit ? (blas ? 1 : 0) : -1;
  if (m == 3) return (sweeps_fit && !lights) ? (blas ? 1 : 0) : -1;
  if (m == 5) return (flat_fits && !lights) ? 2 : -1;
  bool single_leaves = true;          // (the reference builds BVH<Object> leaves of one object; the streamed sweeps rely on it)
  for (const WaveInterior& w : F.wave_tlas)
    if ((w.l_ref < 0 && w.l_cnt > 1u) || (w.r_ref < 0 && w.r_cnt > 1u)) single_leaves = false;
  const bool sweeps_stream = sweeps_fit && blas && single_leaves && F.lazy_objects.size() <= kMaxLazy && stream_fits(F);
  if (m == 6) return stream_fits(F) ? 3 : -1;
  if (m == 7) return sweeps_stream ? 4 : -1;
  // auto: a few objects, some of them meshes with a real BVH<Triangle> (BASELINE configs[4]): the sweeps stay, the walks of
  // those meshes are queued (streamed sweeps); more objects than the sweeps take: every ray through the ray-cast kernel
  if (sweeps_stream) return 4;
  if ((blas || !sweeps_fit) && stream_fits(F)) return 3;
  // auto: the sweeps whenever the scene has few enough objects.  Meshes with a real BVH<Triangle> are walked per lane
  // inside the sweeps, only by the rays that can reach them and compacted over the wave (object_testN): 407 / 771 / 1458
  // Mrays/s on the 131 k / 8 k / 512-triangle test scenes against 367 / 475 / 676 for the lane-per-sample kernel,
  // which remains the path for larger scenes (and is still ahead of the flattened walk there).
  if (sweeps_fit) return blas ? 1 : 0;
  return -1;
}
bool wave_kernel_applies(const srt_pt* pt) { return wave_trav(pt) >= 0; }

// Samples per pixel one launch handles: 64, fewer for very large shards so that the per-sample buffer (16 B per
// sample) stays under 2 GiB.
uint32_t samples_per_launch(uint32_t px) {
  const uint64_t fit = px ? (1ull << 27) / px : 64;
  return (uint32_t)(fit >= 64 ? 64 : (fit < 4 ? 4 : fit));
}

// One epoch with the wave-uniform persistent kernel: launches of <= 64 samples per pixel, each followed by the
// ordered per-pixel reduction.
// Timing brackets for the dominant kernel (srt_pt_kernel_time).
int time_begin(srt_pt* pt, hipStream_t s) {
  if (!pt->timing) return SRT_OK;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!pt->spare.empty()) { ev = pt->spare.back(); pt->spare.pop_back(); }
  else { SRT_HIP(hipEventCreate(&ev.first)); SRT_HIP(hipEventCreate(&ev.second)); }
  pt->timed.push_back(ev);
  SRT_HIP(hipEventRecord(ev.first, s));
  return SRT_OK;
}
// Brackets for one kernel of the streamed form (srt_pt_stream_times; diagnostic, off by default).
int stream_time_begin(srt_pt* pt, hipStream_t s, int which) {
  if (!pt->stream_timing) return SRT_OK;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!pt->spare.empty()) { ev = pt->spare.back(); pt->spare.pop_back(); }
  else { SRT_HIP(hipEventCreate(&ev.first)); SRT_HIP(hipEventCreate(&ev.second)); }
  pt->stream_timed[which].push_back(ev);
  SRT_HIP(hipEventRecord(ev.first, s));
  return SRT_OK;
}
int stream_time_end(srt_pt* pt, hipStream_t s, int which) {
  if (!pt->stream_timing) return SRT_OK;
  SRT_HIP(hipEventRecord(pt->stream_timed[which].back().second, s));
  return SRT_OK;
}
int time_end(srt_pt* pt, hipStream_t s) {
  if (!pt->timing) return SRT_OK;
  SRT_HIP(hipEventRecord(pt->timed.back().second, s));
  return SRT_OK;
}

int render_epoch_stream(srt_pt* pt, hipStream_t s, uint64_t seed, uint32_t sample_base, uint32_t samples, float* d_tiles_out);

int render_epoch_wave(srt_pt* pt, hipStream_t s, uint64_t seed, uint32_t sample_base, uint32_t samples, float* d_tiles_out) {
  const TileMap& T = pt->tiles;
  const uint32_t px = T.local_tiles * T.tile_w * T.tile_h;
  const FlatScene& F = pt->built.flat;
  const int trav = wave_trav(pt);
  if (trav >= 3) return render_epoch_stream(pt, s, seed, sample_base, samples, d_tiles_out);
  const bool stamp = pt->kernel_mode == 3;
  const size_t nq = (F.use_bvh && trav != 2) ? F.wave_tlas.size() : 0;
  const bool dl = !F.delta_lights.empty() || pt->env_type != 0;
  // two-ray batches (dead BSDF-sampled direct ray not traced): asked for, sweep build, no delta / environment light, and
  // every continuous BSDF Lambertian (pt_wave.h)
  const bool two = elision_provable(pt) && !stamp && trav != 2;
  const uint32_t burst = two ? 2u : kBurst;
  const size_t lds = (size_t)4 * (nq > 0 ? nq - 1 : 0) * (2 * burst) * 64 * sizeof(float)   // 4 waves x (Q - 1) x rays x 2 fields
                     + (cold_in_lds(trav) ? (size_t)kColdWords * 256 * sizeof(uint32_t) : 0); // + the lanes' cold path state
  const void* kern = two ? (trav == 0 ? (const void*)pt_wave_kernel<false, 0, false, 2> : (const void*)pt_wave_kernel<false, 1, false, 2>) : stamp ? (trav == 0 ? (const void*)pt_wave_kernel<true, 0, false, 3> : trav == 1 ? (const void*)pt_wave_kernel<true, 1, false, 3> : (const void*)pt_wave_kernel<true, 2, false, 3>)
                     : dl  ? (trav == 0 ? (const void*)pt_wave_kernel<false, 0, true, 3> : (const void*)pt_wave_kernel<false, 1, true, 3>)
                           : (trav == 0 ? (const void*)pt_wave_kernel<false, 0, false, 3> : trav == 1 ? (const void*)pt_wave_kernel<false, 1, false, 3> : (const void*)pt_wave_kernel<false, 2, false, 3>);
  if (pt->wave_blocks == 0 || pt->wave_lds != lds || pt->wave_mode != pt->kernel_mode || pt->wave_kern != kern) {
    pt->wave_mode = pt->kernel_mode;
    pt->wave_kern = kern;
    int per_cu = 0, cus = 0;
    SRT_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SRT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds));
    SRT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pt->device));
    if (per_cu < 1) return srt::fail(SRT_ERR_UNSUPPORTED, "wave kernel does not fit on a CU (LDS %zu bytes)", lds);
    pt->wave_blocks = per_cu * cus;
    pt->wave_lds = lds;
    if (getenv("SRT_DEBUG")) fprintf(stderr, "[srt] pt_wave_kernel: %d blocks/CU x %d CUs, %zu B LDS per block\n", per_cu, cus, lds);
  }
  const uint32_t chunk = samples_per_launch(px);
  const uint32_t nlanes = (uint32_t)pt->wave_blocks * 256;
  int st;
  srt_pt::EpochBuffers* Bp = nullptr;
  if ((st = stream_buffers(pt, s, &Bp)) != SRT_OK) return st;
  srt_pt::EpochBuffers& B = *Bp;
  if ((st = ensure(&B.d_samples, &B.samples_floats, (size_t)px * chunk * 4)) != SRT_OK) return st;
  if ((st = ensure(&B.d_records, &B.records_floats, (size_t)nlanes * kRecFields * kMaxPathDepth)) != SRT_OK) return st;
  if ((st = ensure(&B.d_running, &B.running_floats, (size_t)px * 4)) != SRT_OK) return st;
  if (!B.d_queue) {
    SRT_HIP(hipMalloc(&B.d_queue, (1 + ST_COUNT_) * sizeof(unsigned long long)));
    SRT_HIP(hipMemsetAsync(B.d_queue, 0, (1 + ST_COUNT_) * sizeof(unsigned long long), s));   // (on `s`: see stream_buffers)
  }
  for (uint32_t done = 0; done < samples || (samples == 0 && done == 0); done += chunk) {
    const uint32_t n = samples - done < chunk ? samples - done : chunk;
    WaveParams P;
    P.T = T; P.seed = seed; P.sample_base = sample_base + done; P.samples = n;
    P.singles = (n >= 4 * burst) ? burst + n % burst : n % burst;   // 3..5 of a big launch's samples per pixel, else the remainder
    // small shards (<= 64 samples per resident lane, e.g. 1/4 or 1/8 of the bench image): a longer run of short items
    // at the end shortens the tail by more than the singles cost (8.9 vs 9.1 ms for a 1/8 shard)
    if (n >= 8 * burst && (uint64_t)px * n <= 64ull * nlanes) P.singles += 2 * burst;
    if (getenv("SRT_WAVE_SINGLES") && n >= 4 * burst) {   // experiments: more single-sample units at the tail
      const uint32_t want = (uint32_t)atoi(getenv("SRT_WAVE_SINGLES"));
      if (want < n) P.singles = want - (want % burst) + n % burst;
    }
    P.groups3 = (n - P.singles) / burst;
    P.units3 = px * P.groups3;
    P.total_units = px * (P.groups3 + P.singles); P.nlanes = nlanes;
    P.sample_out = B.d_samples; P.records = B.d_records;
    // one unit per lane per queue atomic: with the 512-unit grabs of the first version the last grabs decided the
    // launch time (a 1/8 image shard ran at 56 % of the full-image rate; 83 % with 64, and the full image gained 6 %)
    P.npix = px;
    P.chunk = getenv("SRT_WAVE_CHUNK") ? (uint32_t)atoi(getenv("SRT_WAVE_CHUNK")) : kChunk;
    P.flat_ready = getenv("SRT_FLAT_READY") ? (uint32_t)atoi(getenv("SRT_FLAT_READY")) : kFlatReady;
    P.flat_interior = getenv("SRT_FLAT_INTERIOR") ? (uint32_t)atoi(getenv("SRT_FLAT_INTERIOR")) : kFlatInteriorMin;
    P.queue_head = B.d_queue; P.ray_counter = pt->d_totals + C_COUNT; P.elided_counter = pt->d_totals + C_COUNT + 1; P.stamps = B.d_queue + 1;
    P.host_cancel = pt->d_host_cancel; P.dev_cancel = B.d_cancel;
    P.watch = (pt->wave_blocks >= 2 && !getenv("SRT_NO_CANCEL_WATCH")) ? 1u : 0u;   // the last workgroup watches for srt_pt_cancel (pt_wave.h; the variable: A/B runs)
    if (n) {
      SRT_HIP(hipMemsetAsync(B.d_queue, 0, sizeof(unsigned long long), s));
      const DScene DS = device_scene(pt, &B);
#define SRT_LAUNCH_WAVE(STAMP_, TRAV_, DL_, NR_)                                                                              \
  pt_wave_kernel<STAMP_, TRAV_, DL_, NR_><<<dim3(pt->wave_blocks), dim3(256), lds, s>>>(DS, P, DS.objects, DS.tris, DS.tri_nrm,       \
                                                                                   DS.nodes, DS.lights, DS.light_tris,          \
                                                                                   DS.materials, DS.wave_tlas, DS.blas_recs,    \
                                                                                   P.records, P.sample_out)
      if ((st = time_begin(pt, s)) != SRT_OK) return st;
      if (two) { if (trav == 0) SRT_LAUNCH_WAVE(false, 0, false, 2); else SRT_LAUNCH_WAVE(false, 1, false, 2); }
      else if (stamp) { if (trav == 0) SRT_LAUNCH_WAVE(true, 0, false, 3); else if (trav == 1) SRT_LAUNCH_WAVE(true, 1, false, 3); else SRT_LAUNCH_WAVE(true, 2, false, 3); }
      else if (dl) { if (trav == 0) SRT_LAUNCH_WAVE(false, 0, true, 3); else SRT_LAUNCH_WAVE(false, 1, true, 3); }
      else { if (trav == 0) SRT_LAUNCH_WAVE(false, 0, false, 3); else if (trav == 1) SRT_LAUNCH_WAVE(false, 1, false, 3); else SRT_LAUNCH_WAVE(false, 2, false, 3); }
#undef SRT_LAUNCH_WAVE
      SRT_HIP(hipGetLastError());
      if ((st = time_end(pt, s)) != SRT_OK) return st;
    }
    const int first = done == 0, last = done + chunk >= samples;
    if (d_tiles_out) pt_reduce_kernel<<<dim3((px + 255) / 256), dim3(256), 0, s>>>(T, pt->w, pt->h, n, B.d_samples, B.d_running, first, last, d_tiles_out, B.d_cancel);
    B.last_samples = n; B.last_npix = px;
    SRT_HIP(hipGetLastError());
    if (samples == 0) break;
  }
  return SRT_OK;
}

// One epoch in the streamed form (pt_stream.h): per launch of <= 64 samples per pixel a fixed number of generations, each
// one logic kernel (pt_wave_kernel<.., TRAV = 3, ..>: consume hits, shade, refill, emit rays) and one ray-cast kernel;
// then the ordered per-pixel reduction.  Nothing here waits for the device.
int render_epoch_stream(srt_pt* pt, hipStream_t s, uint64_t seed, uint32_t sample_base, uint32_t samples, float* d_tiles_out) {
  const TileMap& T = pt->tiles;
  const uint32_t px = T.local_tiles * T.tile_w * T.tile_h;
  const FlatScene& F = pt->built.flat;
  const int trav = wave_trav(pt);                         // 3: every ray through the ray-cast kernel, 4: sweeps in the logic kernel, walks queued
  const bool dl = !F.delta_lights.empty() || pt->env_type != 0;
  const bool two = elision_provable(pt);
  const uint32_t burst = two ? 2u : kBurst;
  const uint32_t nslots = trav == 4 ? (uint32_t)F.lazy_objects.size() * burst : burst;   // queue slots per path slot
  int st;
  // the cast kernel's launch shape: frames per lane from the scene's tree depths, as many waves per CU as the LDS holds
  const uint32_t depth = (trav == 4 ? 0u : F.max_tlas_depth) + F.max_blas_depth + 1u;
  const void* ckern = trav == 4 ? (const void*)pt_cast_kernel<false, true> : (const void*)pt_cast_kernel<false, false>;
  const void* ckern_stats = trav == 4 ? (const void*)pt_cast_kernel<true, true> : (const void*)pt_cast_kernel<true, false>;
  if (pt->cast_blocks == 0 || pt->cast_depth != depth || pt->cast_kern != ckern) {
    pt->cast_kern = ckern;
    int cus = 0;
    SRT_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pt->device));
    // 13 frames of 12 bytes x 64 lanes: 16 waves per CU fit into the 160 KB (what the general build's ~100 VGPRs allow); the
    // walk-only build needs 85 VGPRs - five waves per SIMD - and takes 10 frames for 20 waves per CU (cast 138 -> 132 ms)
    const uint32_t lds_max = getenv("SRT_CAST_LDS_FRAMES") ? (uint32_t)atoi(getenv("SRT_CAST_LDS_FRAMES")) : (trav == 4 ? 10u : 13u);
    const uint32_t lds_frames = depth < lds_max ? depth : (lds_max < 1u ? 1u : lds_max);
    const size_t per_wave = (size_t)lds_frames * 3u * 64u * sizeof(uint32_t);
    int best_waves = 0;
    pt->cast_blocks = 0;
    for (int w : {4, 2, 1}) {
      const size_t lds = per_wave * (size_t)w;
      if (lds > 160u * 1024u) continue;
      if (hipFuncSetAttribute(ckern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) continue;
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ckern, 64 * w, lds) != hipSuccess || per_cu < 1) continue;
      if (per_cu * w > best_waves) { best_waves = per_cu * w; pt->cast_threads = 64 * w; pt->cast_blocks = per_cu * cus; pt->cast_lds = lds; }
    }
    if (best_waves == 0) return srt::fail(SRT_ERR_UNSUPPORTED, "the ray-cast kernel's traversal stack (%u frames per lane) does not fit into LDS", depth);
    pt->cast_depth = depth; pt->cast_lds_frames = lds_frames;
    SRT_HIP(hipFuncSetAttribute(ckern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pt->cast_lds));
    if (getenv("SRT_CAST_STATS") && !pt->d_cast_stats) {
      SRT_HIP(hipMalloc(&pt->d_cast_stats, CS_COUNT * sizeof(unsigned long long)));
      SRT_HIP(hipMemsetAsync(pt->d_cast_stats, 0, CS_COUNT * sizeof(unsigned long long), s));
    }
    if (pt->d_cast_stats) SRT_HIP(hipFuncSetAttribute(ckern_stats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pt->cast_lds));
    if (getenv("SRT_DEBUG")) fprintf(stderr, "[srt] pt_cast_kernel: %d blocks x %d threads, %zu B LDS per block (%u of %u frames per lane), %d waves/CU\n",
                                     pt->cast_blocks, pt->cast_threads, pt->cast_lds, lds_frames, depth, best_waves);
  }
  // the logic kernel: 1024-thread blocks; the streamed sweeps keep their per-wave slots in LDS (16 waves)
  const size_t nq = (trav == 4 && F.use_bvh) ? F.wave_tlas.size() : 0;
  const size_t lds_per_wave = (size_t)(nq > 0 ? nq - 1 : 0) * (2 * burst) * 64 * sizeof(float);
  const uint32_t lthreads = kStreamBlock;
  const size_t logic_lds = (size_t)(lthreads / 64u) * lds_per_wave;
  // the streamed sweeps without delta / environment lights run a generation's two passes as two kernels (pt_wave.h, PHASE):
  // resolve (sweep slots in LDS) and probe (no LDS); SRT_STREAM_FUSED=1 (diagnostic) keeps them in one kernel
  const bool split = trav == 4 && !dl && !getenv("SRT_STREAM_FUSED");
  const void* lkern = trav == 4 ? (two ? (split ? (const void*)pt_wave_kernel<false, 4, false, 2, 1> : (const void*)pt_wave_kernel<false, 4, false, 2>) : dl ? (const void*)pt_wave_kernel<false, 4, true, 3>
                                           : (split ? (const void*)pt_wave_kernel<false, 4, false, 3, 1> : (const void*)pt_wave_kernel<false, 4, false, 3>))
                                : (two ? (const void*)pt_wave_kernel<false, 3, false, 2> : dl ? (const void*)pt_wave_kernel<false, 3, true, 3> : (const void*)pt_wave_kernel<false, 3, false, 3>);
  if (logic_lds > 160u * 1024u) return srt::fail(SRT_ERR_UNSUPPORTED, "the streamed sweeps' LDS slots (%zu bytes) do not fit", logic_lds);
  SRT_HIP(hipFuncSetAttribute(lkern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)logic_lds));
  const uint32_t chunk = samples_per_launch(px);
  srt_pt::EpochBuffers* Bp = nullptr;
  if ((st = stream_buffers(pt, s, &Bp)) != SRT_OK) return st;
  srt_pt::EpochBuffers& B = *Bp;
  if ((st = ensure(&B.d_samples, &B.samples_floats, (size_t)px * chunk * 4)) != SRT_OK) return st;
  if ((st = ensure(&B.d_running, &B.running_floats, (size_t)px * 4)) != SRT_OK) return st;
  if (!B.d_sc) SRT_HIP(hipMalloc(&B.d_sc, sizeof(StreamCounters)));
  const size_t spill_words = (size_t)(depth - pt->cast_lds_frames) * 3u * (size_t)pt->cast_blocks * (size_t)pt->cast_threads;
  if (spill_words && (st = ensure(&B.d_cast_spill, &B.cast_spill_words, spill_words)) != SRT_OK) return st;
  const uint32_t shadow_batches = (uint32_t)((F.delta_lights.size() + 2) / 3);
  const DScene DS = device_scene(pt, &B);
  for (uint32_t done = 0; done < samples || (samples == 0 && done == 0); done += chunk) {
    const uint32_t n = samples - done < chunk ? samples - done : chunk;
    WaveParams P{};
    P.T = T; P.seed = seed; P.sample_base = sample_base + done; P.samples = n;
    P.singles = (n >= 4 * burst) ? burst + n % burst : n % burst;   // as the persistent kernel: a few one-sample units at the end
    P.groups3 = (n - P.singles) / burst;
    P.units3 = px * P.groups3;
    P.total_units = px * (P.groups3 + P.singles);
    // path slots: every unit its own while they are few, else a fixed population that is refilled from the unit queue
    // (default 3 Mi slots: with the logic split in two kernels the ray-cast kernel's gain from a larger population - fewer, fuller
    //  generations: 133 -> 119 ms per epoch of BASELINE configs[4]'s stand-in - outweighs what the logic kernels lose once their state
    //  outgrows the Infinity Cache; 2 Mi: 188.9, 3 Mi: 183.1, 4 Mi: 184.7, 8 Mi: 201.9 ms per epoch)
    uint64_t want_slots = pt->stream_slots ? pt->stream_slots : (3u << 20);
    if (!pt->stream_slots && getenv("SRT_STREAM_SLOTS")) {    // diagnostic override; anything outside [1, kMaxStreamSlots] is ignored
      const long long v = atoll(getenv("SRT_STREAM_SLOTS"));
      if (v >= 1 && v <= (long long)kMaxStreamSlots) want_slots = (uint64_t)v;
    }
    if (want_slots < lthreads) want_slots = lthreads;
    if (want_slots > kMaxStreamSlots) want_slots = kMaxStreamSlots;
    const uint32_t nlanes = (uint32_t)std::min<uint64_t>(((uint64_t)P.total_units + lthreads - 1) / lthreads * lthreads,
                                                         (want_slots + lthreads - 1) / lthreads * lthreads);
    if (n && !nlanes) return srt::fail(SRT_ERR_STATE, "streamed launch with %u samples but no path slots (units %u)", n, P.total_units);
    if (n && nlanes) {
      P.nlanes = nlanes;
      const uint32_t nblocks = nlanes / lthreads;
      if ((st = ensure(&B.d_records, &B.records_floats, (size_t)nlanes * kRecFields * kMaxPathDepth)) != SRT_OK) return st;
      if ((st = ensure(&B.d_state, &B.state_words, (size_t)nlanes * SW_DL_WORDS)) != SRT_OK) return st;
      if ((st = ensure(&B.d_ray_o, &B.ray_o_n, 2 * (size_t)nlanes * nslots)) != SRT_OK) return st;   // {origin, direction} per request
      if ((st = ensure(&B.d_ray_id, &B.ray_id_n, (size_t)nlanes * nslots)) != SRT_OK) return st;
      if ((st = ensure(&B.d_hits, &B.hits_n, (size_t)nlanes * nslots)) != SRT_OK) return st;
      if ((st = ensure(&B.d_alive_list, &B.alive_list_n, (size_t)nlanes)) != SRT_OK) return st;
      if (B.block_counters_n < 2 * (size_t)nblocks) {
        if ((st = ensure(&B.d_block_counters, &B.block_counters_n, 2 * (size_t)nblocks)) != SRT_OK) return st;
        SRT_HIP(hipMemsetAsync(B.d_block_counters, 0, 2 * (size_t)nblocks * sizeof(unsigned long long), s));
      }
      P.sample_out = B.d_samples; P.records = B.d_records; P.npix = px; P.chunk = kChunk;
      P.queue_head = &B.d_sc->queue_head; P.ray_counter = pt->d_totals + C_COUNT; P.elided_counter = pt->d_totals + C_COUNT + 1;
      P.stamps = nullptr; P.flat_ready = 0; P.flat_interior = 0;
      P.state = B.d_state; P.ray_o = B.d_ray_o; P.ray_d = B.d_ray_o + 1; P.hits = B.d_hits; P.sc = B.d_sc;
      P.block_counters = B.d_block_counters;
      P.obj_shift = stream_obj_shift(F);
      P.host_cancel = pt->d_host_cancel; P.dev_cancel = B.d_cancel;
      P.alive_list = B.d_alive_list;
      SRT_HIP(hipMemsetAsync(B.d_sc, 0, sizeof(StreamCounters), s));
      SRT_HIP(hipMemsetAsync(B.d_state, 0, 2 * (size_t)nlanes * sizeof(uint32_t), s));   // the flags and emit planes: every slot idle
      // generations: list scheduling of units of <= M batches on nlanes slots
      const uint64_t M = 1ull + (uint64_t)burst * pt->max_depth * (1ull + (dl ? shadow_batches : 0ull));
      const uint64_t gens = ((uint64_t)P.total_units * M + nlanes - 1) / nlanes + M + 2;
      CastParams C{};
      C.ray_o = B.d_ray_o; C.ray_d = B.d_ray_o + 1; C.ray_id = B.d_ray_id; C.hits = B.d_hits; C.nlanes = nlanes;
      C.depth = depth; C.lds_frames = pt->cast_lds_frames; C.spill = B.d_cast_spill; C.obj_shift = P.obj_shift; C.sc = B.d_sc; C.total_units = P.total_units;
      C.walk_nr = trav == 4 ? burst : 0u;
      for (size_t i = 0; i < F.lazy_objects.size() && i < 4; i++) C.lazy_obj[i] = F.lazy_objects[i];
      C.fetch_min = getenv("SRT_CAST_FETCH") ? (uint32_t)atoi(getenv("SRT_CAST_FETCH")) : 16u;
      C.interior_min = getenv("SRT_CAST_INTERIOR") ? (uint32_t)atoi(getenv("SRT_CAST_INTERIOR")) : 16u;
      if (C.fetch_min < 1u) C.fetch_min = 1u;
      C.leaf_min = getenv("SRT_CAST_LEAF") ? (uint32_t)atoi(getenv("SRT_CAST_LEAF")) : 12u;
      C.object_min = getenv("SRT_CAST_OBJECT") ? (uint32_t)atoi(getenv("SRT_CAST_OBJECT")) : (trav == 4 ? 16u : 24u);
      C.own_share = getenv("SRT_CAST_OWN") ? (uint32_t)atoi(getenv("SRT_CAST_OWN")) : 128u;
      C.grab = getenv("SRT_CAST_GRAB") ? (uint32_t)atoi(getenv("SRT_CAST_GRAB")) : 32u;
      if (C.grab < 1u) C.grab = 1u;
      if (C.own_share > 256u) C.own_share = 256u;
      C.pops = getenv("SRT_CAST_POPS") ? (uint32_t)atoi(getenv("SRT_CAST_POPS")) : 2u;
      if (C.leaf_min < 1u) C.leaf_min = 1u;
      if (C.object_min < 1u) C.object_min = 1u;
      C.stats = pt->d_cast_stats;
      C.tri_packed = pt->d_tri_packed;
      C.dev_cancel = B.d_cancel;
      const dim3 lgrid(nblocks), lblock(lthreads);
      const dim3 cgrid((nlanes + kCompactChunk - 1) / kCompactChunk);
      if ((st = time_begin(pt, s)) != SRT_OK) return st;
      for (uint64_t g = 0; g < gens; g++) {
        P.gen = (uint32_t)g;
#define SRT_LAUNCH_LOGIC(TRAV_, DL_, NR_)                                                                                          \
  pt_wave_kernel<false, TRAV_, DL_, NR_><<<lgrid, lblock, logic_lds, s>>>(DS, P, DS.objects, DS.tris, DS.tri_nrm, DS.nodes,         \
                                                                          DS.lights, DS.light_tris, DS.materials, DS.wave_tlas,    \
                                                                          DS.blas_recs, P.records, P.sample_out)
#define SRT_LAUNCH_PHASE(NR_, PHASE_, LDS_)                                                                                         \
  pt_wave_kernel<false, 4, false, NR_, PHASE_><<<lgrid, lblock, LDS_, s>>>(DS, P, DS.objects, DS.tris, DS.tri_nrm, DS.nodes,        \
                                                                           DS.lights, DS.light_tris, DS.materials, DS.wave_tlas,   \
                                                                           DS.blas_recs, P.records, P.sample_out)
        if ((st = stream_time_begin(pt, s, 0)) != SRT_OK) return st;
        if (split) {
          if (two) SRT_LAUNCH_PHASE(2, 1, logic_lds); else SRT_LAUNCH_PHASE(3, 1, logic_lds);
          if ((st = stream_time_end(pt, s, 0)) != SRT_OK || (st = stream_time_begin(pt, s, 3)) != SRT_OK) return st;
          if (two) SRT_LAUNCH_PHASE(2, 2, 0); else SRT_LAUNCH_PHASE(3, 2, 0);
          if ((st = stream_time_end(pt, s, 3)) != SRT_OK || (st = stream_time_begin(pt, s, 0)) != SRT_OK) return st;   // (an empty bracket closes slot 0 below)
        }
        else if (trav == 4) { if (two) SRT_LAUNCH_LOGIC(4, false, 2); else if (dl) SRT_LAUNCH_LOGIC(4, true, 3); else SRT_LAUNCH_LOGIC(4, false, 3); }
        else { if (two) SRT_LAUNCH_LOGIC(3, false, 2); else if (dl) SRT_LAUNCH_LOGIC(3, true, 3); else SRT_LAUNCH_LOGIC(3, false, 3); }
#undef SRT_LAUNCH_LOGIC
#undef SRT_LAUNCH_PHASE
        if ((st = stream_time_end(pt, s, 0)) != SRT_OK || (st = stream_time_begin(pt, s, 1)) != SRT_OK) return st;
        pt_compact_kernel<<<cgrid, dim3(1024), 0, s>>>(B.d_state + (size_t)SW_EMIT * nlanes, nlanes, nslots, B.d_sc, (uint32_t)g, B.d_ray_id, pt->d_totals + C_COUNT + 2, pt->d_host_cancel, B.d_cancel, B.d_alive_list);
        if ((st = stream_time_end(pt, s, 1)) != SRT_OK || (st = stream_time_begin(pt, s, 2)) != SRT_OK) return st;
        C.nrays = &B.d_sc->nrays[g & 1]; C.head = &B.d_sc->cast_head[g & 1]; C.gen = (uint32_t)g;
        const dim3 kgrid(pt->cast_blocks), kblock(pt->cast_threads);
        if (trav == 4) { if (pt->d_cast_stats) pt_cast_kernel<true, true><<<kgrid, kblock, pt->cast_lds, s>>>(DS, C); else pt_cast_kernel<false, true><<<kgrid, kblock, pt->cast_lds, s>>>(DS, C); }
        else { if (pt->d_cast_stats) pt_cast_kernel<true, false><<<kgrid, kblock, pt->cast_lds, s>>>(DS, C); else pt_cast_kernel<false, false><<<kgrid, kblock, pt->cast_lds, s>>>(DS, C); }
        if ((st = stream_time_end(pt, s, 2)) != SRT_OK) return st;
      }
      if (pt->stream_timing) pt->stream_generations += gens;
      pt_stream_finish_kernel<<<dim3(1), dim3(256), 0, s>>>(B.d_block_counters, nblocks, pt->d_totals + C_COUNT, B.d_sc, pt->d_fault, B.d_cancel);
      SRT_HIP(hipGetLastError());
      if ((st = time_end(pt, s)) != SRT_OK) return st;
    }
    const int first = done == 0, last = done + chunk >= samples;
    if (d_tiles_out) pt_reduce_kernel<<<dim3((px + 255) / 256), dim3(256), 0, s>>>(T, pt->w, pt->h, n, B.d_samples, B.d_running, first, last, d_tiles_out, B.d_cancel);
    B.last_samples = n; B.last_npix = px;
    SRT_HIP(hipGetLastError());
    if (samples == 0) break;
  }
  return SRT_OK;
}

// One epoch with one lane per sample (general scenes): launches of <= 64 samples per pixel + ordered reduction.
int render_epoch_units(srt_pt* pt, hipStream_t s, uint64_t seed, uint32_t sample_base, uint32_t samples, float* d_tiles_out) {
  const TileMap& T = pt->tiles;
  const uint32_t px = T.local_tiles * T.tile_w * T.tile_h;
  const uint32_t chunk = samples_per_launch(px);
  int st;
  srt_pt::EpochBuffers* Bp = nullptr;
  if ((st = stream_buffers(pt, s, &Bp)) != SRT_OK) return st;
  srt_pt::EpochBuffers& B = *Bp;
  if ((st = ensure(&B.d_samples, &B.samples_floats, (size_t)px * chunk * 4)) != SRT_OK) return st;
  if ((st = ensure(&B.d_running, &B.running_floats, (size_t)px * 4)) != SRT_OK) return st;
  for (uint32_t done = 0; done < samples || (samples == 0 && done == 0); done += chunk) {
    const uint32_t n = samples - done < chunk ? samples - done : chunk;
    const uint64_t units = (uint64_t)px * n;
    if (n) {
      if ((st = time_begin(pt, s)) != SRT_OK) return st;
      pt_unit_kernel<<<dim3((unsigned)((units + 63) / 64)), dim3(64), 0, s>>>(device_scene(pt, &B), T, seed, sample_base + done, n,
                                                                              (uint32_t)units, B.d_samples, pt->d_totals + C_COUNT, pt->d_host_cancel, B.d_cancel);
      SRT_HIP(hipGetLastError());
      if ((st = time_end(pt, s)) != SRT_OK) return st;
    }
    const int first = done == 0, last = done + chunk >= samples;
    if (d_tiles_out) pt_reduce_kernel<<<dim3((px + 255) / 256), dim3(256), 0, s>>>(T, pt->w, pt->h, n, B.d_samples, B.d_running, first, last, d_tiles_out, B.d_cancel);
    B.last_samples = n; B.last_npix = px;
    SRT_HIP(hipGetLastError());
    if (samples == 0) break;
  }
  return SRT_OK;
}

}  // namespace

namespace srt {
int pt_check_fault(srt_pt* pt, const char* what) { return pt ? check_stream_fault(pt, what) : SRT_OK; }
}  // namespace srt

extern "C" {

int srt_pt_create(int device, srt_pt** out) {
  if (!out) return srt::fail(SRT_ERR_INVALID, "srt_pt_create: out is NULL");
  *out = nullptr;
  srt_pt* pt = new (std::nothrow) srt_pt();
  if (!pt) return srt::fail(SRT_ERR_INVALID, "out of host memory");
  if (device >= 0) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
      delete pt;
      return srt::fail(SRT_ERR_NO_DEVICE, "no HIP device available (%s); this path has no CPU fallback",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device >= count) { delete pt; return srt::fail(SRT_ERR_INVALID, "device %d out of range [0,%d)", device, count); }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&pt->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&pt->d_totals, (C_COUNT + 4) * sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc((void**)&pt->h_fault, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&pt->d_fault, pt->h_fault, 0) != hipSuccess ||
        hipHostMalloc((void**)&pt->h_cancel, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&pt->d_host_cancel, pt->h_cancel, 0) != hipSuccess ||
        hipMemset(pt->d_totals, 0, (C_COUNT + 4) * sizeof(unsigned long long)) != hipSuccess) {
      delete pt;
      return srt::fail(SRT_ERR_HIP, "HIP context setup failed on device %d", device);
    }
    *pt->h_fault = 0u;
    *pt->h_cancel = 0u;
    (void)hipDeviceSynchronize();                         // (the null-stream memset above: done before any non-blocking stream's first kernel)
    pt->device = device;
  }
  *out = pt;
  return SRT_OK;
}

int srt_pt_destroy(srt_pt* pt) {
  if (!pt) return SRT_OK;
  if (pt->device >= 0) {
    (void)hipSetDevice(pt->device);
    (void)hipStreamSynchronize(pt->stream);
    if (pt->d_cast_stats) {                               // SRT_CAST_STATS=1 (diagnostic): the sums, on stderr
      unsigned long long h[CS_COUNT];
      static const char* names[CS_COUNT] = {"outer", "fetch", "interior_trips", "interior_lanes", "leaf_trips", "leaf_lanes", "leaf_tris",
                                            "object_trips", "object_lanes", "cycles_fetch", "cycles_interior", "walking_lanes", "cycles_leaf", "cycles_object"};
      if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, pt->d_cast_stats, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
        for (int i = 0; i < CS_COUNT; i++) fprintf(stderr, "[srt] cast %s %llu\n", names[i], h[i]);
      (void)hipFree(pt->d_cast_stats);
    }
    (void)hipFree(pt->d_nodes); (void)hipFree(pt->d_tris); (void)hipFree(pt->d_tri_packed); (void)hipFree(pt->d_nrm); (void)hipFree(pt->d_objects);
    (void)hipFree(pt->d_lights); (void)hipFree(pt->d_ltris); (void)hipFree(pt->d_mats); (void)hipFree(pt->d_wave); (void)hipFree(pt->d_blas); (void)hipFree(pt->d_wave_lazy); (void)hipFree(pt->d_dlights); (void)hipFree(pt->d_env_map);
    (void)hipFree(pt->d_tile_buf); (void)hipFree(pt->d_image); (void)hipFree(pt->d_totals);
    if (pt->h_fault) (void)hipHostFree(pt->h_fault);
    if (pt->h_cancel) (void)hipHostFree(pt->h_cancel);
    for (auto& kv : pt->epoch_buffers) {
      (void)hipFree(kv.second.d_cancel); (void)hipFree(kv.second.d_ray_log); (void)hipFree(kv.second.d_alive_list);
      (void)hipFree(kv.second.d_samples); (void)hipFree(kv.second.d_records); (void)hipFree(kv.second.d_running); (void)hipFree(kv.second.d_queue);
      (void)hipFree(kv.second.d_state); (void)hipFree(kv.second.d_ray_o); (void)hipFree(kv.second.d_ray_d); (void)hipFree(kv.second.d_ray_id); (void)hipFree(kv.second.d_cast_spill);
      (void)hipFree(kv.second.d_hits); (void)hipFree(kv.second.d_sc); (void)hipFree(kv.second.d_block_counters);
    }
    for (auto& v : {&pt->timed, &pt->spare, &pt->stream_timed[0], &pt->stream_timed[1], &pt->stream_timed[2]})
      for (auto& ev : *v) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    (void)hipStreamDestroy(pt->stream);
  }
  delete pt;
  return SRT_OK;
}

int srt_pt_scene_begin(srt_pt* pt) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_scene_begin: NULL context");
  pt->inputs.clear();
  pt->materials.clear();
  pt->delta_lights.clear();
  pt->env_type = 0;
  pt->env_map.clear(); pt->env_w = pt->env_h = 0;
  pt->committed = false;
  return SRT_OK;
}

int srt_pt_add_material(srt_pt* pt, const srt_pt_material* m, uint32_t* index_out) {
  if (!pt || !m) return srt::fail(SRT_ERR_INVALID, "srt_pt_add_material: NULL argument");
  if (m->type > SRT_MAT_REFRACT) return srt::fail(SRT_ERR_INVALID, "unknown material type %u", m->type);
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  Material mm;
  mm.type = m->type;
  for (int i = 0; i < 3; i++) { mm.a[i] = m->a[i]; mm.b[i] = m->b[i]; }
  mm.ior = m->ior;
  pt->materials.push_back(mm);
  if (index_out) *index_out = (uint32_t)pt->materials.size() - 1;
  return SRT_OK;
}

int srt_pt_add_mesh(srt_pt* pt, const float* positions, const float* normals, uint32_t nverts, const uint32_t* indices,
                    uint32_t nindices, const float trans[16], uint32_t material, int is_area_light) {
  if (!pt || !positions || !normals || !indices || !trans) return srt::fail(SRT_ERR_INVALID, "srt_pt_add_mesh: NULL argument");
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  if (nverts == 0 || nindices == 0 || nindices % 3) return srt::fail(SRT_ERR_INVALID, "mesh needs >= 1 triangle (nindices %u)", nindices);
  if (material >= pt->materials.size()) return srt::fail(SRT_ERR_INVALID, "material %u not defined", material);
  for (uint32_t i = 0; i < nindices; i++)
    if (indices[i] >= nverts) return srt::fail(SRT_ERR_INVALID, "index %u (= %u) out of range (nverts %u)", i, indices[i], nverts);
  ObjectInput o;
  o.kind = OBJ_MESH;
  std::memcpy(&o.trans, trans, sizeof(Mat4));
  o.material = material;
  o.is_light = is_area_light != 0;
  o.mesh.pos.assign(positions, positions + 3 * (size_t)nverts);
  o.mesh.nrm.assign(normals, normals + 3 * (size_t)nverts);
  o.mesh.idx.assign(indices, indices + nindices);
  pt->inputs.push_back(std::move(o));
  return SRT_OK;
}

int srt_pt_add_sphere(srt_pt* pt, float radius, const float trans[16], uint32_t material) {
  if (!pt || !trans) return srt::fail(SRT_ERR_INVALID, "srt_pt_add_sphere: NULL argument");
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  if (material >= pt->materials.size()) return srt::fail(SRT_ERR_INVALID, "material %u not defined", material);
  ObjectInput o;
  o.kind = OBJ_SPHERE;
  std::memcpy(&o.trans, trans, sizeof(Mat4));
  o.material = material;
  o.radius = radius;
  pt->inputs.push_back(std::move(o));
  return SRT_OK;
}

int srt_pt_add_sphere_light(srt_pt* pt, float radius, const float trans[16], uint32_t material, const float* positions,
                            const float* normals, uint32_t nverts, const uint32_t* indices, uint32_t nindices) {
  if (!pt || !trans || !positions || !normals || !indices) return srt::fail(SRT_ERR_INVALID, "srt_pt_add_sphere_light: NULL argument");
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  if (material >= pt->materials.size()) return srt::fail(SRT_ERR_INVALID, "material %u not defined", material);
  if (!nverts || !nindices || nindices % 3) return srt::fail(SRT_ERR_INVALID, "the light mesh needs triangles (%u vertices, %u indices)", nverts, nindices);
  ObjectInput o;
  o.kind = OBJ_SPHERE;
  std::memcpy(&o.trans, trans, sizeof(Mat4));
  o.material = material;
  o.radius = radius;
  o.is_light = true;
  o.mesh.pos.assign(positions, positions + 3 * (size_t)nverts);
  o.mesh.nrm.assign(normals, normals + 3 * (size_t)nverts);
  o.mesh.idx.assign(indices, indices + nindices);
  pt->inputs.push_back(std::move(o));
  return SRT_OK;
}

int srt_pt_add_light(srt_pt* pt, uint32_t type, const float radiance[3], const float angle_bounds[2], const float trans[16]) {
  if (!pt || !radiance || !trans) return srt::fail(SRT_ERR_INVALID, "srt_pt_add_light: NULL argument");
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  if (type > SRT_LIGHT_SPOT) return srt::fail(SRT_ERR_INVALID, "unknown light type %u", type);
  if (type == SRT_LIGHT_SPOT && !angle_bounds) return srt::fail(SRT_ERR_INVALID, "a spot light needs angle_bounds");
  Mat4 T;
  std::memcpy(&T, trans, sizeof(Mat4));
  pt->delta_lights.push_back(make_delta_light(type, radiance, angle_bounds, T));
  return SRT_OK;
}

int srt_pt_set_env_light(srt_pt* pt, uint32_t type, const float radiance[3]) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_env_light: NULL context");
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  if (type > SRT_ENV_HEMISPHERE) return srt::fail(SRT_ERR_INVALID, "unknown environment light type %u (image maps: srt_pt_set_env_map)", type);
  if (type != SRT_ENV_NONE && !radiance) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_env_light: radiance is NULL");
  pt->env_type = type;
  for (int i = 0; i < 3; i++) pt->env_radiance[i] = (type != SRT_ENV_NONE) ? radiance[i] : 0.0f;
  return SRT_OK;
}

int srt_pt_set_env_map(srt_pt* pt, uint32_t width, uint32_t height, const float* rgb) {
  if (!pt || !rgb) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_env_map: NULL argument");
  if (pt->committed) return srt::fail(SRT_ERR_STATE, "scene already committed; call srt_pt_scene_begin first");
  if (!width || !height || (uint64_t)width * height > (1ull << 28)) return srt::fail(SRT_ERR_INVALID, "environment map size %ux%u", width, height);
  pt->env_type = SRT_ENV_MAP;
  pt->env_w = width; pt->env_h = height;
  pt->env_map.assign(rgb, rgb + 3 * (size_t)width * height);
  return SRT_OK;
}

int srt_pt_scene_commit(srt_pt* pt, int use_bvh) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_scene_commit: NULL context");
  // BVH<Triangle> builds of big meshes run on the device (pt_bvh_device.hip: identical arrays); srt_pt_set_bvh_builder
  const char* be = getenv("SRT_BVH_BUILDER");
  const int bmode = be ? (strcmp(be, "host") == 0 ? 0 : 1) : pt->bvh_builder;
  if (pt->device >= 0 && bmode != 0) { SRT_HIP(hipSetDevice(pt->device)); set_device_bvh_builder(build_bvh_device, pt->bvh_device_min); }
  else set_device_bvh_builder(nullptr, 0);
  const std::string err = build_scene(pt->inputs, pt->materials, use_bvh != 0, &pt->built);
  set_device_bvh_builder(nullptr, 0);
  if (!err.empty()) return srt::fail(SRT_ERR_UNSUPPORTED, "%s", err.c_str());
  pt->built.flat.delta_lights = pt->delta_lights;
  const FlatScene& F = pt->built.flat;
  if ((int)F.max_tlas_depth > kMaxTlasDepth || (int)F.max_blas_depth > kMaxBlasDepth)
    return srt::fail(SRT_ERR_UNSUPPORTED, "BVH too deep for the traversal stacks (TLAS %u > %d or BLAS %u > %d)",
                     F.max_tlas_depth, kMaxTlasDepth, F.max_blas_depth, kMaxBlasDepth);
  if (pt->device >= 0) {
    SRT_HIP(hipSetDevice(pt->device));
    SRT_HIP(hipStreamSynchronize(pt->stream));
    int st;
    if ((st = upload(&pt->d_nodes, F.nodes)) || (st = upload(&pt->d_tris, F.tris)) || (st = upload(&pt->d_nrm, F.tri_nrm)) ||
        (st = upload(&pt->d_tri_packed, F.tri_packed)) ||
        (st = upload(&pt->d_objects, F.objects)) || (st = upload(&pt->d_lights, F.lights)) ||
        (st = upload(&pt->d_ltris, F.light_tris)) || (st = upload(&pt->d_mats, F.materials)) ||
        (st = upload(&pt->d_wave, F.wave_tlas)) || (st = upload(&pt->d_blas, F.blas_recs)) || (st = upload(&pt->d_wave_lazy, F.wave_lazy)) ||
        (st = upload(&pt->d_dlights, F.delta_lights)) || (st = upload(&pt->d_env_map, pt->env_map)))
      return st;
  }
  pt->committed = true;
  pt->cast_blocks = 0;                                    // the ray-cast kernel's stack depth follows the scene
  return SRT_OK;
}

int srt_pt_set_stream_slots(srt_pt* pt, uint32_t slots) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_stream_slots: NULL context");
  if (slots > kMaxStreamSlots) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_stream_slots: at most %u path slots (got %u)", kMaxStreamSlots, slots);
  pt->stream_slots = slots;
  return SRT_OK;
}

int srt_pt_set_bvh_builder(srt_pt* pt, int device, uint32_t min_primitives) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_bvh_builder: NULL context");
  pt->bvh_builder = device ? 1 : 0;
  pt->bvh_device_min = min_primitives;
  return SRT_OK;
}

int srt_pt_set_camera(srt_pt* pt, const float iview[16], float vert_fov_deg, float aspect_ratio) {
  if (!pt || !iview) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_camera: NULL argument");
  pt->cam = make_camera(iview, vert_fov_deg, aspect_ratio);
  pt->have_cam = true;
  return SRT_OK;
}

int srt_pt_set_params(srt_pt* pt, uint32_t width, uint32_t height, uint32_t max_depth) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_params: NULL context");
  if (!width || !height) return srt::fail(SRT_ERR_INVALID, "image must be at least 1x1 (got %ux%u)", width, height);
  if ((uint64_t)width * height > 0x7fffffffull) return srt::fail(SRT_ERR_UNSUPPORTED, "image larger than 2^31 pixels");
  if (width > 65535u || height > 65535u) return srt::fail(SRT_ERR_UNSUPPORTED, "image sides above 65535 are not supported (got %ux%u)", width, height);
  if (max_depth > (uint32_t)kMaxPathDepth) return srt::fail(SRT_ERR_UNSUPPORTED, "max_depth %u > %d is not supported", max_depth, kMaxPathDepth);
  pt->w = width; pt->h = height; pt->max_depth = max_depth;
  update_tiling(pt);
  return SRT_OK;
}

int srt_pt_set_tiling(srt_pt* pt, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_tiling: NULL context");
  if (!tile_w || !tile_h || tile_w % 8 || tile_h % 8) return srt::fail(SRT_ERR_INVALID, "tile size must be a positive multiple of 8 (got %ux%u)", tile_w, tile_h);
  if (!world || rank >= world) return srt::fail(SRT_ERR_INVALID, "rank %u / world %u is not a valid shard", rank, world);
  pt->tiles.tile_w = tile_w; pt->tiles.tile_h = tile_h; pt->tiles.rank = rank; pt->tiles.world = world;
  update_tiling(pt);
  return SRT_OK;
}

int srt_pt_tile_info(srt_pt* pt, uint32_t* local_tiles, uint32_t* tiles_per_rank, uint32_t* floats_per_tile) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_tile_info: NULL context");
  if (!pt->w) return srt::fail(SRT_ERR_STATE, "srt_pt_tile_info before srt_pt_set_params");
  if (local_tiles) *local_tiles = pt->tiles.local_tiles;
  if (tiles_per_rank) *tiles_per_rank = pt->tiles_per_rank;
  if (floats_per_tile) *floats_per_tile = pt->tiles.tile_w * pt->tiles.tile_h * 3;
  return SRT_OK;
}

int srt_pt_set_kernel(srt_pt* pt, int mode) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_kernel: NULL context");
  if (mode < 0 || mode > 7)
    return srt::fail(SRT_ERR_INVALID, "kernel mode must be 0 (auto), 1 (per-lane, lane per pixel), 2 (wave-uniform), 3 (wave-uniform, stamped), 4 (per-lane, lane per sample), 5 (persistent waves, flattened per-lane walk), 6 (streamed: logic + ray-cast kernels) or 7 (streamed sweeps: BVH<Triangle> walks queued)");
  pt->kernel_mode = mode;
  return SRT_OK;
}

}  // extern "C"

namespace {
// srt_pt_render_epoch_device (d_tiles_out: the epoch's tile radiance) and srt_pt_render_samples_device (d_tiles_out == NULL: the
// launch's per-sample radiance stays in the stream's sample buffer for srt_pt_fold_epochs_device).
int render_device(srt_pt* pt, const char* what, void* stream, uint64_t seed, uint32_t sample_base, uint32_t samples, float* d_tiles_out) {
  int st = need_ready(pt, what);
  if (st != SRT_OK) return st;
  if (__atomic_load_n(pt->h_cancel, __ATOMIC_ACQUIRE) != 0u) return SRT_CANCELLED;   // nothing more is enqueued until srt_pt_clear_cancel
  hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream; NULL is the HIP default stream
  const TileMap& T = pt->tiles;
  const uint64_t lanes = (uint64_t)T.local_tiles * T.tile_w * T.tile_h;
  if (lanes * 64 > 0xffffffffull) return srt::fail(SRT_ERR_UNSUPPORTED, "more than 2^32 sample units per launch");
  if (lanes) {
    if ((pt->kernel_mode == 2 || pt->kernel_mode == 3) && !wave_kernel_applies(pt))
      return srt::fail(SRT_ERR_UNSUPPORTED, "wave-uniform kernel needs 1..%u objects (scene has %zu)", kWaveMaxObjects,
                       pt->built.flat.objects.size());
    if (pt->kernel_mode == 5 && !wave_kernel_applies(pt))
      return srt::fail(SRT_ERR_UNSUPPORTED, "the flattened-walk kernel needs 1..31 objects (scene has %zu)", pt->built.flat.objects.size());
    if (pt->kernel_mode == 7 && !wave_kernel_applies(pt))
      return srt::fail(SRT_ERR_UNSUPPORTED, "the streamed sweeps need 1..%u objects, of which 1..%u meshes with a real BVH<Triangle> (scene has %zu objects, %zu such meshes)",
                       kWaveMaxObjects, kMaxLazy, pt->built.flat.objects.size(), pt->built.flat.lazy_objects.size());
    if (pt->kernel_mode == 6 && !wave_kernel_applies(pt))
      return srt::fail(SRT_ERR_UNSUPPORTED, "the streamed form cannot pack this scene's hits (%zu objects, %zu triangles)", pt->built.flat.objects.size(),
                       pt->built.flat.tris.size());
    if (wave_kernel_applies(pt)) {
      st = render_epoch_wave(pt, s, seed, sample_base, samples, d_tiles_out);
      if (st != SRT_OK) return st;
    } else if (pt->kernel_mode != 1) {
      st = render_epoch_units(pt, s, seed, sample_base, samples, d_tiles_out);
      if (st != SRT_OK) return st;
    } else {
      const uint32_t blocks = (uint32_t)((lanes + 63) / 64);
      srt_pt::EpochBuffers* Bp = nullptr;
      if ((st = stream_buffers(pt, s, &Bp)) != SRT_OK) return st;
      if ((st = time_begin(pt, s)) != SRT_OK) return st;
      pt_epoch_kernel<<<dim3(blocks), dim3(64), 0, s>>>(device_scene(pt, Bp), T, seed, sample_base, samples, d_tiles_out,
                                                        pt->d_totals + C_COUNT, pt->d_host_cancel, Bp->d_cancel);
      SRT_HIP(hipGetLastError());
      if ((st = time_end(pt, s)) != SRT_OK) return st;
    }
    // camera samples of this epoch: pixels of this rank's tiles that lie inside the image
    uint64_t px = 0;
    for (uint32_t k = 0; k < T.local_tiles; k++) {
      const uint32_t tile = T.rank + k * T.world;
      const uint32_t x0 = (tile % T.tiles_x) * T.tile_w, y0 = (tile / T.tiles_x) * T.tile_h;
      px += (uint64_t)std::min(T.tile_w, pt->w - x0) * std::min(T.tile_h, pt->h - y0);
    }
    pt->camera_samples += px * samples;
  }
  return SRT_OK;
}
}  // namespace

extern "C" {

int srt_pt_render_epoch_device(srt_pt* pt, void* stream, uint64_t seed, uint32_t sample_base, uint32_t samples,
                               float* d_tiles_out) {
  if (!d_tiles_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_render_epoch_device: output is NULL");
  return render_device(pt, "srt_pt_render_epoch_device", stream, seed, sample_base, samples, d_tiles_out);
}

// ---- launches decoupled from the reference's epochs -------------------------------------------------------------------
int srt_pt_max_samples_per_launch(srt_pt* pt, uint32_t* samples) {
  if (!pt || !samples) return srt::fail(SRT_ERR_INVALID, "srt_pt_max_samples_per_launch: NULL argument");
  if (!pt->w) return srt::fail(SRT_ERR_STATE, "srt_pt_max_samples_per_launch before srt_pt_set_params");
  *samples = samples_per_launch(pt->tiles.local_tiles * pt->tiles.tile_w * pt->tiles.tile_h);
  return SRT_OK;
}

int srt_pt_accumulator_floats(srt_pt* pt, size_t* nfloats) {
  if (!pt || !nfloats) return srt::fail(SRT_ERR_INVALID, "srt_pt_accumulator_floats: NULL argument");
  if (!pt->w) return srt::fail(SRT_ERR_STATE, "srt_pt_accumulator_floats before srt_pt_set_params");
  const size_t px = (size_t)pt->tiles.local_tiles * pt->tiles.tile_w * pt->tiles.tile_h;
  *nfloats = 8 * (px ? px : 1);
  return SRT_OK;
}

int srt_pt_render_samples_device(srt_pt* pt, void* stream, uint64_t seed, uint32_t sample_base, uint32_t samples) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_render_samples_device: NULL context");
  if (pt->kernel_mode == 1) return srt::fail(SRT_ERR_UNSUPPORTED, "srt_pt_render_samples_device: the lane-per-pixel kernel (mode 1) keeps no per-sample radiance");
  if (pt->w) {
    const uint32_t most = samples_per_launch(pt->tiles.local_tiles * pt->tiles.tile_w * pt->tiles.tile_h);
    if (samples == 0 || samples > most) return srt::fail(SRT_ERR_INVALID, "srt_pt_render_samples_device: 1..%u samples per launch (got %u)", most, samples);
  }
  return render_device(pt, "srt_pt_render_samples_device", stream, seed, sample_base, samples, nullptr);
}

int srt_pt_fold_epochs_device(srt_pt* pt, void* stream, uint32_t samples_per_epoch, uint32_t position, uint32_t total_samples,
                              uint32_t accumulator_samples, float* d_accumulator) {
  int st = need_ready(pt, "srt_pt_fold_epochs_device");
  if (st != SRT_OK) return st;
  if (!d_accumulator || !samples_per_epoch) return srt::fail(SRT_ERR_INVALID, "srt_pt_fold_epochs_device: bad argument");
  hipStream_t s = (hipStream_t)stream;
  auto it = pt->epoch_buffers.find(s);
  const uint32_t px = pt->tiles.local_tiles * pt->tiles.tile_w * pt->tiles.tile_h;
  if (!px) return SRT_OK;                               // this rank owns no tile
  if (it == pt->epoch_buffers.end() || !it->second.d_samples || it->second.last_npix != px || !it->second.last_samples)
    return srt::fail(SRT_ERR_STATE, "srt_pt_fold_epochs_device: no launch of srt_pt_render_samples_device on this stream to fold");
  const srt_pt::EpochBuffers& B = it->second;
  if ((uint64_t)position + B.last_samples > total_samples)
    return srt::fail(SRT_ERR_INVALID, "srt_pt_fold_epochs_device: samples %u + %u exceed the render's %u", position, B.last_samples, total_samples);
  pt_fold_kernel<<<dim3((px + 255) / 256), dim3(256), 0, s>>>(pt->tiles, pt->w, pt->h, B.last_samples, B.d_samples, samples_per_epoch, position, total_samples,
                                                              accumulator_samples, d_accumulator, B.d_cancel);
  SRT_HIP(hipGetLastError());
  return SRT_OK;
}

int srt_pt_accumulator_tiles_device(srt_pt* pt, void* stream, const float* d_accumulator, float* d_tiles_out) {
  int st = need_ready(pt, "srt_pt_accumulator_tiles_device");
  if (st != SRT_OK) return st;
  if (!d_accumulator || !d_tiles_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_accumulator_tiles_device: NULL buffer");
  const uint32_t px = pt->tiles.local_tiles * pt->tiles.tile_w * pt->tiles.tile_h;
  if (!px) return SRT_OK;
  pt_acc_image_kernel<<<dim3((px + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(pt->tiles, d_accumulator, d_tiles_out);
  SRT_HIP(hipGetLastError());
  return SRT_OK;
}

int srt_pt_untile_device(srt_pt* pt, void* stream, const float* d_gathered, float* d_image) {
  int st = need_ready(pt, "srt_pt_untile_device");
  if (st != SRT_OK) return st;
  if (!d_gathered || !d_image) return srt::fail(SRT_ERR_INVALID, "srt_pt_untile_device: NULL buffer");
  hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream; NULL is the HIP default stream
  const uint32_t n = pt->w * pt->h;
  pt_untile_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(pt->tiles, pt->w, pt->h, pt->tiles_per_rank, d_gathered, d_image);
  SRT_HIP(hipGetLastError());
  return SRT_OK;
}

int srt_pt_accumulate_device(srt_pt* pt, void* stream, float* d_accumulator, const float* d_epoch, size_t nfloats,
                             uint32_t accumulator_samples) {
  int st = need_device(pt, "srt_pt_accumulate_device");
  if (st != SRT_OK) return st;
  if (!d_accumulator || !d_epoch || !accumulator_samples) return srt::fail(SRT_ERR_INVALID, "srt_pt_accumulate_device: bad argument");
  hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream; NULL is the HIP default stream
  pt_accumulate_kernel<<<dim3((unsigned)((nfloats + 255) / 256)), dim3(256), 0, s>>>(d_accumulator, d_epoch, nfloats,
                                                                                 1.0f / accumulator_samples);
  SRT_HIP(hipGetLastError());
  return SRT_OK;
}

int srt_pt_render_epoch(srt_pt* pt, uint64_t seed, uint32_t sample_base, uint32_t samples, float* rgb_out) {
  int st = need_ready(pt, "srt_pt_render_epoch");
  if (st != SRT_OK) return st;
  if (!rgb_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_render_epoch: output is NULL");
  const TileMap& T = pt->tiles;
  const size_t per_tile = (size_t)T.tile_w * T.tile_h * 3;
  const size_t need = per_tile * (T.local_tiles ? T.local_tiles : 1);
  if (pt->tile_buf_floats < need) {
    if (pt->d_tile_buf) SRT_HIP(hipFree(pt->d_tile_buf));
    pt->d_tile_buf = nullptr;
    SRT_HIP(hipMalloc(&pt->d_tile_buf, need * sizeof(float)));
    pt->tile_buf_floats = need;
  }
  st = srt_pt_render_epoch_device(pt, (void*)pt->stream, seed, sample_base, samples, pt->d_tile_buf);
  if (st != SRT_OK) return st;
  SRT_HIP(hipStreamSynchronize(pt->stream));
  if (__atomic_load_n(pt->h_cancel, __ATOMIC_ACQUIRE) != 0u) return SRT_CANCELLED;   // the epoch was cut short: rgb_out is not written
  if ((st = check_stream_fault(pt, "srt_pt_render_epoch")) != SRT_OK) return st;
  std::vector<float> host(per_tile * T.local_tiles);
  if (!host.empty()) SRT_HIP(hipMemcpy(host.data(), pt->d_tile_buf, host.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (uint32_t k = 0; k < T.local_tiles; k++) {
    const uint32_t tile = T.rank + k * T.world;
    const uint32_t x0 = (tile % T.tiles_x) * T.tile_w, y0 = (tile / T.tiles_x) * T.tile_h;
    for (uint32_t ly = 0; ly < T.tile_h && y0 + ly < pt->h; ly++)
      for (uint32_t lx = 0; lx < T.tile_w && x0 + lx < pt->w; lx++) {
        const float* src = &host[k * per_tile + ((size_t)ly * T.tile_w + lx) * 3];
        float* dst = rgb_out + ((size_t)(y0 + ly) * pt->w + (x0 + lx)) * 3;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
      }
  }
  return SRT_OK;
}

int srt_pt_section_cycles(srt_pt* pt, uint64_t out[8], int reset) {
  int st = need_device(pt, "srt_pt_section_cycles");
  if (st != SRT_OK) return st;
  if (!out) return srt::fail(SRT_ERR_INVALID, "srt_pt_section_cycles: NULL argument");
  for (int i = 0; i < 8; i++) out[i] = 0;
  SRT_HIP(hipDeviceSynchronize());
  for (auto& kv : pt->epoch_buffers) {   // summed over the streams that rendered
    if (!kv.second.d_queue) continue;
    unsigned long long h[ST_COUNT_];
    SRT_HIP(hipMemcpy(h, kv.second.d_queue + 1, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < ST_COUNT_; i++) out[i] += h[i];
    if (reset) SRT_HIP(hipMemset(kv.second.d_queue + 1, 0, sizeof h));
  }
  return SRT_OK;
}

int srt_pt_kernel_time(srt_pt* pt, int enable, double* total_ms, uint64_t* launches) {
  int st = need_device(pt, "srt_pt_kernel_time");
  if (st != SRT_OK) return st;
  double sum = 0.0;
  uint64_t n = 0;
  for (auto& ev : pt->timed) {
    SRT_HIP(hipEventSynchronize(ev.second));
    float ms = 0.f;
    SRT_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
    sum += ms;
    n++;
    pt->spare.push_back(ev);
  }
  pt->timed.clear();
  pt->timing = enable != 0;
  if (total_ms) *total_ms = sum;
  if (launches) *launches = n;
  return SRT_OK;
}

int srt_pt_stream_times(srt_pt* pt, int enable, double ms_out[4], uint64_t* generations) {
  int st = need_device(pt, "srt_pt_stream_times");
  if (st != SRT_OK) return st;
  for (int k = 0; k < 4; k++) {
    double sum = 0.0;
    for (auto& ev : pt->stream_timed[k]) {
      SRT_HIP(hipEventSynchronize(ev.second));
      float ms = 0.f;
      SRT_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
      sum += ms;
      pt->spare.push_back(ev);
    }
    pt->stream_timed[k].clear();
    if (ms_out) ms_out[k] = sum;
  }
  if (generations) *generations = pt->stream_generations;
  pt->stream_generations = 0;
  pt->stream_timing = enable != 0;
  return SRT_OK;
}

int srt_pt_stream_counters(srt_pt* pt, uint64_t out[4], int reset) {
  int st = need_device(pt, "srt_pt_stream_counters");
  if (st != SRT_OK) return st;
  if (!out) return srt::fail(SRT_ERR_INVALID, "srt_pt_stream_counters: out is NULL");
  SRT_HIP(hipDeviceSynchronize());
  unsigned long long h[2] = {0, 0};
  SRT_HIP(hipMemcpy(h, pt->d_totals + C_COUNT + 2, sizeof h, hipMemcpyDeviceToHost));
  out[0] = h[0]; out[1] = h[1];
  // Bytes the streamed forms move through memory per unit, from the layout in pt_wave.h / pt_stream.h (three-ray batches):
  //   per alive slot and generation: the resolve kernel loads and stores the flags word and 27 state words (28 + 28), the probe kernel
  //   loads the flags and the 14 words of the batch's rays and stores the emit word (15 + 1), the compaction reads the emit word (1)
  //   per queued entry: origin + direction planes (2 x 16 B) and the list entry (4 B) written and read once each, the hit (8 B) written and read
  out[2] = (28u + 28u + 15u + 1u + 1u) * 4u;
  out[3] = (32u + 4u + 8u) * 2u;
  if (reset) { SRT_HIP(hipMemset(pt->d_totals + C_COUNT + 2, 0, sizeof h)); SRT_HIP(hipDeviceSynchronize()); }
  return SRT_OK;
}

int srt_pt_kernel_form(srt_pt* pt, int* form) {
  if (!pt || !form) return srt::fail(SRT_ERR_INVALID, "srt_pt_kernel_form: NULL argument");
  if (!pt->committed) return srt::fail(SRT_ERR_STATE, "srt_pt_kernel_form before srt_pt_scene_commit");
  const int t = wave_trav(pt);
  *form = t >= 0 ? t : (pt->kernel_mode == 1 ? -2 : -1);
  return SRT_OK;
}

int srt_pt_ray_count(srt_pt* pt, uint64_t* rays, uint64_t* camera_samples, int reset) {
  int st = need_device(pt, "srt_pt_ray_count");
  if (st != SRT_OK) return st;
  SRT_HIP(hipDeviceSynchronize());  // epochs may be in flight on any stream
  if ((st = check_stream_fault(pt, "srt_pt_ray_count")) != SRT_OK) return st;
  unsigned long long r = 0;
  SRT_HIP(hipMemcpy(&r, pt->d_totals + C_COUNT, sizeof r, hipMemcpyDeviceToHost));
  if (rays) *rays = r;
  if (camera_samples) *camera_samples = pt->camera_samples;
  if (reset) {
    SRT_HIP(hipMemset(pt->d_totals + C_COUNT, 0, sizeof r));
    SRT_HIP(hipDeviceSynchronize());                     // (null-stream memset: done before the next epoch's atomics on another stream)
    pt->camera_samples = 0;
  }
  return SRT_OK;
}

int srt_pt_set_elision(srt_pt* pt, int on) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_elision: NULL context");
  pt->elide = on ? 1 : 0;
  pt->wave_blocks = 0;                                    // the launch configuration is re-derived for the other build
  return SRT_OK;
}

int srt_pt_rays_elided(srt_pt* pt, uint64_t* elided, int reset) {
  int st = need_device(pt, "srt_pt_rays_elided");
  if (st != SRT_OK) return st;
  SRT_HIP(hipDeviceSynchronize());
  unsigned long long r = 0;
  SRT_HIP(hipMemcpy(&r, pt->d_totals + C_COUNT + 1, sizeof r, hipMemcpyDeviceToHost));
  if (elided) *elided = r;
  if (reset) { SRT_HIP(hipMemset(pt->d_totals + C_COUNT + 1, 0, sizeof r)); SRT_HIP(hipDeviceSynchronize()); }
  return SRT_OK;
}

int srt_pt_trace_samples(srt_pt* pt, uint64_t seed, const uint32_t* xs, const uint32_t* ys, const uint32_t* ss, size_t n,
                         float* rgb_out, uint32_t* draws_out, uint32_t* rays_out) {
  int st = need_ready(pt, "srt_pt_trace_samples");
  if (st != SRT_OK) return st;
  if (n == 0) return SRT_OK;
  if (!xs || !ys || !ss || !rgb_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_trace_samples: NULL argument");
  if (n > 0x7fffffffull) return srt::fail(SRT_ERR_UNSUPPORTED, "too many samples in one call");
  for (size_t i = 0; i < n; i++)
    if (xs[i] >= pt->w || ys[i] >= pt->h) return srt::fail(SRT_ERR_INVALID, "sample %zu: pixel (%u,%u) outside %ux%u", i, xs[i], ys[i], pt->w, pt->h);
  uint32_t *dx = nullptr, *dy = nullptr, *ds = nullptr, *dd = nullptr, *dr = nullptr;
  float* drgb = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&dx, n * 4)); SRT_HIP(tmp.alloc(&dy, n * 4)); SRT_HIP(tmp.alloc(&ds, n * 4));
  SRT_HIP(tmp.alloc(&dd, n * 4)); SRT_HIP(tmp.alloc(&dr, n * 4)); SRT_HIP(tmp.alloc(&drgb, n * 12));
  SRT_HIP(hipMemcpyAsync(dx, xs, n * 4, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(dy, ys, n * 4, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(ds, ss, n * 4, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemsetAsync(pt->d_totals, 0, C_COUNT * sizeof(unsigned long long), pt->stream));
  pt_samples_kernel<true><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, pt->stream>>>(device_scene(pt), seed, dx, dy, ds, (uint32_t)n,
                                                                                   drgb, dd, dr, pt->d_totals);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(rgb_out, drgb, n * 12, hipMemcpyDeviceToHost, pt->stream));
  if (draws_out) SRT_HIP(hipMemcpyAsync(draws_out, dd, n * 4, hipMemcpyDeviceToHost, pt->stream));
  if (rays_out) SRT_HIP(hipMemcpyAsync(rays_out, dr, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipMemcpyAsync(pt->last_counters, pt->d_totals, sizeof pt->last_counters, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

int srt_pt_hit(srt_pt* pt, const float* origins, const float* dirs, const float* bounds, size_t n, float* out9) {
  int st = need_device(pt, "srt_pt_hit");
  if (st != SRT_OK) return st;
  if (!pt->committed) return srt::fail(SRT_ERR_STATE, "srt_pt_hit before srt_pt_scene_commit");
  if (n == 0) return SRT_OK;
  if (!origins || !dirs || !bounds || !out9) return srt::fail(SRT_ERR_INVALID, "srt_pt_hit: NULL argument");
  float *dorg = nullptr, *ddir = nullptr, *db = nullptr, *dout = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&dorg, n * 12)); SRT_HIP(tmp.alloc(&ddir, n * 12)); SRT_HIP(tmp.alloc(&db, n * 8)); SRT_HIP(tmp.alloc(&dout, n * 36));
  SRT_HIP(hipMemcpyAsync(dorg, origins, n * 12, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(ddir, dirs, n * 12, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(db, bounds, n * 8, hipMemcpyHostToDevice, pt->stream));
  DScene S = device_scene(pt);
  if (pt->kernel_mode == 5) {
    if (!flat_walk_fits(pt->built.flat)) return srt::fail(SRT_ERR_UNSUPPORTED, "the flattened walk needs 1..31 objects");
    pt_hit_kernel<true><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, pt->stream>>>(S, dorg, ddir, db, (uint32_t)n, dout);
  } else {
    pt_hit_kernel<false><<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, pt->stream>>>(S, dorg, ddir, db, (uint32_t)n, dout);
  }
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(out9, dout, n * 36, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

int srt_pt_particles_step_device(srt_pt* pt, void* stream, float* d_pos, float* d_vel, float* d_age, size_t n, float dt, float radius,
                                 uint8_t* d_alive) {
  int st = need_device(pt, "srt_pt_particles_step_device");
  if (st != SRT_OK) return st;
  if (!pt->committed) return srt::fail(SRT_ERR_STATE, "srt_pt_particles_step before srt_pt_scene_commit");
  if (n == 0) return SRT_OK;
  if (!d_pos || !d_vel || !d_age || !d_alive) return srt::fail(SRT_ERR_INVALID, "srt_pt_particles_step: NULL argument");
  if (n > 0x7fffffffull) return srt::fail(SRT_ERR_UNSUPPORTED, "too many particles in one call");
  pt_particles_kernel<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream>>>(device_scene(pt), d_pos, d_vel, d_age, (uint32_t)n, dt,
                                                                                           radius, d_alive);
  SRT_HIP(hipGetLastError());
  return SRT_OK;
}

int srt_pt_particles_step(srt_pt* pt, float* pos, float* vel, float* age, size_t n, float dt, float radius, uint8_t* alive) {
  int st = need_device(pt, "srt_pt_particles_step");
  if (st != SRT_OK) return st;
  if (n == 0) return SRT_OK;
  if (!pos || !vel || !age || !alive) return srt::fail(SRT_ERR_INVALID, "srt_pt_particles_step: NULL argument");
  float *dp = nullptr, *dv = nullptr, *da = nullptr;
  uint8_t* dl = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&dp, n * 12)); SRT_HIP(tmp.alloc(&dv, n * 12)); SRT_HIP(tmp.alloc(&da, n * 4)); SRT_HIP(tmp.alloc(&dl, n));
  SRT_HIP(hipMemcpyAsync(dp, pos, n * 12, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(dv, vel, n * 12, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(da, age, n * 4, hipMemcpyHostToDevice, pt->stream));
  st = srt_pt_particles_step_device(pt, (void*)pt->stream, dp, dv, da, n, dt, radius, dl);
  if (st != SRT_OK) return st;
  SRT_HIP(hipMemcpyAsync(pos, dp, n * 12, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipMemcpyAsync(vel, dv, n * 12, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipMemcpyAsync(age, da, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipMemcpyAsync(alive, dl, n, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

long srt_pt_dump_bvh(srt_pt* pt, int which, float* boxes, uint32_t* links, size_t cap, uint32_t* order) {
  if (!pt || !boxes || !links) return srt::fail(SRT_ERR_INVALID, "srt_pt_dump_bvh: NULL argument");
  if (!pt->committed) return srt::fail(SRT_ERR_STATE, "srt_pt_dump_bvh before srt_pt_scene_commit");
  if (!pt->built.flat.use_bvh) return srt::fail(SRT_ERR_STATE, "scene was committed without BVHs");
  const HostBVH* b = &pt->built.tlas;
  const ObjectInput* in = nullptr;
  if (which >= 0) {
    if ((size_t)which >= pt->built.tlas.prim.size()) return srt::fail(SRT_ERR_INVALID, "object slot %d out of range", which);
    const uint32_t obj = pt->built.tlas.prim[which];
    in = &pt->built.inputs[obj];
    if (in->kind != OBJ_MESH) return srt::fail(SRT_ERR_INVALID, "object slot %d is not a mesh", which);
    b = &pt->built.blas[obj];
  }
  for (size_t i = 0; i < b->nodes.size() && i < cap; i++) {
    const HostNode& nd = b->nodes[i];
    for (int a = 0; a < 3; a++) { boxes[6 * i + a] = nd.mn[a]; boxes[6 * i + 3 + a] = nd.mx[a]; }
    links[4 * i] = nd.start; links[4 * i + 1] = nd.size; links[4 * i + 2] = nd.l; links[4 * i + 3] = nd.r;
  }
  if (order)
    for (size_t i = 0; i < b->prim.size(); i++) order[i] = in ? in->mesh.idx[3 * b->prim[i]] : b->prim[i] + 1;
  return (long)b->nodes.size();
}

int srt_pt_counters(srt_pt* pt, uint64_t out[8]) {
  if (!pt || !out) return srt::fail(SRT_ERR_INVALID, "srt_pt_counters: NULL argument");
  for (int k = 0; k < C_COUNT; k++) out[k] = pt->last_counters[k];
  return SRT_OK;
}

int srt_pt_math_cos_sin(srt_pt* pt, const float* x, size_t n, float* cos_out, float* sin_out) {
  int st = need_device(pt, "srt_pt_math_cos_sin");
  if (st != SRT_OK) return st;
  if (n == 0) return SRT_OK;
  float *dx = nullptr, *dc = nullptr, *dsn = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&dx, n * 4)); SRT_HIP(tmp.alloc(&dc, n * 4)); SRT_HIP(tmp.alloc(&dsn, n * 4));
  SRT_HIP(hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, pt->stream));
  pt_math_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pt->stream>>>(dx, n, dc, dsn);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(cos_out, dc, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipMemcpyAsync(sin_out, dsn, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

int srt_pt_math_acos(srt_pt* pt, const float* x, size_t n, float* out) {
  int st = need_device(pt, "srt_pt_math_acos");
  if (st != SRT_OK) return st;
  if (!x || !out) return srt::fail(SRT_ERR_INVALID, "srt_pt_math_acos: NULL argument");
  if (!n) return SRT_OK;
  float *dx = nullptr, *dout = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&dx, n * 4)); SRT_HIP(tmp.alloc(&dout, n * 4));
  SRT_HIP(hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, pt->stream));
  pt_acos_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pt->stream>>>(dx, n, dout);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

int srt_pt_math_atan2(srt_pt* pt, const float* y, const float* x, size_t n, float* out) {
  int st = need_device(pt, "srt_pt_math_atan2");
  if (st != SRT_OK) return st;
  if (!y || !x || !out) return srt::fail(SRT_ERR_INVALID, "srt_pt_math_atan2: NULL argument");
  if (!n) return SRT_OK;
  float *dy = nullptr, *dx = nullptr, *dout = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&dy, n * 4)); SRT_HIP(tmp.alloc(&dx, n * 4)); SRT_HIP(tmp.alloc(&dout, n * 4));
  SRT_HIP(hipMemcpyAsync(dy, y, n * 4, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, pt->stream));
  pt_atan2_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pt->stream>>>(dy, dx, n, dout);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

int srt_pt_tonemap_device(srt_pt* pt, void* stream, const float* d_rgb, uint32_t width, uint32_t height, float exposure, uint8_t* d_rgba) {
  int st = need_device(pt, "srt_pt_tonemap_device");
  if (st != SRT_OK) return st;
  if (!d_rgb || !d_rgba) return srt::fail(SRT_ERR_INVALID, "srt_pt_tonemap_device: NULL argument");
  if (!(exposure > 0.0f)) return srt::fail(SRT_ERR_INVALID, "srt_pt_tonemap_device: exposure must be positive (got %g)", (double)exposure);
  if ((reinterpret_cast<uintptr_t>(d_rgba) & 3u) != 0) return srt::fail(SRT_ERR_INVALID, "srt_pt_tonemap_device: rgba must be 4-byte aligned");
  const size_t px = (size_t)width * height;
  if (!px) return SRT_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);      // exactly the caller's stream; NULL is the HIP default stream, as for every *_device call
  pt_tonemap_kernel<<<dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s>>>(d_rgb, width, height, exposure, reinterpret_cast<uint32_t*>(d_rgba));
  SRT_HIP(hipGetLastError());
  return SRT_OK;
}

int srt_pt_tonemap(srt_pt* pt, const float* rgb, uint32_t width, uint32_t height, float exposure, uint8_t* rgba_out) {
  int st = need_device(pt, "srt_pt_tonemap");
  if (st != SRT_OK) return st;
  if (!rgb || !rgba_out) return srt::fail(SRT_ERR_INVALID, "srt_pt_tonemap: NULL argument");
  if (!(exposure > 0.0f)) return srt::fail(SRT_ERR_INVALID, "srt_pt_tonemap: exposure must be positive (got %g)", (double)exposure);
  const size_t px = (size_t)width * height;
  if (!px) return SRT_OK;
  float* d_in = nullptr; uint8_t* d_out = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&d_in, px * 12));
  SRT_HIP(tmp.alloc(&d_out, px * 4));
  st = SRT_OK;
  if (hipMemcpyAsync(d_in, rgb, px * 12, hipMemcpyHostToDevice, pt->stream) != hipSuccess) st = srt::fail(SRT_ERR_HIP, "srt_pt_tonemap: upload failed");
  if (st == SRT_OK) st = srt_pt_tonemap_device(pt, (void*)pt->stream, d_in, width, height, exposure, d_out);
  if (st == SRT_OK && (hipMemcpyAsync(rgba_out, d_out, px * 4, hipMemcpyDeviceToHost, pt->stream) != hipSuccess ||
                       hipStreamSynchronize(pt->stream) != hipSuccess))
    st = srt::fail(SRT_ERR_HIP, "srt_pt_tonemap: download failed");
  return st;
}

int srt_pt_math_exp(srt_pt* pt, const float* x, size_t n, float* out) {
  int st = need_device(pt, "srt_pt_math_exp");
  if (st != SRT_OK) return st;
  if (!x || !out) return srt::fail(SRT_ERR_INVALID, "srt_pt_math_exp: NULL argument");
  if (!n) return SRT_OK;
  float *dx = nullptr, *dout = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&dx, n * 4)); SRT_HIP(tmp.alloc(&dout, n * 4));
  SRT_HIP(hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, pt->stream));
  pt_exp_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pt->stream>>>(dx, n, dout);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

int srt_pt_math_pow(srt_pt* pt, const float* x, const float* y, size_t n, float* out) {
  int st = need_device(pt, "srt_pt_math_pow");
  if (st != SRT_OK) return st;
  if (!x || !y || !out) return srt::fail(SRT_ERR_INVALID, "srt_pt_math_pow: NULL argument");
  if (!n) return SRT_OK;
  float *dx = nullptr, *dy = nullptr, *dout = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&dx, n * 4)); SRT_HIP(tmp.alloc(&dy, n * 4)); SRT_HIP(tmp.alloc(&dout, n * 4));
  SRT_HIP(hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, pt->stream));
  SRT_HIP(hipMemcpyAsync(dy, y, n * 4, hipMemcpyHostToDevice, pt->stream));
  pt_pow_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pt->stream>>>(dx, dy, n, dout);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}

int srt_pt_math_div_sqrt(srt_pt* pt, const float* in, size_t lanes, int shared_c2, float* out) {
  int st = need_device(pt, "srt_pt_math_div_sqrt");
  if (st != SRT_OK) return st;
  if (!in || !out) return srt::fail(SRT_ERR_INVALID, "srt_pt_math_div_sqrt: NULL argument");
  if (!lanes) return SRT_OK;
  const size_t n3 = lanes * 3;
  float *din = nullptr, *dout = nullptr;
  srt::DeviceScratch tmp;
  SRT_HIP(tmp.alloc(&din, 5 * n3 * 4)); SRT_HIP(tmp.alloc(&dout, 4 * n3 * 4));
  SRT_HIP(hipMemcpyAsync(din, in, 5 * n3 * 4, hipMemcpyHostToDevice, pt->stream));
  pt_div_sqrt_kernel<<<dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, pt->stream>>>(din, lanes, shared_c2, dout);
  SRT_HIP(hipGetLastError());
  SRT_HIP(hipMemcpyAsync(out, dout, 4 * n3 * 4, hipMemcpyDeviceToHost, pt->stream));
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return SRT_OK;
}


// ---- Pathtracer::cancel ---------------------------------------------------------------------------------------------
int srt_pt_cancel(srt_pt* pt) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_cancel: NULL context");
  if (pt->h_cancel) __atomic_store_n(pt->h_cancel, 1u, __ATOMIC_RELEASE);   // (no HIP call: any thread, any time)
  return SRT_OK;
}

int srt_pt_cancel_requested(srt_pt* pt) {
  return pt && pt->h_cancel && __atomic_load_n(pt->h_cancel, __ATOMIC_ACQUIRE) != 0u ? 1 : 0;
}

int srt_pt_clear_cancel(srt_pt* pt) {
  int st = need_device(pt, "srt_pt_clear_cancel");
  if (st != SRT_OK) return st;
  SRT_HIP(hipDeviceSynchronize());                       // what was in flight has drained (it ends early once the flag is seen)
  __atomic_store_n(pt->h_cancel, 0u, __ATOMIC_RELEASE);
  for (auto& kv : pt->epoch_buffers)
    if (kv.second.d_cancel) SRT_HIP(hipMemset(kv.second.d_cancel, 0, sizeof(uint32_t)));
  SRT_HIP(hipDeviceSynchronize());                       // (the memsets are on the null stream: done before any stream's next kernel)
  *(volatile uint32_t*)pt->h_fault = 0u;                 // (a cancelled streamed launch leaves no fault, but nothing stale either)
  return SRT_OK;
}

// ---- Pathtracer::log_ray --------------------------------------------------------------------------------------------
int srt_pt_set_ray_log(srt_pt* pt, uint32_t capacity) {
  if (!pt) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_ray_log: NULL context");
  if (capacity > (1u << 26)) return srt::fail(SRT_ERR_INVALID, "srt_pt_set_ray_log: at most 2^26 rays per read (got %u)", capacity);
  pt->ray_log_cap = capacity;                            // the rings follow at the next epoch on each stream (stream_buffers)
  return SRT_OK;
}

namespace {
// `rays` == nullptr: only count what is waiting in the ring (nothing is consumed).
int read_ring(srt_pt::EpochBuffers& B, hipStream_t s, std::vector<srt_pt_logged_ray>* rays, size_t* waiting, uint64_t* dropped) {
  if (!B.d_ray_log) return SRT_OK;
  uint32_t count = 0;
  SRT_HIP(hipMemcpyAsync(&count, B.d_ray_log, sizeof count, hipMemcpyDeviceToHost, s));
  SRT_HIP(hipStreamSynchronize(s));
  const uint32_t n = count < B.ray_log_cap ? count : B.ray_log_cap;
  if (waiting) *waiting += n;
  if (!rays) return SRT_OK;
  if (dropped) *dropped += count - n;
  if (!count) return SRT_OK;
  std::vector<uint32_t> raw((size_t)kRayLogWords * n);
  if (n) SRT_HIP(hipMemcpyAsync(raw.data(), B.d_ray_log + kRayLogHeader, raw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  SRT_HIP(hipMemsetAsync(B.d_ray_log, 0, sizeof(uint32_t), s));
  SRT_HIP(hipStreamSynchronize(s));
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t* e = &raw[(size_t)kRayLogWords * i];
    srt_pt_logged_ray r;
    std::memcpy(r.point, e, 12); std::memcpy(r.dir, e + 3, 12);
    r.t = 5.0f;                                          // log_ray(world_ray_task6, 5.0f), student/pathtracer.cpp:148
    r.pixel = e[6]; r.sample = e[7] >> 4; r.bounce = e[7] & 15u;
    rays->push_back(r);
  }
  return SRT_OK;
}
int deliver_rays(std::vector<srt_pt_logged_ray>& rays, srt_pt_logged_ray* out, size_t cap, size_t* n_out, uint64_t* dropped) {
  // the order a single-threaded do_trace would log them in: pixel-major, then sample, then bounce
  std::sort(rays.begin(), rays.end(), [](const srt_pt_logged_ray& a, const srt_pt_logged_ray& b) {
    if (a.pixel != b.pixel) return a.pixel < b.pixel;
    if (a.sample != b.sample) return a.sample < b.sample;
    return a.bounce < b.bounce;
  });
  const size_t n = rays.size() < cap ? rays.size() : cap;
  if (n) std::memcpy(out, rays.data(), n * sizeof(srt_pt_logged_ray));
  if (dropped) *dropped += rays.size() - n;              // (what the caller's buffer did not take is gone as well)
  if (n_out) *n_out = n;
  return SRT_OK;
}
int read_log(srt_pt* pt, bool all_streams, hipStream_t s, srt_pt_logged_ray* out, size_t cap, size_t* n_out, uint64_t* dropped) {
  if (dropped) *dropped = 0;
  if (n_out) *n_out = 0;
  int st;
  if (!out) {                                            // how many are waiting
    size_t waiting = 0;
    if (all_streams) SRT_HIP(hipDeviceSynchronize());
    for (auto& kv : pt->epoch_buffers)
      if ((all_streams || kv.first == s) && (st = read_ring(kv.second, kv.first, nullptr, &waiting, nullptr)) != SRT_OK) return st;
    if (n_out) *n_out = waiting;
    return SRT_OK;
  }
  std::vector<srt_pt_logged_ray> rays;
  if (all_streams) SRT_HIP(hipDeviceSynchronize());
  for (auto& kv : pt->epoch_buffers)
    if ((all_streams || kv.first == s) && (st = read_ring(kv.second, kv.first, &rays, nullptr, dropped)) != SRT_OK) return st;
  return deliver_rays(rays, out, cap, n_out, dropped);
}
}  // namespace

int srt_pt_read_ray_log_stream(srt_pt* pt, void* stream, srt_pt_logged_ray* out, size_t cap, size_t* n_out, uint64_t* dropped) {
  int st = need_device(pt, "srt_pt_read_ray_log_stream");
  if (st != SRT_OK) return st;
  return read_log(pt, false, (hipStream_t)stream, out, cap, n_out, dropped);
}

int srt_pt_read_ray_log(srt_pt* pt, srt_pt_logged_ray* out, size_t cap, size_t* n_out, uint64_t* dropped) {
  int st = need_device(pt, "srt_pt_read_ray_log");
  if (st != SRT_OK) return st;
  return read_log(pt, true, nullptr, out, cap, n_out, dropped);
}

int srt_pt_sync(srt_pt* pt) {
  int st = need_device(pt, "srt_pt_sync");
  if (st != SRT_OK) return st;
  SRT_HIP(hipStreamSynchronize(pt->stream));
  return check_stream_fault(pt, "srt_pt_sync");
}

}  // extern "C"

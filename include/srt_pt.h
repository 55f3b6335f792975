/*
 * srt_pt.h — C ABI of the MI355X (gfx950) path-tracer hot path.
 *
 * Drop-in boundary for PT::Pathtracer of the reference (paths relative to
 * /root/reference/Assignments/Scotty3D/src/):
 *
 *   srt_pt_scene_begin / add_material / add_mesh / add_sphere / scene_commit
 *                          <- Pathtracer::build_scene            rays/pathtracer.cpp:66-176
 *                             (Object ctor rays/object.h:18-34, Tri_Mesh::build student/tri_mesh.cpp:145-170,
 *                              BVH<>::build student/bvh.inl:35-163 — host or device, structure-identical)
 *   srt_pt_set_camera      <- `camera = cam` in begin_render     rays/pathtracer.cpp:267 (Camera: util/camera.h)
 *   srt_pt_set_params      <- Pathtracer::set_params             rays/pathtracer.cpp:182-189
 *   srt_pt_render_epoch    <- Pathtracer::do_trace(samples)      rays/pathtracer.cpp:209-231, i.e. for every pixel
 *                             trace_pixel (student/pathtracer.cpp:14-40) -> trace (:174-218) ->
 *                             sample_direct_lighting (:78-172) / sample_indirect_lighting (:42-76) ->
 *                             BVH<>::hit (student/bvh.inl:227-276), Triangle::hit (student/tri_mesh.cpp:32-111),
 *                             Sphere::hit (student/shapes.cpp:17-80), BBox::hit (student/bbox.cpp:5-62),
 *                             BSDF_*::scatter (student/bsdf.cpp:69-154), Samplers (student/samplers.cpp)
 *   srt_pt_accumulate      <- Pathtracer::accumulate             rays/pathtracer.cpp:195-207
 *
 * The epoch loop, progress/cancel bookkeeping and the GUI texture stay in the host class
 * (soft-rendering-toolsets_amd/host/pathtracer_hip.cpp); see INTEGRATION.md.
 *
 * Determinism: util/rand.cpp (thread_local mt19937 seeded from random_device) is replaced by SRT-RNG v1,
 * a counter-keyed generator re-keyed per (seed, pixel, sample); DESIGN.md states it.  Image rows follow
 * HDR_Image: row 0 = bottom (util/hdr_image.cpp:54-57).  Matrices are 16 floats in Mat4::data order
 * (column-major, lib/mat4.h).
 *
 *   srt_pt_tonemap         <- HDR_Image::tonemap_to              util/hdr_image.cpp:161-187 (Spectrum::to_srgb lib/spectrum.h:61-75)
 *   srt_pt_add_light / set_env_light / set_env_map / add_sphere_light
 *                          <- Pathtracer::build_lights           rays/pathtracer.cpp:26-64 (rays/light.cpp, student/env_light.cpp)
 *
 * Status codes and srt_last_error(): see srt_raster.h.  No CPU fallback: srt_pt_create fails without a
 * HIP device.
 *
 * Threads: a context is used from one host thread at a time (the reference's Pathtracer owns one worker per render).
 * The *_device entry points only enqueue work on the given stream; epochs enqueued on different streams may overlap on
 * the GPU (the library keeps one set of epoch scratch buffers per stream).
 */
#ifndef SRT_PT_H
#define SRT_PT_H

#include <stddef.h>
#include <stdint.h>

#include "srt_raster.h" /* srt_status, srt_last_error */

#ifdef __cplusplus
extern "C" {
#endif

/* Material kinds = the BSDF variants of rays/bsdf.h. */
enum {
    SRT_MAT_LAMBERTIAN = 0,    /* a = albedo as passed to BSDF_Lambertian(albedo) (divided by PI_F inside) */
    SRT_MAT_MIRROR = 1,        /* a = reflectance */
    SRT_MAT_GLASS = 2,         /* a = transmittance, b = reflectance, ior */
    SRT_MAT_DIFFUSE_LIGHT = 3, /* a = emitted radiance (Material::emissive()) */
    SRT_MAT_REFRACT = 4        /* a = transmittance, ior (the reference's scatter is a stub) */
};

typedef struct srt_pt_material {
    uint32_t type;
    float a[3];
    float b[3];
    float ior;
} srt_pt_material;

typedef struct srt_pt srt_pt; /* opaque */

int srt_pt_create(int device, srt_pt** out);
int srt_pt_destroy(srt_pt* pt);

/* ---- one render over the GPUs of a node, inside one process ----------------------------------
 * The reference's parallel axis is the epoch fan-out of Pathtracer::begin_render over its thread pool
 * (rays/pathtracer.cpp:250-280); here an epoch is cut into 32 x 32 image tiles dealt round-robin to `n` devices.  A group
 * owns one context per rank (rank r renders the tiles t with t % n == r; the scene is replicated: make every scene / camera /
 * kernel call on each srt_pt_group_context(g, r), r = 0 .. n-1), one stream per rank, and the exchange buffers.
 * srt_pt_group_render_epoch = every rank's srt_pt_render_epoch_device, ONE ncclGather of the tile radiance to rank 0 over
 * RCCL / xGMI (no reduction: the tiles are disjoint), srt_pt_untile_device on rank 0; the image is bit-identical to a
 * single context's.  RCCL is loaded at run time when n > 1 and every rank has its own device; ranks that share a device
 * (how the path is exercised on a one-GPU box) gather with device-to-device copies (SRT_PT_GATHER=copy|rccl forces one).
 * srt_pt_group_set_params replaces srt_pt_set_params for the members (it also sizes the exchange buffers).
 * The device form leaves the width*height*3 float image (row 0 = bottom) on rank 0's device and returns the stream it is
 * ordered on (srt_pt_accumulate_device / srt_pt_tonemap_device can follow on that stream); the host form copies it out.
 * One process per GPU with an external collective (torch.distributed / RCCL) remains available through srt_pt_set_tiling. */
typedef struct srt_pt_group srt_pt_group; /* opaque */
int srt_pt_create_multi(const int* devices, int n, srt_pt_group** out);
int srt_pt_group_destroy(srt_pt_group* g);
int srt_pt_group_size(srt_pt_group* g);
srt_pt* srt_pt_group_context(srt_pt_group* g, int rank);
int srt_pt_group_uses_rccl(srt_pt_group* g);
int srt_pt_group_set_params(srt_pt_group* g, uint32_t width, uint32_t height, uint32_t max_depth);
int srt_pt_group_render_epoch(srt_pt_group* g, uint64_t seed, uint32_t sample_base, uint32_t samples, float* rgb_out);
int srt_pt_group_render_epoch_device(srt_pt_group* g, uint64_t seed, uint32_t sample_base, uint32_t samples, float** d_image_out,
                                     void** stream_out);
/* Device time of the exchange step of srt_pt_group_render_epoch[_device] - from the moment rank 0's own tiles are rendered to the end of
 * the un-tiling kernel on rank 0's stream: the gather (RCCL, or copies between ranks that share a device) and what it waits for, i.e. the
 * slowest other rank - summed over the epochs since the previous call (HIP events; waits for them); then recording on / off.  A diagnostic
 * for bench.py --group: together with the members' srt_pt_kernel_time it decomposes a multi-GPU step. */
int srt_pt_group_gather_time(srt_pt_group* g, int enable, double* total_ms, uint64_t* epochs);

/* ---- build_scene ---------------------------------------------------------------------------- */
int srt_pt_scene_begin(srt_pt* pt);
int srt_pt_add_material(srt_pt* pt, const srt_pt_material* m, uint32_t* index_out);
/* Object(Tri_Mesh(mesh, use_bvh), id, material, T).  positions/normals: nverts*3 floats; indices: 3 per
 * triangle.  is_area_light != 0 also appends the Tri_Mesh(mesh, false) copy to the area-light list
 * (materials of kind DIFFUSE_LIGHT, rays/pathtracer.cpp:105-116). */
int srt_pt_add_mesh(srt_pt* pt, const float* positions, const float* normals, uint32_t nverts,
                    const uint32_t* indices, uint32_t nindices, const float trans[16], uint32_t material,
                    int is_area_light);
/* Object(Shape(Sphere(radius)), id, material, T). */
int srt_pt_add_sphere(srt_pt* pt, float radius, const float trans[16], uint32_t material);
/* An emissive analytic sphere (a Scene_Object with a Shape and a diffuse_light material): rays intersect the sphere
 * (Sphere::hit), while the area-light list gets its triangle approximation obj.posed_mesh() as a Tri_Mesh without BVH
 * (rays/pathtracer.cpp:105-116).  positions / normals / indices = that mesh. */
int srt_pt_add_sphere_light(srt_pt* pt, float radius, const float trans[16], uint32_t material, const float* positions,
                            const float* normals, uint32_t nverts, const uint32_t* indices, uint32_t nindices);

/* A delta light (Pathtracer::point_lights, rays/pathtracer.cpp:26-64; rays/light.{h,cpp}): radiance =
 * Scene_Light::radiance(), trans = light.pose.transform() (column-major), angle_bounds (degrees) for spot lights
 * only.  Shadow rays of these lights are counted as rays like every other scene.hit. */
#define SRT_LIGHT_DIRECTIONAL 0u
#define SRT_LIGHT_POINT 1u
#define SRT_LIGHT_SPOT 2u
int srt_pt_add_light(srt_pt* pt, uint32_t type, const float radiance[3], const float angle_bounds[2], const float trans[16]);

/* The environment light (Pathtracer::env_light, rays/env_light.h): a uniform sphere (Env_Sphere) or upper hemisphere
 * (Env_Hemisphere) of the given radiance; rays that leave the scene see it, and sample_area_lights /
 * area_lights_pdf mix it with the area lights as the reference does (a coin flip, the mean of the pdfs). */
#define SRT_ENV_NONE 0u
#define SRT_ENV_SPHERE 1u
#define SRT_ENV_HEMISPHERE 2u
int srt_pt_set_env_light(srt_pt* pt, uint32_t type, const float radiance[3]);
/* Env_Map (an HDR_Image as environment): rgb = width * height * 3 floats, pixel (x, y) at index y * width + x as in
 * HDR_Image::at.  As in the reference fork, directions are sampled uniformly (pdf 1 / 4 PI) and Env_Map::evaluate
 * looks the image up bilinearly (student/env_light.cpp:7-93).  Every kernel form takes it (the wave kernel's DL build looks the map
 * up from the regenerated camera direction when a sample that left the scene is resolved). */
#define SRT_ENV_MAP 3u
int srt_pt_set_env_map(srt_pt* pt, uint32_t width, uint32_t height, const float* rgb);

/* Builds every BVH<Triangle> (leaf size 4) and the BVH<Object> (leaf size 1) exactly as the reference
 * does — or the List<> forms when use_bvh == 0 — flattens them and uploads the scene. */
int srt_pt_scene_commit(srt_pt* pt, int use_bvh);
/* Where srt_pt_scene_commit runs BVH<Primitive>::build (student/bvh.inl:35-163): device != 0 (default) builds primitive sets of at
 * least min_primitives (default 16384) on the GPU, smaller ones and device == 0 on the host.  Both produce the reference's node
 * arrays and primitive order bit for bit (the candidate planes' std::partition sequence included); SRT_BVH_BUILDER=host in the
 * environment forces the host build. */
int srt_pt_set_bvh_builder(srt_pt* pt, int device, uint32_t min_primitives);
/* The streamed forms (kernel modes 6 / 7, what auto takes for scenes with a real BVH<Triangle> or many objects) keep this many
 * paths in flight per launch (rounded up to 256; 0 = the default, 3 Mi; at most 2^26, SRT_ERR_INVALID beyond; ~1 KB of device
 * memory per slot).  The image does not depend on it. */
int srt_pt_set_stream_slots(srt_pt* pt, uint32_t slots);

int srt_pt_set_camera(srt_pt* pt, const float iview[16], float vert_fov_deg, float aspect_ratio);
/* Image size and Pathtracer::max_depth.  Limits (SRT_ERR_UNSUPPORTED beyond): width, height <= 65535, width * height < 2^31,
 * max_depth <= 16. */
int srt_pt_set_params(srt_pt* pt, uint32_t width, uint32_t height, uint32_t max_depth);

/* ---- image-tile sharding (one process per GPU) ---------------------------------------------------
 * The image is cut into tile_w x tile_h tiles numbered row-major; rank r of `world` renders the tiles
 * t with t % world == r.  Defaults: 32 x 32, rank 0, world 1. */
int srt_pt_set_tiling(srt_pt* pt, uint32_t tile_w, uint32_t tile_h, uint32_t rank, uint32_t world);
/* Number of tiles this rank renders / the per-rank capacity every rank pads to (ceil(ntiles / world)),
 * and floats per tile (tile_w * tile_h * 3). */
int srt_pt_tile_info(srt_pt* pt, uint32_t* local_tiles, uint32_t* tiles_per_rank, uint32_t* floats_per_tile);

/* ---- do_trace ------------------------------------------------------------------------------------
 * One epoch: for every pixel of this rank's tiles, the mean over the VALID samples
 * sample_base .. sample_base + samples - 1 of trace_pixel (invalid = non-finite, dropped as in
 * rays/pathtracer.cpp:219-222).
 * Host form: rgb_out is a full width*height*3 float image (row 0 = bottom); only this rank's pixels are
 * written.  Synchronous. */
int srt_pt_render_epoch(srt_pt* pt, uint64_t seed, uint32_t sample_base, uint32_t samples, float* rgb_out);
/* Device form: d_tiles_out is DEVICE memory of tiles_per_rank * floats_per_tile floats, tile-major
 * (local tile k = global tile rank + k * world); enqueued on `stream` (hipStream_t; NULL = the HIP
 * default stream, which is also PyTorch's default), not synchronized.  This is the buffer the RCCL gather moves. */
int srt_pt_render_epoch_device(srt_pt* pt, void* stream, uint64_t seed, uint32_t sample_base, uint32_t samples,
                               float* d_tiles_out);
/* d_gathered: DEVICE, world * tiles_per_rank * floats_per_tile floats as gathered on the root (rank-major).
 * Scatters the tiles into the width*height*3 DEVICE image d_image (row 0 = bottom). */
int srt_pt_untile_device(srt_pt* pt, void* stream, const float* d_gathered, float* d_image);
/* Pathtracer::accumulate on DEVICE buffers of nfloats: acc += (epoch - acc) * (1.0f / accumulator_samples). */
int srt_pt_accumulate_device(srt_pt* pt, void* stream, float* d_accumulator, const float* d_epoch, size_t nfloats,
                             uint32_t accumulator_samples);

/* Kernel selection for render_epoch*: 0 = automatic (default: the persistent wave kernel with wave-uniform
 * sweeps for scenes of <= 16 objects whose meshes are single BVH leaves - the Cornell boxes -, the streamed forms 7 / 6
 * for scenes with a real BVH<Triangle> or more objects, the per-lane kernel with one lane per sample for what is left), 1 = per-lane kernel, one lane per pixel (any scene), 2 = persistent wave kernel with wave-uniform
 * sweeps (<= 16 objects; fails otherwise), 3 = the same with in-kernel section stamps (diagnostic build, slower),
 * 4 = per-lane kernel, one lane per sample (any scene), 5 = persistent wave kernel with the flattened per-lane
 * walk of both tree levels (<= 31 objects; fails otherwise; srt_pt_hit then also goes through that walk),
 * 6 = streamed form (any number of objects): a logic kernel per generation (consume hits, shade, refill, emit rays) and a
 * persistent ray-cast kernel that walks one ray per lane through both tree levels with LDS stacks and pulls rays from a
 * dense queue; 7 = streamed sweeps (<= 16 objects of which 1..4 meshes with a real BVH<Triangle>: BASELINE configs[4]): the
 * wave-uniform sweeps stay in the logic kernel, only the walks of those meshes are queued to the ray-cast kernel.
 * Automatic picks 7 where it applies, 6 for scenes the sweeps do not take.  All
 * produce bit-identical images; the switch exists for A/B tests and profiling. */
int srt_pt_set_kernel(srt_pt* pt, int mode);   /* modes 6 and 7: see below */
/* Mode 3 only: shader-clock cycles summed over waves per loop section
 * {refill, top-down sweep, leaf objects, combine, finish-direct, shade, terminate, 0}. */
int srt_pt_section_cycles(srt_pt* pt, uint64_t out[8], int reset);

/* Device time of the dominant kernel of render_epoch[_device] (pt_wave_kernel / pt_unit_kernel / pt_epoch_kernel,
 * whichever the scene selects), measured with HIP events recorded on the launch stream around each launch.
 * Returns the sum over the launches recorded since the previous call (waits for them), then switches
 * recording on (enable != 0) or off.  Off by default. */
int srt_pt_kernel_time(srt_pt* pt, int enable, double* total_ms, uint64_t* launches);

/* Streamed forms (kernel modes 6, 7) only: device time of the kernels of a generation - {logic (the resolve kernel where the
 * generation is split), compaction, ray cast, probe (0 where logic is one kernel)} - summed over every generation launched since
 * the previous call (HIP events around each launch, on the launch stream; waits for them), and the number of generations
 * enqueued; then switches recording on (enable != 0) or off.  Off by default: a diagnostic. */
int srt_pt_stream_times(srt_pt* pt, int enable, double ms_out[4], uint64_t* generations);
/* Streamed forms only: {entries queued to the ray-cast kernel, alive path-slot generations} summed over every generation of every launch since
 * the last reset (device counters of the compaction kernel; waits for the device), and the bytes the forms move through memory per alive
 * slot-generation (saved path state, both logic kernels) and per queued entry (ray planes, list entry, hit) as the kernels' layout has them.
 * bench.py prices the "ray state" term of SURVEY.md 8(d)'s byte figure with these. */
int srt_pt_stream_counters(srt_pt* pt, uint64_t out[4], int reset);
/* Which form render_epoch* takes for the committed scene under the current kernel mode: 0 persistent wave kernel with sweeps,
 * 1 the same with inline BVH<Triangle> walks, 2 persistent waves with the flattened walk, 3 streamed (every ray through the
 * ray-cast kernel), 4 streamed sweeps (BVH<Triangle> walks queued), -1 lane per sample, -2 lane per pixel. */
int srt_pt_kernel_form(srt_pt* pt, int* form);

/* Rays (scene.hit calls) and camera samples traced by this context since the last reset. */
int srt_pt_ray_count(srt_pt* pt, uint64_t* rays, uint64_t* camera_samples, int reset);

/* ---- parity / inspection -------------------------------------------------------------------------- */
/* Dead-ray elision.  In sample_direct_lighting the reference adds the term of the BSDF-sampled
 * direct ray and subtracts it again (student/pathtracer.cpp:118-125): without delta / environment lights the ray cannot
 * change the result of a Lambertian bounce (its random draws are still consumed).  on != 0 lets the kernels skip tracing
 * it where that is provable - no delta or environment light, every continuous BSDF Lambertian: the wave-uniform kernel
 * then runs two-ray batches, the per-lane kernels skip the call; other scenes (and the flattened-walk and stamped
 * diagnostic builds) ignore the switch.  The image is bit-identical either way.
 * srt_pt_ray_count keeps counting the rays the REFERENCE issues; srt_pt_rays_elided reports how many of them were not
 * traced.  Default: off (every ray traced). */
int srt_pt_set_elision(srt_pt* pt, int on);
int srt_pt_rays_elided(srt_pt* pt, uint64_t* elided, int reset);

/* trace_pixel for explicit (x, y, sample) triples (host arrays of n).  rgb_out: 3 floats per sample;
 * draws_out / rays_out (nullable): RNG draws and scene.hit calls of that sample. */
int srt_pt_trace_samples(srt_pt* pt, uint64_t seed, const uint32_t* xs, const uint32_t* ys, const uint32_t* ss,
                         size_t n, float* rgb_out, uint32_t* draws_out, uint32_t* rays_out);
/* scene.hit for explicit rays.  out9: {hit, distance, position[3], normal[3], material} per ray (Trace).  Directions need
 * not be normalised (the particle step passes velocities): times, distances and dist_bounds follow lib/ray.h and
 * student/bvh.inl exactly as the reference scales them. */
int srt_pt_hit(srt_pt* pt, const float* origins, const float* dirs, const float* bounds, size_t n, float* out9);
/* Scene_Particles::Particle::update (student/particles.cpp:5-59) for n particles against the committed scene - the loop body of
 * Scene_Particles::step2 (scene/particles.cpp:134-138), the second caller of Object::hit besides the path tracer (the scene
 * it collides with is Simulate::build_scene's BVH<Object>, gui/simulate.cpp:69-99: commit the same objects).  pos / vel: 3
 * floats per particle, age: 1, updated in place; alive[k] = update()'s return value (age > 0).  dt is one simulation step
 * (Options::dt), radius the particle radius times Options::scale.  Spawning and removing particles stay with the caller.
 * The reference's loop never returns when its hit_time stays <= 0; the kernel gives such a particle up after 4096 legs.
 * Host form: synchronous.  Device form: device pointers, enqueued on `stream` (NULL = the HIP default stream). */
int srt_pt_particles_step(srt_pt* pt, float* pos, float* vel, float* age, size_t n, float dt, float radius, uint8_t* alive);
int srt_pt_particles_step_device(srt_pt* pt, void* stream, float* d_pos, float* d_vel, float* d_age, size_t n, float dt, float radius,
                                 uint8_t* d_alive);

/* Host-side BVH arrays after commit.  which = -1: the BVH<Object> (order = 1-based insertion index of the
 * objects in BVH primitive order); which >= 0: the BVH<Triangle> of the which-th BVH<Object> primitive
 * (order = first vertex index of each triangle in BVH primitive order).  boxes: 6 floats per node,
 * links: {start, size, l, r} per node.  Returns the node count or a negative status. */
long srt_pt_dump_bvh(srt_pt* pt, int which, float* boxes, uint32_t* links, size_t cap, uint32_t* order);
/* Tone mapping for display: HDR_Image::tonemap_to (util/hdr_image.cpp:161-187) with Spectrum::to_srgb
 * (lib/spectrum.h:61-75).  rgb: height*width*3 floats, row 0 first (the accumulated radiance, as srt_pt_render_epoch
 * returns it); rgba_out: height*width*4 bytes, rows flipped as the reference flips them for display, per channel
 * (unsigned char)round(to_srgb(1 - exp(-c * exposure)) * 255), alpha 255.  exposure must be positive (the reference
 * substitutes the image's own exposure for e <= 0; the caller passes that value).  Bit-identical to the reference built
 * against glibc 2.35 on an x86-64 host with FMA.  The _device form takes device pointers (rgba 4-byte aligned) and
 * enqueues on `stream` exactly as the other *_device calls do (hipStream_t; NULL = the HIP default stream) without
 * synchronising: a tone map enqueued behind srt_pt_accumulate_device on the same stream sees that accumulate's result. */
int srt_pt_tonemap(srt_pt* pt, const float* rgb, uint32_t width, uint32_t height, float exposure, uint8_t* rgba_out);
int srt_pt_tonemap_device(srt_pt* pt, void* stream, const float* d_rgb, uint32_t width, uint32_t height, float exposure,
                          uint8_t* d_rgba);

/* Traversal counters of the LAST srt_pt_trace_samples call (an instrumented launch):
 * {rays, box_tests, objects_entered, tri_tests, sphere_tests, tlas_nodes, blas_nodes, light_tri_tests}. */
int srt_pt_counters(srt_pt* pt, uint64_t out[8]);
/* cosf/sinf of the kernel (SRT-MATH v2) for n host floats; parity tests compare them with glibc. */
int srt_pt_math_cos_sin(srt_pt* pt, const float* x, size_t n, float* cos_out, float* sin_out);
/* The kernels' atan2f (glibc 2.35's algorithm restated; Spot_Light::sample) evaluated on the device. */
int srt_pt_math_atan2(srt_pt* pt, const float* y, const float* x, size_t n, float* out);
/* The kernels' acosf (glibc 2.35's algorithm restated; Samplers::Hemisphere::Uniform) evaluated on the device. */
int srt_pt_math_acos(srt_pt* pt, const float* x, size_t n, float* out);

/* The epilogue's expf / powf (glibc 2.35's algorithms restated, FMA build) evaluated on the device. */
int srt_pt_math_exp(srt_pt* pt, const float* x, size_t n, float* out);
int srt_pt_math_pow(srt_pt* pt, const float* x, const float* y, size_t n, float* out);
/* The wave kernel's batched IEEE divide / square root (pt_device.h: div3x3, sqrt3) on host operands, called exactly as
 * the batch tests call them: lane i handles operands 3i, 3i+1, 3i+2.  in: five planes of 3*lanes floats (num0, num1,
 * num2, den, x); out: four planes (num0/den, num1/den, num2/den, sqrt(x)).  shared_c2 != 0: a lane's three rays share
 * num2[3i].  Parity tests compare the planes with the host's correctly rounded `/` and sqrtf. */
int srt_pt_math_div_sqrt(srt_pt* pt, const float* in, size_t lanes, int shared_c2, float* out);

/* Waits for the context's own stream.  Like srt_pt_render_epoch and srt_pt_ray_count it returns SRT_ERR_STATE (once) when a
 * streamed launch since the last such call ended with unfinished work units - the epoch image of that launch is invalid. */
int srt_pt_sync(srt_pt* pt);

#ifdef __cplusplus
}
#endif
#endif /* SRT_PT_H */

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
 *   srt_pt_cancel          <- Pathtracer::cancel                 rays/pathtracer.cpp:282-290 (cancel_flag tested per sample, :224)
 *   srt_pt_read_ray_log    <- Pathtracer::log_ray                rays/pathtracer.cpp:191-193, called from student/pathtracer.cpp:148,
 *                             sink Gui::Widget_Render::log_ray   gui/widgets.cpp:625-628
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

/* One ray of the GUI's ray log (srt_pt_read_ray_log below). */
typedef struct srt_pt_logged_ray {
    float point[3];   /* Ray::point: the shading point */
    float dir[3];     /* Ray::dir, normalised by Ray's constructor */
    float t;          /* 5.0f: the length the reference asks the GUI to draw */
    uint32_t pixel;   /* y * width + x */
    uint32_t sample;  /* absolute sample index (sample_base + k), low 28 bits */
    uint32_t bounce;  /* 0 = the camera ray's hit */
} srt_pt_logged_ray;

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
/* Epochs in flight side by side.  The members render on one stream each; a second LANE - another stream per rank, another set of
 * exchange buffers and another image on rank 0 - lets the caller enqueue epoch k + 1 while epoch k runs, so that the tail of one
 * launch (the last paths of a few lanes) is filled by the next launch's blocks (Multi-GPU).  lane 0 is what
 * srt_pt_group_render_epoch[_device] use; lanes 0 .. 3 exist on demand.  *d_image_out stays valid until the next epoch on the
 * same lane. */
int srt_pt_group_render_epoch_lane(srt_pt_group* g, int lane, uint64_t seed, uint32_t sample_base, uint32_t samples, float** d_image_out,
                                   void** stream_out);
/* A whole render on the group with the accumulator kept ON THE DEVICES, every rank folding its own tiles (the running mean is per
 * pixel): no exchange per epoch at all - the ONE gather happens when somebody looks at the image.
 *   srt_pt_group_reset_accumulator   zero every rank's accumulator (a render that does not add samples)
 *   srt_pt_group_render_samples      srt_pt_render_samples_device on every rank, on lane `lane`
 *   srt_pt_group_wait_lane           wait for that lane's launches (then: cancelled? srt_pt_group_cancel_requested)
 *   srt_pt_group_fold                srt_pt_fold_epochs_device on every rank for the launch last rendered on `lane`; folds are ordered
 *                                    among themselves and against srt_pt_group_accumulator_image (events): call them in launch order
 *   srt_pt_group_accumulator_image   every rank's running mean as tile radiance, ONE ncclGather to rank 0 (copies between ranks that
 *                                    share a device), un-tiled: the width * height * 3 image on rank 0's device and its stream
 * Threads: fold and accumulator_image are not reentrant against each other - the caller serialises them (the drop-in class holds
 * its accumulator mutex); render_samples / wait_lane / fold belong to the render thread, accumulator_image may come from another. */
int srt_pt_group_reset_accumulator(srt_pt_group* g);
int srt_pt_group_max_samples_per_launch(srt_pt_group* g, uint32_t* samples);
int srt_pt_group_render_samples(srt_pt_group* g, int lane, uint64_t seed, uint32_t sample_base, uint32_t samples);
int srt_pt_group_wait_lane(srt_pt_group* g, int lane);
int srt_pt_group_fold(srt_pt_group* g, int lane, uint32_t samples_per_epoch, uint32_t position, uint32_t total_samples, uint32_t accumulator_samples);
int srt_pt_group_accumulator_image(srt_pt_group* g, float** d_image_out, void** stream_out);
int srt_pt_group_cancel_requested(srt_pt_group* g);
/* srt_pt_cancel / srt_pt_clear_cancel on every member; a cancelled srt_pt_group_render_epoch returns SRT_CANCELLED. */
int srt_pt_group_cancel(srt_pt_group* g);
int srt_pt_group_clear_cancel(srt_pt_group* g);
/* srt_pt_set_ray_log on every member / the rays logged by the epochs of `lane` on every member, merged in log order (waits for
 * that lane's streams). */
int srt_pt_group_set_ray_log(srt_pt_group* g, uint32_t capacity);
int srt_pt_group_read_ray_log(srt_pt_group* g, int lane, srt_pt_logged_ray* out, size_t cap, size_t* n_out, uint64_t* dropped);
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

/* ---- launches decoupled from the reference's epochs (what the drop-in class renders with) -------------------------------
 * Pathtracer::begin_render cuts a render into epochs of samples_per_epoch = max(1, n / (hardware_concurrency * 10)) samples
 * (rays/pathtracer.cpp:250-256) - one sample per epoch for a 256-spp render on a 32-thread host - and the image is the running
 * mean of the epoch means in epoch order (:195-207), so the epoch size is part of the result.  A GPU launch wants tens of samples
 * per pixel.  These entry points keep both: srt_pt_render_samples_device renders ONE launch of up to
 * srt_pt_max_samples_per_launch samples per pixel of this rank's tiles and leaves every sample's radiance in the stream's
 * sample buffer; srt_pt_fold_epochs_device then replays do_trace's and accumulate's arithmetic over that buffer, epoch by epoch
 * and sample by sample in order - epoch mean = (sum of the valid samples) * (1.0f / count), accumulator += (mean - accumulator) *
 * (1.0f / k) - into the rank's accumulator: bit-identical to one srt_pt_render_epoch + srt_pt_accumulate_device per epoch.
 *   d_accumulator   DEVICE, srt_pt_accumulator_floats floats (per pixel slot: the running mean, and the epoch in progress so that
 *                   an epoch may span launches), zeroed by the caller before a render that does not add samples
 *   position        samples of this render folded before this launch;  total_samples: of the whole render (its last epoch may
 *                   be short);  accumulator_samples: epochs the accumulator held before the render (Add Samples continues it)
 * The fold goes on the same stream as the launch it folds (it reads that stream's sample buffer) and is skipped when the
 * launch was cancelled.  Launches on two streams overlap; their folds must be ordered by the caller (events).
 * srt_pt_accumulator_tiles_device writes the running mean in the tile layout of srt_pt_render_epoch_device (gather, un-tile). */
int srt_pt_max_samples_per_launch(srt_pt* pt, uint32_t* samples);
int srt_pt_accumulator_floats(srt_pt* pt, size_t* nfloats);
int srt_pt_render_samples_device(srt_pt* pt, void* stream, uint64_t seed, uint32_t sample_base, uint32_t samples);
int srt_pt_fold_epochs_device(srt_pt* pt, void* stream, uint32_t samples_per_epoch, uint32_t position, uint32_t total_samples,
                              uint32_t accumulator_samples, float* d_accumulator);
int srt_pt_accumulator_tiles_device(srt_pt* pt, void* stream, const float* d_accumulator, float* d_tiles_out);

/* ---- cancel (Pathtracer::cancel, rays/pathtracer.cpp:282-290) ------------------------------------------------------
 * The reference's do_trace tests cancel_flag after every sample (rays/pathtracer.cpp:224) and drops the epoch.  srt_pt_cancel
 * raises a flag in pinned host memory that the kernels watch: a persistent launch stops handing out work units (one of its
 * waves looks at the flag whenever it fetches work and then drains the unit queue for all the others), the streamed forms end
 * at the next generation, and every launch of the epoch still enqueued returns at its first instruction - an epoch of 2048
 * samples per pixel ends within a few milliseconds.  It is the ONE call that may be made from another host thread while a
 * render call is running (it does nothing but store the flag).  From then on srt_pt_render_epoch returns SRT_CANCELLED (1, not
 * an error; its output is not written) and srt_pt_render_epoch_device enqueues nothing and returns the same; epochs of the
 * *_device forms that were in flight leave unspecified tiles - the caller drops them, as the reference drops its partial
 * epoch.  srt_pt_clear_cancel waits for the device and lowers the flag (before the next render). */
int srt_pt_cancel(srt_pt* pt);
int srt_pt_cancel_requested(srt_pt* pt);   /* 1 while the flag is raised */
int srt_pt_clear_cancel(srt_pt* pt);

/* ---- log_ray (Pathtracer::log_ray -> Gui::Widget_Render::log_ray, gui/widgets.cpp:625-628) ----------------------------
 * sample_direct_lighting flips RNG::coin_flip(0.0005f) at every shading point of a continuous BSDF and, when it comes up,
 * logs the ray it is about to trace toward the light / along the BSDF sample: log_ray(world_ray_task6, 5.0f)
 * (student/pathtracer.cpp:146-148; the GUI draws point .. point + t * dir in white).  The kernels always draw the coin (the RNG
 * ledger needs it); with a ring of `capacity` > 0 rays per stream they also record the selected rays, and srt_pt_read_ray_log
 * hands them to the host - the drop-in class replays them into gui.log_ray after every epoch.  Rays beyond the capacity
 * between two reads are counted in *dropped.  Order: pixel (y * width + x), then sample, then bounce - the order a
 * single-threaded do_trace logs them in.  srt_pt_read_ray_log waits for the device and reads every stream's ring;
 * the _stream form waits for `stream` only and reads the ring of the epochs rendered on it.  out may be NULL: *n_out is then
 * the number of rays waiting, and they stay; otherwise the rings are emptied, and what `cap` does not take counts as dropped. */
int srt_pt_set_ray_log(srt_pt* pt, uint32_t capacity);   /* 0 (default): nothing is recorded */
int srt_pt_read_ray_log(srt_pt* pt, srt_pt_logged_ray* out, size_t cap, size_t* n_out, uint64_t* dropped);
int srt_pt_read_ray_log_stream(srt_pt* pt, void* stream, srt_pt_logged_ray* out, size_t cap, size_t* n_out, uint64_t* dropped);

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

/* (Kernel selection, timing brackets, traversal counters and the device math probes: include/srt_pt_debug.h.) */

/* Waits for the context's own stream.  Like srt_pt_render_epoch and srt_pt_ray_count it returns SRT_ERR_STATE (once) when a
 * streamed launch since the last such call ended with unfinished work units - the epoch image of that launch is invalid. */
int srt_pt_sync(srt_pt* pt);

#ifdef __cplusplus
}
#endif
#endif /* SRT_PT_H */

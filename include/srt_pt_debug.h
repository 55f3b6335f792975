/*
 * srt_pt_debug.h - diagnostic and test entry points of the path-tracer library (libsrt_hip.so).  NOT part of the drop-in
 * boundary (include/srt_pt.h): nothing here is needed to render; the parity tests, bench.py and the profiling tools use them.
 *
 *   kernel selection / timing brackets / counters of the forms that implement Pathtracer::do_trace
 *   (rays/pathtracer.cpp:209-231; student/pathtracer.cpp:14-218, student/bvh.inl:166-276), and the device's restatements of the
 *   libm functions the reference's arithmetic goes through (glibc 2.35), evaluated for host arrays.
 */
#ifndef SRT_PT_DEBUG_H
#define SRT_PT_DEBUG_H

#include "srt_pt.h"

#ifdef __cplusplus
extern "C" {
#endif

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

#ifdef __cplusplus
}
#endif
#endif /* SRT_PT_DEBUG_H */

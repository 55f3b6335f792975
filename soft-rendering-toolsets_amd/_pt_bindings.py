"""ctypes bindings of include/srt_pt.h and a mirror of PT::Pathtracer's public surface
(/root/reference/Assignments/Scotty3D/src/rays/pathtracer.h:24-40).  Plumbing only: every number
is produced by the HIP kernels behind the C ABI."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_float, c_int, c_long, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np


class PtMaterial(ctypes.Structure):
    _fields_ = [("type", c_uint32), ("a", c_float * 3), ("b", c_float * 3), ("ior", c_float)]


# srt_pt_logged_ray (include/srt_pt.h): one call of Pathtracer::log_ray
LOGGED_RAY_DTYPE = np.dtype([("point", np.float32, 3), ("dir", np.float32, 3), ("t", np.float32), ("pixel", np.uint32),
                             ("sample", np.uint32), ("bounce", np.uint32)])
SRT_CANCELLED = 1


class SrtCancelled(Exception):
    """A render call returned SRT_CANCELLED: srt_pt_cancel cut it short, its output was not written."""


COUNTER_NAMES = ("rays", "box_tests", "objects_entered", "tri_tests", "sphere_tests", "tlas_nodes", "blas_nodes",
                 "light_tri_tests")


def bind(lib: ctypes.CDLL) -> None:
    lib.srt_pt_create.argtypes = [c_int, POINTER(c_void_p)]
    lib.srt_pt_create_multi.argtypes = [c_void_p, c_int, POINTER(c_void_p)]
    lib.srt_pt_group_destroy.argtypes = [c_void_p]
    lib.srt_pt_group_size.argtypes = [c_void_p]
    lib.srt_pt_group_context.argtypes = [c_void_p, c_int]
    lib.srt_pt_group_context.restype = c_void_p
    lib.srt_pt_group_uses_rccl.argtypes = [c_void_p]
    lib.srt_pt_group_set_params.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32]
    lib.srt_pt_group_render_epoch.argtypes = [c_void_p, c_uint64, c_uint32, c_uint32, c_void_p]
    lib.srt_pt_group_render_epoch_device.argtypes = [c_void_p, c_uint64, c_uint32, c_uint32, POINTER(c_void_p), POINTER(c_void_p)]
    lib.srt_pt_group_gather_time.argtypes = [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(c_uint64)]
    lib.srt_pt_destroy.argtypes = [c_void_p]
    lib.srt_pt_scene_begin.argtypes = [c_void_p]
    lib.srt_pt_add_material.argtypes = [c_void_p, POINTER(PtMaterial), POINTER(c_uint32)]
    lib.srt_pt_add_mesh.argtypes = [c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_uint32, c_void_p, c_uint32, c_int]
    lib.srt_pt_set_env_light.argtypes = [c_void_p, c_uint32, c_void_p]
    lib.srt_pt_set_env_map.argtypes = [c_void_p, c_uint32, c_uint32, c_void_p]
    lib.srt_pt_add_sphere_light.argtypes = [c_void_p, c_float, c_void_p, c_uint32, c_void_p, c_void_p, c_uint32, c_void_p, c_uint32]
    lib.srt_pt_add_light.argtypes = [c_void_p, c_uint32, c_void_p, c_void_p, c_void_p]
    lib.srt_pt_add_sphere.argtypes = [c_void_p, c_float, c_void_p, c_uint32]
    lib.srt_pt_scene_commit.argtypes = [c_void_p, c_int]
    lib.srt_pt_set_bvh_builder.argtypes = [c_void_p, c_int, c_uint32]
    lib.srt_pt_set_stream_slots.argtypes = [c_void_p, c_uint32]
    lib.srt_pt_set_camera.argtypes = [c_void_p, c_void_p, c_float, c_float]
    lib.srt_pt_set_params.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32]
    lib.srt_pt_set_tiling.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32, c_uint32]
    lib.srt_pt_tile_info.argtypes = [c_void_p, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]
    lib.srt_pt_render_epoch.argtypes = [c_void_p, c_uint64, c_uint32, c_uint32, c_void_p]
    lib.srt_pt_render_epoch_device.argtypes = [c_void_p, c_void_p, c_uint64, c_uint32, c_uint32, c_void_p]
    lib.srt_pt_untile_device.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
    lib.srt_pt_accumulate_device.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_uint32]
    lib.srt_pt_set_kernel.argtypes = [c_void_p, c_int]
    lib.srt_pt_section_cycles.argtypes = [c_void_p, c_void_p, c_int]
    lib.srt_pt_kernel_time.argtypes = [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(c_uint64)]
    lib.srt_pt_ray_count.argtypes = [c_void_p, POINTER(c_uint64), POINTER(c_uint64), c_int]
    lib.srt_pt_stream_times.argtypes = [c_void_p, c_int, c_void_p, POINTER(c_uint64)]
    lib.srt_pt_stream_counters.argtypes = [c_void_p, c_void_p, c_int]
    lib.srt_pt_kernel_form.argtypes = [c_void_p, POINTER(c_int)]
    lib.srt_pt_trace_samples.argtypes = [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]
    lib.srt_pt_hit.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.srt_pt_particles_step.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p]
    lib.srt_pt_particles_step_device.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p]
    lib.srt_pt_dump_bvh.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.srt_pt_dump_bvh.restype = c_long
    lib.srt_pt_counters.argtypes = [c_void_p, c_void_p]
    lib.srt_pt_math_cos_sin.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]
    lib.srt_pt_math_acos.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    lib.srt_pt_math_atan2.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.srt_pt_set_elision.argtypes = [c_void_p, c_int]
    lib.srt_pt_rays_elided.argtypes = [c_void_p, POINTER(c_uint64), c_int]
    lib.srt_pt_math_exp.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    lib.srt_pt_math_pow.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.srt_pt_tonemap.argtypes = [c_void_p, c_void_p, c_uint32, c_uint32, c_float, c_void_p]
    lib.srt_pt_tonemap_device.argtypes = [c_void_p, c_void_p, c_void_p, c_uint32, c_uint32, c_float, c_void_p]
    lib.srt_pt_math_div_sqrt.argtypes = [c_void_p, c_void_p, c_size_t, ctypes.c_int, c_void_p]
    lib.srt_pt_sync.argtypes = [c_void_p]
    lib.srt_pt_cancel.argtypes = [c_void_p]
    lib.srt_pt_cancel_requested.argtypes = [c_void_p]
    lib.srt_pt_clear_cancel.argtypes = [c_void_p]
    lib.srt_pt_set_ray_log.argtypes = [c_void_p, c_uint32]
    lib.srt_pt_read_ray_log.argtypes = [c_void_p, c_void_p, c_size_t, POINTER(c_size_t), POINTER(c_uint64)]
    lib.srt_pt_read_ray_log_stream.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, POINTER(c_size_t), POINTER(c_uint64)]
    lib.srt_pt_group_render_epoch_lane.argtypes = [c_void_p, c_int, c_uint64, c_uint32, c_uint32, POINTER(c_void_p), POINTER(c_void_p)]
    lib.srt_pt_group_cancel.argtypes = [c_void_p]
    lib.srt_pt_group_clear_cancel.argtypes = [c_void_p]
    lib.srt_pt_group_set_ray_log.argtypes = [c_void_p, c_uint32]
    lib.srt_pt_group_read_ray_log.argtypes = [c_void_p, c_int, c_void_p, c_size_t, POINTER(c_size_t), POINTER(c_uint64)]


def _p(a):
    return a.ctypes.data_as(c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Scene:
    """A scene description (see scenes.py): materials, objects, camera — the inputs of build_scene."""

    def __init__(self, description: dict):
        self.description = description


class Pathtracer:
    """Mirror of PT::Pathtracer for the HIP path.

    Reference call sequence (gui/widgets.cpp:921-967): set_params(w, h, samples, depth, use_bvh) ->
    begin_render(scene, camera) -> poll in_progress()/progress() -> get_output().  Here begin_render is
    synchronous (the C++ drop-in keeps the reference's asynchronous worker); the epoch
    scheme is the reference's: samples_per_epoch = max(1, n / (n_threads * 10)), running mean of epoch
    means (rays/pathtracer.cpp:250-280, 195-207).

    device = -1 creates a host-only context: scene assembly and BVH inspection work, rendering raises.
    """

    def __init__(self, device: int = 0, n_threads: int | None = None, _borrowed_ctx=None):
        from . import SrtError, _check, load_library

        self._SrtError, self._check = SrtError, _check
        self._lib = load_library()
        self._ctx = c_void_p()
        self._borrowed = _borrowed_ctx is not None       # a member context of a PathtracerGroup: the group destroys it
        if self._borrowed:
            self._ctx = c_void_p(_borrowed_ctx)
        else:
            _check(self._lib, self._lib.srt_pt_create(device, ctypes.byref(self._ctx)))
        self.n_threads = n_threads or os.cpu_count() or 1
        self.out_w = self.out_h = 0
        self.n_samples = 0
        self.max_depth = 8
        self.scene_use_bvh = True
        self.accumulator = None
        self.accumulator_samples = 0
        self.total_epochs = self.completed_epochs = 0
        self.seed = 0
        self._sample_cursor = 0

    def close(self) -> None:
        if self._ctx:
            if not self._borrowed:
                self._lib.srt_pt_destroy(self._ctx)
            self._ctx = c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- reference surface -------------------------------------------------------------------------
    def set_params(self, w: int, h: int, pixel_samples: int, depth: int, use_bvh: bool) -> None:
        self.out_w, self.out_h, self.n_samples, self.max_depth = int(w), int(h), int(pixel_samples), int(depth)
        self.scene_use_bvh = bool(use_bvh)
        self.accumulator = np.zeros((self.out_h, self.out_w, 3), np.float32)
        self._check(self._lib, self._lib.srt_pt_set_params(self._ctx, self.out_w, self.out_h, self.max_depth))

    def set_samples(self, samples: int) -> None:
        self.n_samples = int(samples)

    def build_scene(self, scene) -> None:
        d = scene.description if isinstance(scene, Scene) else scene
        L = self._lib
        self._check(L, L.srt_pt_scene_begin(self._ctx))
        for m in d["materials"]:
            pm = PtMaterial(int(m["type"]), (c_float * 3)(*[float(v) for v in m["a"]]), (c_float * 3)(*[float(v) for v in m["b"]]),
                            float(m["ior"]))
            self._check(L, L.srt_pt_add_material(self._ctx, ctypes.byref(pm), None))
        for o in d["objects"]:
            T = _f32(o["T"])
            if o["kind"] == "mesh":
                pos, nrm = _f32(o["pos"]), _f32(o["nrm"])
                idx = np.ascontiguousarray(o["idx"], np.uint32)
                self._check(L, L.srt_pt_add_mesh(self._ctx, _p(pos), _p(nrm), len(pos), _p(idx), len(idx), _p(T),
                                                 int(o["material"]), int(bool(o["is_light"]))))
            elif o.get("light_mesh") is not None:   # emissive sphere
                lm = o["light_mesh"]
                pos, nrm, idx = _f32(lm["pos"]), _f32(lm["nrm"]), np.ascontiguousarray(lm["idx"], np.uint32)
                self._check(L, L.srt_pt_add_sphere_light(self._ctx, float(o["radius"]), _p(T), int(o["material"]), _p(pos), _p(nrm),
                                                         len(pos), _p(idx), len(idx)))
            else:
                self._check(L, L.srt_pt_add_sphere(self._ctx, float(o["radius"]), _p(T), int(o["material"])))
        if d.get("env"):               # {"type": 1 sphere | 2 hemisphere, "radiance"} or {"type": 3, "image": float32 [h, w, 3]}
            if int(d["env"]["type"]) == 3:
                img = _f32(d["env"]["image"])
                self._check(L, L.srt_pt_set_env_map(self._ctx, img.shape[1], img.shape[0], _p(img)))
            else:
                rad = _f32(d["env"]["radiance"])
                self._check(L, L.srt_pt_set_env_light(self._ctx, int(d["env"]["type"]), _p(rad)))
        for l in d.get("lights", []):   # delta lights (Pathtracer::build_lights): type 0 directional, 1 point, 2 spot
            rad, ab, T = _f32(l["radiance"]), _f32(l.get("angle_bounds", (0.0, 0.0))), _f32(l["T"])
            self._check(L, L.srt_pt_add_light(self._ctx, int(l["type"]), _p(rad), _p(ab), _p(T)))
        self._check(L, L.srt_pt_scene_commit(self._ctx, int(self.scene_use_bvh)))

    def set_camera(self, camera: dict) -> None:
        iv = _f32(camera["iview"])
        self._check(self._lib, self._lib.srt_pt_set_camera(self._ctx, _p(iv), float(camera["vfov"]), float(camera["ar"])))

    def begin_render(self, scene, camera: dict | None = None, add_samples: bool = False, samples_per_epoch: int | None = None) -> None:
        d = scene.description if isinstance(scene, Scene) else scene
        spe = samples_per_epoch or max(1, self.n_samples // (self.n_threads * 10))
        self.total_epochs = self.n_samples // spe + (1 if self.n_samples % spe else 0)
        self.completed_epochs = 0
        if not add_samples:
            self.accumulator[...] = 0
            self.accumulator_samples = 0
            self._sample_cursor = 0
            self.build_scene(d)
        self.set_camera(camera or d["camera"])
        s = 0
        while s < self.n_samples:
            n = min(spe, self.n_samples - s)
            epoch = self.render_epoch(self.seed, self._sample_cursor, n)
            self.accumulate(epoch)
            self._sample_cursor += n
            self.completed_epochs += 1
            s += n

    def accumulate(self, sample: np.ndarray) -> None:
        """rays/pathtracer.cpp:195-207 on the host copy (the device form is srt_pt_accumulate_device)."""
        self.accumulator_samples += 1
        inv = np.float32(1.0) / np.float32(self.accumulator_samples)
        self.accumulator += ((sample - self.accumulator).astype(np.float32) * inv).astype(np.float32)

    def in_progress(self) -> bool:
        return self.completed_epochs < self.total_epochs

    def progress(self) -> float:
        return self.completed_epochs / self.total_epochs if self.total_epochs else 0.0

    def get_output(self) -> np.ndarray:
        return self.accumulator

    def cancel(self) -> None:
        self.completed_epochs = self.total_epochs = 0

    # -- C-ABI steps ---------------------------------------------------------------------------------
    def set_tiling(self, tile_w: int, tile_h: int, rank: int, world: int) -> None:
        self._check(self._lib, self._lib.srt_pt_set_tiling(self._ctx, tile_w, tile_h, rank, world))

    def tile_info(self):
        a, b, c = c_uint32(), c_uint32(), c_uint32()
        self._check(self._lib, self._lib.srt_pt_tile_info(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def render_epoch(self, seed: int, sample_base: int, samples: int, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.zeros((self.out_h, self.out_w, 3), np.float32)
        st = self._lib.srt_pt_render_epoch(self._ctx, seed, sample_base, samples, _p(out))
        if st == SRT_CANCELLED:
            raise SrtCancelled()
        self._check(self._lib, st)
        return out

    def render_epoch_device(self, stream: int, seed: int, sample_base: int, samples: int, d_tiles_out: int) -> None:
        st = self._lib.srt_pt_render_epoch_device(self._ctx, c_void_p(stream), seed, sample_base, samples, c_void_p(d_tiles_out))
        if st == SRT_CANCELLED:
            raise SrtCancelled()
        self._check(self._lib, st)

    # -- Pathtracer::cancel / Pathtracer::log_ray on the C ABI ------------------------------------------
    def cancel_device(self) -> None:
        """srt_pt_cancel: may be called from another thread while render_epoch runs."""
        self._check(self._lib, self._lib.srt_pt_cancel(self._ctx))

    def cancel_requested(self) -> bool:
        return bool(self._lib.srt_pt_cancel_requested(self._ctx))

    def clear_cancel(self) -> None:
        self._check(self._lib, self._lib.srt_pt_clear_cancel(self._ctx))

    def set_ray_log(self, capacity: int) -> None:
        self._check(self._lib, self._lib.srt_pt_set_ray_log(self._ctx, int(capacity)))

    def read_ray_log(self, stream: int | None = None):
        """(rays as a LOGGED_RAY_DTYPE array in log order, dropped): what Pathtracer::log_ray received since the last read."""
        n, dropped = c_size_t(), c_uint64()
        if stream is None:
            self._check(self._lib, self._lib.srt_pt_read_ray_log(self._ctx, None, 0, ctypes.byref(n), ctypes.byref(dropped)))
        else:
            self._check(self._lib, self._lib.srt_pt_read_ray_log_stream(self._ctx, c_void_p(stream), None, 0, ctypes.byref(n), ctypes.byref(dropped)))
        out = np.zeros(n.value, LOGGED_RAY_DTYPE)
        got = c_size_t()
        if stream is None:
            self._check(self._lib, self._lib.srt_pt_read_ray_log(self._ctx, _p(out) if n.value else None, n.value, ctypes.byref(got), ctypes.byref(dropped)))
        else:
            self._check(self._lib, self._lib.srt_pt_read_ray_log_stream(self._ctx, c_void_p(stream), _p(out) if n.value else None, n.value,
                                                                        ctypes.byref(got), ctypes.byref(dropped)))
        return out[:got.value], int(dropped.value)

    def untile_device(self, stream: int, d_gathered: int, d_image: int) -> None:
        self._check(self._lib, self._lib.srt_pt_untile_device(self._ctx, c_void_p(stream), c_void_p(d_gathered), c_void_p(d_image)))

    def accumulate_device(self, stream: int, d_acc: int, d_epoch: int, nfloats: int, k: int) -> None:
        self._check(self._lib, self._lib.srt_pt_accumulate_device(self._ctx, c_void_p(stream), c_void_p(d_acc), c_void_p(d_epoch), nfloats, k))

    def set_stream_slots(self, slots: int) -> None:
        """Paths in flight per launch of the streamed forms (0 = default); the image does not depend on it."""
        self._check(self._lib, self._lib.srt_pt_set_stream_slots(self._ctx, int(slots)))

    def set_bvh_builder(self, device: bool, min_primitives: int = 16384) -> None:
        """Where build_scene runs BVH::build: on the GPU for primitive sets of at least `min_primitives`, else on the host."""
        self._check(self._lib, self._lib.srt_pt_set_bvh_builder(self._ctx, int(bool(device)), int(min_primitives)))

    def set_kernel(self, mode: int) -> None:
        """0 auto, 1 lane per pixel, 2 wave-uniform sweeps, 3 the same with section stamps, 4 lane per sample,
        5 persistent waves with the flattened per-lane walk, 6 streamed form: logic + ray-cast kernels (include/srt_pt.h)."""
        self._check(self._lib, self._lib.srt_pt_set_kernel(self._ctx, int(mode)))

    def section_cycles(self, reset: bool = False) -> dict:
        out = np.zeros(8, np.uint64)
        self._check(self._lib, self._lib.srt_pt_section_cycles(self._ctx, _p(out), int(reset)))
        names = ("refill", "top_down", "leaf_objects", "combine", "finish_direct", "shade", "terminate")
        return dict(zip(names, (int(v) for v in out)))

    def kernel_time(self, enable: bool = True):
        """(total_ms, launches) of the dominant kernel since the previous call (HIP events on the launch stream,
        recorded inside the library); then switches recording on/off."""
        ms, n = ctypes.c_double(), c_uint64()
        self._check(self._lib, self._lib.srt_pt_kernel_time(self._ctx, int(enable), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def stream_times(self, enable: bool = True):
        """({"logic_ms", "compact_ms", "cast_ms", "probe_ms"}, generations) of the streamed forms since the previous call; then
        recording on/off.  logic_ms is the resolve kernel where a generation's logic is split in two (probe_ms is 0 otherwise)."""
        ms = np.zeros(4, np.float64)
        g = c_uint64()
        self._check(self._lib, self._lib.srt_pt_stream_times(self._ctx, int(enable), _p(ms), ctypes.byref(g)))
        return {"logic_ms": float(ms[0]), "compact_ms": float(ms[1]), "cast_ms": float(ms[2]), "probe_ms": float(ms[3])}, int(g.value)

    def stream_counters(self, reset: bool = False) -> dict:
        """Streamed forms: entries queued to the ray-cast kernel and alive path-slot generations since the last reset, and the bytes per unit."""
        out = np.zeros(4, np.uint64)
        self._check(self._lib, self._lib.srt_pt_stream_counters(self._ctx, _p(out), int(reset)))
        return {"entries_queued": int(out[0]), "alive_slot_generations": int(out[1]), "bytes_per_alive_slot_generation": int(out[2]),
                "bytes_per_queued_entry": int(out[3])}

    def kernel_form(self) -> int:
        """Form render_epoch takes for the committed scene: 0 / 1 persistent sweeps (1: inline mesh walks), 2 flattened walk,
        3 streamed, 4 streamed sweeps, -1 lane per sample, -2 lane per pixel."""
        f = c_int()
        self._check(self._lib, self._lib.srt_pt_kernel_form(self._ctx, ctypes.byref(f)))
        return int(f.value)

    def ray_count(self, reset: bool = False):
        """(rays, camera_samples) traced by render_epoch* since the last reset (synchronizes the device)."""
        r, c = c_uint64(), c_uint64()
        self._check(self._lib, self._lib.srt_pt_ray_count(self._ctx, ctypes.byref(r), ctypes.byref(c), int(reset)))
        return r.value, c.value

    def trace_samples(self, seed: int, xs, ys, ss):
        xs, ys, ss = (np.ascontiguousarray(a, np.uint32) for a in (xs, ys, ss))
        rgb = np.zeros((len(xs), 3), np.float32)
        draws = np.zeros(len(xs), np.uint32)
        rays = np.zeros(len(xs), np.uint32)
        self._check(self._lib, self._lib.srt_pt_trace_samples(self._ctx, seed, _p(xs), _p(ys), _p(ss), len(xs), _p(rgb), _p(draws), _p(rays)))
        return rgb, draws, rays

    def counters(self) -> dict:
        out = np.zeros(8, np.uint64)
        self._check(self._lib, self._lib.srt_pt_counters(self._ctx, _p(out)))
        return dict(zip(COUNTER_NAMES, (int(v) for v in out)))

    def hit(self, org, dirs, bounds) -> np.ndarray:
        org, dirs, bounds = _f32(org), _f32(dirs), _f32(bounds)
        out = np.zeros((len(org), 9), np.float32)
        self._check(self._lib, self._lib.srt_pt_hit(self._ctx, _p(org), _p(dirs), _p(bounds), len(org), _p(out)))
        return out

    def particles_step(self, pos, vel, age, dt: float, radius: float):
        """Scene_Particles::Particle::update for every particle (scene/particles.cpp:134-138): returns (pos, vel, age, alive)."""
        pos, vel, age = _f32(pos).copy(), _f32(vel).copy(), _f32(age).copy()
        alive = np.zeros(len(age), np.uint8)
        self._check(self._lib, self._lib.srt_pt_particles_step(self._ctx, _p(pos), _p(vel), _p(age), len(age), float(dt), float(radius), _p(alive)))
        return pos, vel, age, alive

    def dump_bvh(self, which: int, cap: int = 1 << 22):
        boxes = np.zeros((cap, 6), np.float32)
        links = np.zeros((cap, 4), np.uint32)
        order = np.zeros(cap * 4, np.uint32)
        n = self._lib.srt_pt_dump_bvh(self._ctx, which, _p(boxes), _p(links), cap, _p(order))
        if n < 0:
            raise self._SrtError(int(n), self._lib.srt_last_error().decode())
        return boxes[:n].copy(), links[:n].copy(), order

    def math_acos(self, x):
        x = _f32(x)
        out = np.zeros(len(x), np.float32)
        self._check(self._lib, self._lib.srt_pt_math_acos(self._ctx, _p(x), len(x), _p(out)))
        return out

    def math_atan2(self, y, x):
        y, x = _f32(y), _f32(x)
        out = np.zeros(len(y), np.float32)
        self._check(self._lib, self._lib.srt_pt_math_atan2(self._ctx, _p(y), _p(x), len(y), _p(out)))
        return out

    def set_elision(self, on: bool) -> None:
        """Let the wave kernel skip the provably dead BSDF-sampled direct ray (include/srt_pt.h); images stay bit-identical."""
        self._check(self._lib, self._lib.srt_pt_set_elision(self._ctx, int(bool(on))))

    def rays_elided(self, reset: bool = False) -> int:
        n = c_uint64(0)
        self._check(self._lib, self._lib.srt_pt_rays_elided(self._ctx, ctypes.byref(n), int(reset)))
        return int(n.value)

    def tonemap(self, rgb, exposure: float = 1.0) -> np.ndarray:
        """HDR_Image::tonemap_to: (h, w, 3) float radiance -> (h, w, 4) uint8 sRGB, rows flipped for display."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w = rgb.shape[:2]
        out = np.zeros((h, w, 4), np.uint8)
        self._check(self._lib, self._lib.srt_pt_tonemap(self._ctx, _p(rgb), w, h, float(exposure), _p(out)))
        return out

    def tonemap_device(self, d_rgb_ptr: int, w: int, h: int, exposure: float, d_rgba_ptr: int, stream: int = 0) -> None:
        self._check(self._lib, self._lib.srt_pt_tonemap_device(self._ctx, c_void_p(stream), d_rgb_ptr, w, h, float(exposure), d_rgba_ptr))

    def math_exp(self, x):
        x = _f32(x)
        out = np.zeros(len(x), np.float32)
        self._check(self._lib, self._lib.srt_pt_math_exp(self._ctx, _p(x), len(x), _p(out)))
        return out

    def math_pow(self, x, y):
        x, y = _f32(x), _f32(y)
        out = np.zeros(len(x), np.float32)
        self._check(self._lib, self._lib.srt_pt_math_pow(self._ctx, _p(x), _p(y), len(x), _p(out)))
        return out

    def math_div_sqrt(self, num0, num1, num2, den, x, shared_c2=False):
        """(num0/den, num1/den, num2/den, sqrt(x)) through the wave kernel's div3x3 / sqrt3; len % 3 == 0."""
        planes = np.ascontiguousarray(np.stack([_f32(num0), _f32(num1), _f32(num2), _f32(den), _f32(x)]))
        n3 = planes.shape[1]
        assert n3 % 3 == 0
        out = np.zeros((4, n3), np.float32)
        self._check(self._lib, self._lib.srt_pt_math_div_sqrt(self._ctx, _p(planes), n3 // 3, int(bool(shared_c2)), _p(out)))
        return out[0], out[1], out[2], out[3]

    def math_cos_sin(self, x):
        x = _f32(x)
        c, s = np.zeros_like(x), np.zeros_like(x)
        self._check(self._lib, self._lib.srt_pt_math_cos_sin(self._ctx, _p(x), len(x), _p(c), _p(s)))
        return c, s

    def sync(self) -> None:
        self._check(self._lib, self._lib.srt_pt_sync(self._ctx))


class PathtracerGroup:
    """srt_pt_create_multi: one render sharded by image tile over several devices inside one process (include/srt_pt.h).
    members[r] is a Pathtracer over rank r's context (scene / camera / kernel calls are made on every member)."""

    def __init__(self, devices):
        from . import SrtError, _check, load_library

        self._SrtError, self._check = SrtError, _check
        self._lib = load_library()
        self._g = c_void_p()
        devs = (c_int * len(devices))(*[int(d) for d in devices])
        _check(self._lib, self._lib.srt_pt_create_multi(devs, len(devices), ctypes.byref(self._g)))
        self.members = [Pathtracer(_borrowed_ctx=self._lib.srt_pt_group_context(self._g, r)) for r in range(len(devices))]
        self.out_w = self.out_h = 0

    def uses_rccl(self) -> bool:
        return bool(self._lib.srt_pt_group_uses_rccl(self._g))

    def set_params(self, w: int, h: int, pixel_samples: int, depth: int, use_bvh: bool) -> None:
        for m in self.members:
            m.out_w, m.out_h, m.n_samples, m.max_depth, m.scene_use_bvh = int(w), int(h), int(pixel_samples), int(depth), bool(use_bvh)
        self.out_w, self.out_h = int(w), int(h)
        self._check(self._lib, self._lib.srt_pt_group_set_params(self._g, int(w), int(h), int(depth)))

    def build_scene(self, scene) -> None:
        for m in self.members:
            m.build_scene(scene)

    def set_camera(self, camera) -> None:
        for m in self.members:
            m.set_camera(camera)

    def set_kernel(self, mode: int) -> None:
        for m in self.members:
            m.set_kernel(mode)

    def set_elision(self, on: bool) -> None:
        for m in self.members:
            m.set_elision(on)

    def render_epoch(self, seed: int, sample_base: int, samples: int) -> np.ndarray:
        out = np.zeros((self.out_h, self.out_w, 3), np.float32)
        self._check(self._lib, self._lib.srt_pt_group_render_epoch(self._g, seed, sample_base, samples, _p(out)))
        return out

    def render_epoch_device(self, seed: int, sample_base: int, samples: int):
        """Enqueue one epoch on every rank + the gather; returns (device pointer of the image on rank 0, its stream). No wait."""
        d, s = c_void_p(), c_void_p()
        self._check(self._lib, self._lib.srt_pt_group_render_epoch_device(self._g, seed, sample_base, samples, ctypes.byref(d), ctypes.byref(s)))
        return d.value, s.value

    def render_epoch_lane(self, lane: int, seed: int, sample_base: int, samples: int):
        """As render_epoch_device on lane `lane` (its own streams and buffers): epochs on different lanes overlap."""
        d, s = c_void_p(), c_void_p()
        st = self._lib.srt_pt_group_render_epoch_lane(self._g, int(lane), seed, sample_base, samples, ctypes.byref(d), ctypes.byref(s))
        if st == SRT_CANCELLED:
            raise SrtCancelled()
        self._check(self._lib, st)
        return d.value, s.value

    def cancel_device(self) -> None:
        self._check(self._lib, self._lib.srt_pt_group_cancel(self._g))

    def clear_cancel(self) -> None:
        self._check(self._lib, self._lib.srt_pt_group_clear_cancel(self._g))

    def set_ray_log(self, capacity: int) -> None:
        self._check(self._lib, self._lib.srt_pt_group_set_ray_log(self._g, int(capacity)))

    def read_ray_log(self, lane: int = 0):
        n, dropped = c_size_t(), c_uint64()
        self._check(self._lib, self._lib.srt_pt_group_read_ray_log(self._g, int(lane), None, 0, ctypes.byref(n), ctypes.byref(dropped)))
        out = np.zeros(n.value, LOGGED_RAY_DTYPE)
        got = c_size_t()
        self._check(self._lib, self._lib.srt_pt_group_read_ray_log(self._g, int(lane), _p(out) if n.value else None, n.value, ctypes.byref(got),
                                                                     ctypes.byref(dropped)))
        return out[:got.value], int(dropped.value)

    def gather_time(self, enable: bool = True):
        """(total ms, epochs) of the exchange step (gather + un-tiling, incl. waiting for the slowest rank) since the previous call."""
        ms, n = ctypes.c_double(), c_uint64()
        self._check(self._lib, self._lib.srt_pt_group_gather_time(self._g, int(enable), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def ray_count(self, reset: bool = False):
        r = [m.ray_count(reset) for m in self.members]
        return sum(x[0] for x in r), sum(x[1] for x in r)

    def close(self) -> None:
        if self._g:
            for m in self.members:
                m.close()
            self._lib.srt_pt_group_destroy(self._g)
            self._g = c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

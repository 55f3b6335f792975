"""soft-rendering-toolsets_amd — MI355X-native hot paths of Soft-Rendering-Toolsets.

Python here is plumbing only: it loads the C-ABI shared library built from
``csrc/`` (hand-written HIP for gfx950) and mirrors the reference's two class
surfaces so the parity tests read like the reference's own call sequences:

* :class:`SoftwareRenderer`  <- CMU462::SoftwareRenderer
  (/root/reference/Assignments/DrawSVG/src/software_renderer.h:25-98)
* :class:`Pathtracer`        <- PT::Pathtracer
  (/root/reference/Assignments/Scotty3D/src/rays/pathtracer.h:24-40)

There is NO CPU fallback: if ``lib/libsrt_hip.so`` is missing, or no HIP device
is present when a context is created, the calls raise.

The directory name contains a hyphen, so import it through ``srt_amd.py`` at the
repository root (``import srt_amd``) or with importlib.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRT_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libsrt_hip.so")  # env override: A/B builds

# One record of the ordered primitive stream (include/srt_raster.h: srt_prim, 48 bytes).
PRIM_DTYPE = np.dtype(
    {
        "names": ["kind", "reserved", "v", "rgba"],
        "formats": ["<u4", "<u4", ("<f8", (3,)), ("<f4", (4,))],
        "offsets": [0, 4, 8, 32],
        "itemsize": 48,
    }
)
PRIM_TRIANGLE = 1
PRIM_POINT = 2
PRIM_IMAGE = 3
PRIM_LINE = 4


class SrtError(RuntimeError):
    """A C-ABI entry point returned a negative status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"srt status {status}: {message}")
        self.status = status


class RasterStats(ctypes.Structure):
    _fields_ = [
        ("sample_tests", c_uint64),
        ("sample_tests_in_target", c_uint64),
        ("fragments", c_uint64),
        ("point_samples", c_uint64),
        ("bin_entries", c_uint64),
        ("list_bytes", c_uint64),
    ]


_lib = None


def load_library() -> ctypes.CDLL:
    """Load libsrt_hip.so (built by ``__graft_entry__.build()`` / ``make -C csrc``). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64 and
    # loads them by absolute path; if the system copy (our DT_NEEDED) were loaded first, torch would bring
    # up a second HSA runtime that sees no GPU.  Importing torch first makes the loader resolve our
    # libamdhip64.so.7 to the copy already in the process, so tensors, streams and RCCL share our device.
    try:
        import torch  # noqa: F401
    except ImportError:  # the C ABI itself has no torch dependency
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.srt_last_error.restype = c_char_p
    lib.srt_raster_create.argtypes = [c_int, POINTER(c_void_p)]
    lib.srt_raster_destroy.argtypes = [c_void_p]
    lib.srt_raster_set_target.argtypes = [c_void_p, c_uint32, c_uint32, c_uint32]
    lib.srt_raster_clear.argtypes = [c_void_p]
    lib.srt_raster_submit.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.srt_raster_resolve.argtypes = [c_void_p, c_void_p]
    lib.srt_raster_resolve_device.argtypes = [c_void_p, c_void_p, POINTER(c_void_p)]
    lib.srt_raster_read_samples.argtypes = [c_void_p, c_void_p]
    lib.srt_raster_stats.argtypes = [c_void_p, POINTER(RasterStats)]
    lib.srt_raster_sync.argtypes = [c_void_p]
    lib.srt_raster_bind_output.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.srt_raster_invalidate.argtypes = [c_void_p]
    lib.srt_raster_texture_upload_bytes.argtypes = [c_void_p, POINTER(ctypes.c_uint64)]
    _bind_pathtracer(lib)
    _lib = lib
    return lib


def _bind_pathtracer(lib: ctypes.CDLL) -> None:
    from . import _pt_bindings  # noqa: WPS433  (kept separate: the scene structs are long)

    _pt_bindings.bind(lib)


def _check(lib, status: int) -> None:
    if status != 0:
        raise SrtError(status, lib.srt_last_error().decode("utf-8", "replace"))


def make_prims(n: int) -> np.ndarray:
    return np.zeros(n, dtype=PRIM_DTYPE)


def image_prims(rects: np.ndarray, tex_ids) -> np.ndarray:
    """rects: (n,4) float32 x0 y0 x1 y1 (the float parameters of rasterize_image); tex_ids: texture id per image."""
    rects = np.ascontiguousarray(rects, dtype=np.float32).reshape(-1, 4)
    p = make_prims(len(rects))
    p["kind"] = PRIM_IMAGE
    p["reserved"] = np.asarray(tex_ids, dtype=np.uint32)
    v = np.zeros((len(rects), 6), np.float32)
    v[:, :4] = rects
    p["v"] = v.view(np.float64).reshape(-1, 3)
    return p


def triangle_prims(xy: np.ndarray, rgba: np.ndarray) -> np.ndarray:
    """xy: (n,6) float32 x0 y0 x1 y1 x2 y2; rgba: (n,4) float32."""
    xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 6)
    p = make_prims(len(xy))
    p["kind"] = PRIM_TRIANGLE
    p["v"] = xy.view(np.float64).reshape(-1, 3)
    p["rgba"] = np.asarray(rgba, dtype=np.float32).reshape(-1, 4)
    return p


def point_prims(xy: np.ndarray, rgba: np.ndarray) -> np.ndarray:
    """xy: (n,2) float64; rgba: (n,4) float32."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    p = make_prims(len(xy))
    p["kind"] = PRIM_POINT
    p["v"][:, :2] = xy
    p["rgba"] = np.asarray(rgba, dtype=np.float32).reshape(-1, 4)
    return p


class SoftwareRenderer:
    """Mirror of CMU462::SoftwareRenderer for the HIP path.

    Same call sequence as DrawSVG::resize / redraw (drawsvg.cpp:107-124, 435-455):
    ``set_render_target`` -> ``set_sample_rate`` -> ``clear_target`` -> ``draw_stream``.
    ``draw_stream`` takes the ordered primitive stream the C++ drop-in
    (host/software_renderer_hip.cpp) derives from an SVG; the SVG parser itself is
    outside the hot path and is not re-implemented in Python.
    """

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._ctx = c_void_p()
        _check(self._lib, self._lib.srt_raster_create(device, ctypes.byref(self._ctx)))
        self.sample_rate = 1  # SoftwareRenderer() : sample_rate(1)  (software_renderer.h:28)
        self.render_target = None
        self.target_w = self.target_h = 0

    def close(self) -> None:
        if self._ctx:
            self._lib.srt_raster_bind_output(self._ctx, None, 0)   # unpin while the framebuffer is still alive
            self._lib.srt_raster_destroy(self._ctx)
            self._ctx = c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- reference surface -------------------------------------------------------------------
    def set_sample_rate(self, sample_rate: int) -> None:
        if self.sample_rate == sample_rate:
            return
        self.sample_rate = int(sample_rate)
        if self.target_w:
            _check(self._lib, self._lib.srt_raster_set_target(self._ctx, self.target_w, self.target_h, self.sample_rate))

    def set_render_target(self, render_target: np.ndarray | None, width: int, height: int) -> None:
        """render_target: uint8 array of width*height*4 (owned by the caller) or None to let
        draw_stream allocate one."""
        self.render_target = render_target
        self.target_w, self.target_h = int(width), int(height)
        _check(self._lib, self._lib.srt_raster_set_target(self._ctx, self.target_w, self.target_h, self.sample_rate))
        # the lent framebuffer is pinned once (as the C++ drop-in does in set_render_target): read-backs are one DMA transfer
        if render_target is not None and render_target.flags.c_contiguous:
            _check(self._lib, self._lib.srt_raster_bind_output(self._ctx, render_target.ctypes.data_as(c_void_p), render_target.nbytes))
        else:
            _check(self._lib, self._lib.srt_raster_bind_output(self._ctx, None, 0))

    def clear_target(self) -> None:
        if self.render_target is not None:
            self.render_target[...] = 255
        _check(self._lib, self._lib.srt_raster_clear(self._ctx))

    def draw_stream(self, prims: np.ndarray) -> np.ndarray:
        """clear + ordered fill + resolve; returns the (h, w, 4) uint8 render target."""
        self.clear_target()
        self.submit(prims)
        return self.resolve()

    # -- C-ABI steps, exposed for tests and the benchmark ---------------------------------------
    def submit(self, prims: np.ndarray) -> None:
        prims = np.ascontiguousarray(prims, dtype=PRIM_DTYPE)
        _check(self._lib, self._lib.srt_raster_submit(self._ctx, prims.ctypes.data_as(c_void_p), len(prims)))

    def add_texture(self, levels) -> int:
        """levels: the mip chain [(w, h, texels uint8 [h, w, 4]), ...] (level 0 first) as Sampler2D::generate_mips
        left it.  Returns the texture id that goes into an image primitive's `reserved` field."""
        n = len(levels)
        ws = (ctypes.c_uint32 * n)(*[int(l[0]) for l in levels])
        hs = (ctypes.c_uint32 * n)(*[int(l[1]) for l in levels])
        keep = [np.ascontiguousarray(l[2], dtype=np.uint8) for l in levels]
        for (w, h, _), k in zip(levels, keep):
            assert k.size == 4 * int(w) * int(h)
        ptrs = (c_void_p * n)(*[k.ctypes.data for k in keep])
        tid = ctypes.c_uint32()
        _check(self._lib, self._lib.srt_raster_add_texture(self._ctx, n, ws, hs, ptrs, ctypes.byref(tid)))
        return tid.value

    def clear_textures(self) -> None:
        _check(self._lib, self._lib.srt_raster_clear_textures(self._ctx))

    def texture_upload_bytes(self) -> int:
        """Bytes of texels this context has uploaded to the device so far (a redraw of unchanged textures adds none)."""
        n = ctypes.c_uint64()
        _check(self._lib, self._lib.srt_raster_texture_upload_bytes(self._ctx, ctypes.byref(n)))
        return int(n.value)

    def resolve(self) -> np.ndarray:
        out = self.render_target
        if out is None:
            out = np.empty((self.target_h, self.target_w, 4), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.size == self.target_w * self.target_h * 4 and out.flags.c_contiguous
        _check(self._lib, self._lib.srt_raster_resolve(self._ctx, out.ctypes.data_as(c_void_p)))
        return out.reshape(self.target_h, self.target_w, 4)

    def resolve_device(self, stream: int = 0) -> int:
        """Enqueue one full frame (clear + fill + resolve) on `stream`; returns the device pointer of
        the RGBA8 image. Does not synchronize."""
        ptr = c_void_p()
        _check(self._lib, self._lib.srt_raster_resolve_device(self._ctx, c_void_p(stream), ctypes.byref(ptr)))
        return ptr.value

    def invalidate(self) -> None:
        """The next frame derives bounding boxes, line tables and bin lists again (a full frame of a resident stream)."""
        _check(self._lib, self._lib.srt_raster_invalidate(self._ctx))

    def read_samples(self) -> np.ndarray:
        sr = self.sample_rate
        out = np.empty((self.target_h * sr, self.target_w * sr, 4), dtype=np.float32)
        _check(self._lib, self._lib.srt_raster_read_samples(self._ctx, out.ctypes.data_as(c_void_p)))
        return out

    def stats(self) -> RasterStats:
        st = RasterStats()
        _check(self._lib, self._lib.srt_raster_stats(self._ctx, ctypes.byref(st)))
        return st

    def sync(self) -> None:
        _check(self._lib, self._lib.srt_raster_sync(self._ctx))


from ._pt_bindings import LOGGED_RAY_DTYPE, Pathtracer, PathtracerGroup, Scene, SrtCancelled  # noqa: E402,F401

"""Test-side loaders for the CHECKERS (oracle/ and oracle/_ref/) and for golden fixtures.

Nothing in here is imported by the product package."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_ROOT = "/root/reference"

PRIM_DTYPE = np.dtype(
    {
        "names": ["kind", "reserved", "v", "rgba"],
        "formats": ["<u4", "<u4", ("<f8", (3,)), ("<f4", (4,))],
        "offsets": [0, 4, 8, 32],
        "itemsize": 48,
    }
)


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        _oracle = ctypes.CDLL(os.path.join(ORACLE_DIR, "_build", "liboracle.so"))
    return _oracle


def have_reference():
    return os.path.isdir(REF_ROOT)


_ref_raster = None


def ref_raster():
    """The reference's own rasterizer (oracle/_ref/libref_raster.so) or None."""
    global _ref_raster
    path = os.path.join(ORACLE_DIR, "_ref", "libref_raster.so")
    if _ref_raster is None and os.path.exists(path):
        lib = ctypes.CDLL(path)
        lib.ref_raster_svg_stream.restype = ctypes.c_long
        _ref_raster = lib
    return _ref_raster


def oracle_raster_frame(prims, w, h, sr, want_samples=False):
    """Run oracle/raster_oracle.c on an ordered stream. Returns (rgba8, samples|None, counts)."""
    prims = np.ascontiguousarray(prims, dtype=PRIM_DTYPE)
    rgba = np.zeros((h, w, 4), np.uint8)
    ss = np.zeros((h * sr, w * sr, 4), np.float32) if want_samples else None
    counts = np.zeros(4, np.uint64)
    rc = oracle().srt_oracle_raster_frame(
        P(prims), ctypes.c_size_t(len(prims)), w, h, sr, P(rgba), P(ss) if want_samples else None, P(counts)
    )
    assert rc == 0
    return rgba, ss, counts


def ref_raster_prims(prims, w, h, sr, want_samples=False):
    prims = np.ascontiguousarray(prims, dtype=PRIM_DTYPE)
    rgba = np.zeros((h, w, 4), np.uint8)
    ss = np.zeros((h * sr, w * sr, 4), np.float32) if want_samples else None
    rc = ref_raster().ref_raster_prims(
        P(prims), ctypes.c_size_t(len(prims)), w, h, sr, P(rgba), P(ss) if want_samples else None
    )
    assert rc == 0
    return rgba, ss


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

#!/usr/bin/env python3
"""Generate the tone-mapping golden fixtures from the REFERENCE's own code (HDR_Image::tonemap_to compiled into
oracle/_ref/libref_pt.so; std::exp / std::pow resolve to this image's glibc 2.35).

Run in the authoring container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_tonemap_golden.py

Each fixture holds a radiance image (input), an exposure (input) and the reference's RGBA bytes (expected output).
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _harness as H  # noqa: E402
from _cases import tonemap_image  # noqa: E402


def main():
    lib = H.ref_pt_lib()
    assert lib is not None, "build oracle/_ref first: make -C oracle ref"
    for name, w, h, seed, exposure in (("mixed", 64, 48, 5, 1.0), ("mixed", 40, 30, 6, 0.35), ("mixed", 33, 17, 7, 2.5),
                                        ("render", 32, 32, 0, 1.0)):
        rgb = tonemap_image(name, w, h, seed)
        out = np.zeros((h, w, 4), np.uint8)
        lib.ref_pt_tonemap(ctypes.c_uint32(w), ctypes.c_uint32(h), H.P(rgb), ctypes.c_float(exposure), H.P(out))
        path = os.path.join(HERE, f"tonemap_{name}_{w}x{h}_e{exposure:g}.npz")
        np.savez_compressed(path, rgb=rgb, exposure=np.float32(exposure), rgba=out, name=np.array(name), seed=np.int64(seed))
        print("wrote", os.path.basename(path), "mean byte", float(out[..., :3].mean()))


if __name__ == "__main__":
    main()

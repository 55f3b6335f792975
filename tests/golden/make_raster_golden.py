#!/usr/bin/env python3
"""Generate the rasterizer golden fixtures from the REFERENCE's own code.

Run in the authoring container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_raster_golden.py

For every case the reference's SoftwareRendererImp (compiled from its sources by oracle/Makefile into
oracle/_ref/libref_raster.so) renders the image; the fixture stores
  prims      the ordered primitive stream (inputs; from the SVG via our host walk - one record per rasterize_triangle /
             rasterize_line / rasterize_point / rasterize_image call - or synthetic)
  rgba       the reference's RGBA8 render target            (expected output)
  ss_sha256  SHA-256 of the reference's float supersample buffer (expected output, hashed: it is 16 B/sample)
  meta       w, h, sample_rate
SVG cases additionally assert, at generation time, that feeding `prims` to the reference's private
rasterize_* functions reproduces draw_svg's output bit for bit (i.e. the host walk is faithful).
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _harness as H  # noqa: E402
from _cases import adversarial_stream, image_stream, line_stream  # noqa: E402

SVG_DIR = os.path.join(H.REF_ROOT, "Assignments/DrawSVG/svg")

# name, svg, w, h, sample_rate
SVG_CASES = [
    ("cfg1_triangle1_256_ss1", "subdiv/triangle1.svg", 256, 256, 1),   # BASELINE configs[0]
    ("cfg2_test3_1024_ss4", "basic/test3.svg", 1024, 1024, 4),         # BASELINE configs[1]
    ("test4_512_ss3", "basic/test4.svg", 512, 512, 3),
    ("test5_300x200_ss2", "basic/test5.svg", 300, 200, 2),
    ("test6_256_ss4", "basic/test6.svg", 256, 256, 4),
    ("test2_lines_256_ss1", "basic/test2.svg", 256, 256, 1),
    ("prism_alpha_400x300_ss2", "alpha/01_prism.svg", 400, 300, 2),
    ("buckyball_alpha_256_ss4", "alpha/03_buckyball.svg", 256, 256, 4),
    ("degenerate1_256_ss2", "hardcore/01_degenerate_square1.svg", 256, 256, 2),
    ("degenerate2_512_ss2", "hardcore/02_degenerate_square2.svg", 512, 512, 2),   # 1000 frame-sized triangles (the stress SVG, reduced)
    ("lion_384_ss3", "illustration/05_lion.svg", 384, 384, 3),
    ("hexes_320x240_ss1", "illustration/02_hexes.svg", 320, 240, 1),
    # <image> elements: rasterize_image + Sampler2DImp::sample_trilinear over the reference's own mip chains
    ("test7_image_256_ss2", "basic/test7.svg", 256, 256, 2),
    ("test7_image_128x160_ss3", "basic/test7.svg", 128, 160, 3),
    ("scotty_image_300x200_ss1", "alpha/04_scotty.svg", 300, 200, 1),
]


def main():
    ref = H.ref_raster()
    assert ref is not None, "build oracle/_ref first: make -C oracle ref"
    only_images = "--images-only" in sys.argv
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--stress" in sys.argv:
        # SURVEY.md 8(d)'s stress variant at full size: hardcore/02_degenerate_square2.svg, 1024 x 1024, supersample 4 - 1000
        # triangles, 5.86 G sample tests, 22 s in the reference.  Kept out of the raster_*.npz set (the CPU suite would spend a
        # minute in the oracle on it); the GPU test and bench.py's stress line use it.
        svg, w, h, sr = "hardcore/02_degenerate_square2.svg", 1024, 1024, 4
        path = os.path.join(SVG_DIR, svg).encode()
        rgba = np.zeros((h, w, 4), np.uint8)
        ss = np.zeros((h * sr, w * sr, 4), np.float32)
        assert ref.ref_raster_render_svg(path, w, h, sr, H.P(rgba), H.P(ss)) == 0
        prims = np.zeros(100000, H.PRIM_DTYPE)
        n = ref.ref_raster_svg_stream(path, w, h, sr, H.P(prims), ctypes.c_size_t(len(prims)))
        np.savez_compressed(os.path.join(HERE, "stress_degenerate2_1024_ss4.npz"), prims=prims[:n].copy(), rgba=rgba,
                            ss_sha256=np.array(H.sha(ss)), meta=np.array([w, h, sr], np.int64), source=np.array(svg))
        print(f"stress: {n} prims rgba sha {H.sha(rgba)[:12]}")
        return
    for name, svg, w, h, sr in ([] if only_images else SVG_CASES):
        if only and name not in only:
            continue
        path = os.path.join(SVG_DIR, svg).encode()
        rgba = np.zeros((h, w, 4), np.uint8)
        ss = np.zeros((h * sr, w * sr, 4), np.float32)
        assert ref.ref_raster_render_svg(path, w, h, sr, H.P(rgba), H.P(ss)) == 0
        cap = 2_000_000
        prims = np.zeros(cap, H.PRIM_DTYPE)
        n = ref.ref_raster_svg_stream(path, w, h, sr, H.P(prims), ctypes.c_size_t(cap))
        assert 0 <= n <= cap
        prims = prims[:n].copy()
        tex = H.ref_svg_textures(path, w, h, sr)
        rgba2, ss2 = H.ref_raster_prims(prims, w, h, sr, want_samples=True, textures=tex)
        assert np.array_equal(rgba, rgba2) and np.array_equal(ss.view(np.uint32), ss2.view(np.uint32)), name
        np.savez_compressed(
            os.path.join(HERE, f"raster_{name}.npz"),
            prims=prims, rgba=rgba, ss_sha256=np.array(H.sha(ss)), meta=np.array([w, h, sr], np.int64),
            source=np.array(svg), **(tex.to_npz() if len(tex) else {}),
        )
        print(f"{name}: {n} prims ({int((prims['kind'] == 1).sum())} triangles) rgba sha {H.sha(rgba)[:12]}")

    for sr in (() if (only_images or only) else (1, 2, 3, 4, 5)):
        w, h = 97, 61
        prims = adversarial_stream(seed=1234 + sr, w=w, h=h)
        rgba, ss = H.ref_raster_prims(prims, w, h, sr, want_samples=True)
        np.savez_compressed(
            os.path.join(HERE, f"raster_adversarial_ss{sr}.npz"),
            prims=prims, rgba=rgba, ss_sha256=np.array(H.sha(ss)), meta=np.array([w, h, sr], np.int64),
            source=np.array("tests/_cases.py:adversarial_stream"),
        )
        print(f"adversarial ss{sr}: {len(prims)} prims rgba sha {H.sha(rgba)[:12]}")

    # rasterize_line_xiaolinwu on LINE records (the device expands them): every octant, ties, sub-pixel, off-target, non-finite
    for sr in (() if (only_images or only) else (1, 2, 3, 4, 7)):
        w, h = 83, 67
        prims = line_stream(seed=4321 + sr, w=w, h=h)
        rgba, ss = H.ref_raster_prims(prims, w, h, sr, want_samples=True)
        np.savez_compressed(
            os.path.join(HERE, f"raster_lines_ss{sr}.npz"),
            prims=prims, rgba=rgba, ss_sha256=np.array(H.sha(ss)), meta=np.array([w, h, sr], np.int64),
            source=np.array("tests/_cases.py:line_stream"),
        )
        print(f"lines ss{sr}: {len(prims)} prims ({int((prims['kind'] == 4).sum())} lines) rgba sha {H.sha(rgba)[:12]}")

    # synthetic image streams: textures with the reference's mip chains, images folded over column / row 0,
    # minified below the last level, magnified, drawn under and over translucent triangles
    for sr in (() if only else (1, 2, 4)):
        w, h = 90, 70
        prims, level0 = image_stream(seed=77 + sr, w=w, h=h)
        # Mip chains from the oracle's generate_mips, not the reference's: Sampler2DImp::generate_mips (texture.cpp:53-121)
        # reads past the end of a level on non-square / odd-sized textures (two calls in one process already differ), and
        # three of these textures are such.  Mip chains are application data for the product; what is pinned here is the
        # reference's rasterize_image + sample_trilinear on GIVEN mips.  (test_raster_oracle.py pins the oracle's
        # generate_mips against the reference's on the sizes where the latter is defined.)
        tex = H.Textures.from_level0(level0, H.oracle_generate_mips)
        rgba, ss = H.ref_raster_prims(prims, w, h, sr, want_samples=True, textures=tex)
        np.savez_compressed(
            os.path.join(HERE, f"raster_images_ss{sr}.npz"),
            prims=prims, rgba=rgba, ss_sha256=np.array(H.sha(ss)), meta=np.array([w, h, sr], np.int64),
            source=np.array("tests/_cases.py:image_stream"), **tex.to_npz(),
        )
        print(f"images ss{sr}: {len(prims)} prims rgba sha {H.sha(rgba)[:12]}")


if __name__ == "__main__":
    main()

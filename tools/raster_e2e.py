#!/usr/bin/env python3
"""DrawSVG's redraw through the drop-in class (integration/_build/libdropin_raster.so), wall clock: the three figures bench.py reports as
raster.draw_svg, over more frames.  usage: raster_e2e.py [frames] [svg] [w h sr]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _harness as H

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
svg = sys.argv[2] if len(sys.argv) > 2 else os.path.join(H.GOLDEN, "svg", "test3.svg")
w, h, sr = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (1024, 1024, 4)
lib = ctypes.CDLL(os.path.join(H.ROOT, "integration", "_build", "libdropin_raster.so"))
ms = (ctypes.c_double * 5)()
out = np.zeros((h, w, 4), np.uint8)
for rep in range(3):
    rc = lib.dropin_raster_bench(svg.encode(), 0, w, h, sr, frames, ms, H.P(out))
    assert rc == 0, rc
    print(f"redraw wall {ms[0]:.4f} ms (inside draw_svg {ms[3]:.4f})   unchanged view {ms[1]:.4f} ms (inside draw_svg {ms[4]:.4f})   host stream build {ms[2]:.4f} ms", flush=True)
ph = (ctypes.c_double * 5)()
if hasattr(lib, "dropin_raster_phases") and lib.dropin_raster_phases(svg.encode(), 0, w, h, sr, frames, ph) == 0:
    print("phases (ms): application memset %.4f  stream build %.4f  clear + submit %.4f  resolve (upload, kernels, read-back, wait) %.4f  redraw %.4f" % tuple(ph))
g = os.path.join(H.GOLDEN, "raster_cfg2_test3_1024_ss4.npz")
if len(sys.argv) <= 2 and os.path.exists(g):
    print("framebuffer equals the reference golden:", bool(np.array_equal(out, np.load(g)["rgba"])))

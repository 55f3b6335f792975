#!/usr/bin/env python3
"""Diagnostic: frames of a golden raster stream in a loop (for rocprofv3 kernel stats / PMC passes).
usage: raster_bench.py [golden npz] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import srt_amd

name = sys.argv[1] if len(sys.argv) > 1 else "raster_cfg2_test3_1024_ss4.npz"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
g = np.load(os.path.join(ROOT, "tests", "golden", name))
w, h, sr = (int(x) for x in g["meta"])
ren = srt_amd.SoftwareRenderer(0)
ren.set_render_target(None, w, h)
ren.set_sample_rate(sr)
ren.clear_target()
ren.submit(g["prims"])
out = ren.resolve()
print("matches golden:", bool(np.array_equal(out, g["rgba"])))
st = ren.stats()
stream = torch.cuda.current_stream().cuda_stream
redraw = os.environ.get("SRT_BENCH_REDRAW") == "1"      # frames of an unchanged stream: the tile kernel alone
for _ in range(3):
    ren.invalidate(); ren.resolve_device(stream)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(frames):
    if not redraw:
        ren.invalidate()                                  # a FULL frame: setup + binning + tiles
    ren.resolve_device(stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / frames
print(f"{name}{' (redraw)' if redraw else ''}: {dt*1e3:.3f} ms/frame, {st.fragments/dt/1e6:.0f} Mfrags/s, {st.sample_tests/dt/1e9:.1f} G tests/s, bin entries {st.bin_entries}, list bytes {st.list_bytes}")
# end to end through the host-buffer boundary: submit (identical stream: no upload) / new stream (upload) + frame + read-back
fb = np.empty((h, w, 4), np.uint8)
ren.set_render_target(fb, w, h)
ren.set_sample_rate(sr)
prims = g["prims"]
ren.draw_stream(prims)
t = time.perf_counter()
for _ in range(frames):
    ren.draw_stream(prims)
same = (time.perf_counter() - t) / frames
alt = prims.copy(); alt["rgba"][:, 0] *= 0.5
t = time.perf_counter()
for k in range(frames):
    ren.draw_stream(alt if k % 2 else prims)
new = (time.perf_counter() - t) / frames
print(f"  draw_stream wall (clear + submit + resolve into a pinned host buffer): same stream {same*1e3:.3f} ms, alternating streams {new*1e3:.3f} ms")

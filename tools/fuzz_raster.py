#!/usr/bin/env python3
"""Differential fuzzing of the rasterizer: seeded random frames (random size, sample rate 1..6 / 8 / 16, random and
adversarial triangles, points, lines, image records over random rectangles with random textures) through the C ABI against the
CPU oracle: RGBA8 and the float supersample buffer bit for bit, and the work counters.  usage: fuzz_raster.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import srt_amd
import _harness as H
from _cases import PRIM_DTYPE, adversarial_stream, random_triangles

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
H.build_oracle()
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 140)), int(rng.integers(1, 120))
    sr = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 16, 20, 32]))
    parts = [random_triangles(seed, int(rng.integers(0, 120)), w, h, float(rng.choice([2.0, 15.0, 60.0, 400.0])))]
    if rng.random() < 0.5:
        parts.append(adversarial_stream(seed, w, h))
    textures = None
    if rng.random() < 0.6:
        level0 = [rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 40)), 4), dtype=np.uint8) for _ in range(int(rng.integers(1, 4)))]
        textures = H.Textures.from_level0(level0, H.oracle_generate_mips)
        n = int(rng.integers(1, 10))
        img = np.zeros(n, PRIM_DTYPE)
        img["kind"] = 3
        img["reserved"] = rng.integers(0, len(level0), n)
        v = np.zeros((n, 6), np.float32)
        x0 = rng.uniform(-20, w + 10, n); y0 = rng.uniform(-20, h + 10, n)
        v[:, 0] = x0; v[:, 1] = y0
        v[:, 2] = x0 + rng.choice([0.0, 0.5, 3.0, 17.25, 90.0], n) * rng.random(n)
        v[:, 3] = y0 + rng.choice([0.0, 0.5, 3.0, 17.25, 90.0], n) * rng.random(n)
        img["v"] = v.view(np.float64).reshape(-1, 3)
        parts.append(img)
    if rng.random() < 0.7:     # SRT_PRIM_LINE records: rasterize_line_xiaolinwu expanded on the device
        n = int(rng.integers(1, 60))
        ln = np.zeros(n, PRIM_DTYPE)
        ln["kind"] = 4
        v = np.zeros((n, 6), np.float32)
        a = rng.uniform(-12, [w + 12, h + 12], (n, 2))
        length = rng.choice([0.0, 0.4, 1.0, 3.0, 9.0, 40.0, 300.0], n) * rng.random(n)
        ang = rng.random(n) * 2 * np.pi
        b = a + length[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1)
        snap = rng.random(n) < 0.4                      # end points on quarter pixels: round() ties, exact slopes
        a[snap] = np.round(a[snap] * 4) / 4; b[snap] = np.round(b[snap] * 4) / 4
        v[:, 0:2] = a; v[:, 2:4] = b
        ln["v"] = v.view(np.float64).reshape(-1, 3)
        ln["rgba"] = rng.random((n, 4)).astype(np.float32)
        parts.append(ln)
    prims = np.concatenate(parts)
    prims = prims[rng.permutation(len(prims))] if len(prims) else prims
    ren = srt_amd.SoftwareRenderer(0)
    # odd seeds lend the renderer a framebuffer, as DrawSVG does (pinned: srt_raster_bind_output)
    lent = np.full((h, w, 4), 9, np.uint8) if seed & 1 else None
    ren.set_render_target(lent, w, h)
    ren.set_sample_rate(sr)
    for t in range(len(textures) if textures is not None else 0):
        ren.add_texture(textures.texture(t))
    rgba = ren.draw_stream(prims).copy()
    ss = ren.read_samples()
    st = ren.stats()
    ren.close()
    o_rgba, o_ss, c = H.oracle_raster_frame(prims, w, h, sr, want_samples=True, textures=textures)
    ok = np.array_equal(rgba, o_rgba) and np.array_equal(ss.view(np.uint32), o_ss.view(np.uint32))
    ok_c = (st.sample_tests, st.sample_tests_in_target, st.fragments, st.point_samples) == tuple(int(x) for x in c)
    if not (ok and ok_c):
        bad += 1
        print(f"MISMATCH seed {seed}: {w}x{h} ss{sr}, {len(prims)} prims, image equal {ok}, counters equal {ok_c}", flush=True)
    if (seed - first) % 50 == 49:
        print(f"  {seed - first + 1} frames, {bad} problems, {time.time() - t0:.0f} s", flush=True)
print("fuzz done:", count, "frames,", bad, "problems")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Differential fuzzing of the path-tracer kernels: seeded random scenes (tests/_cases.py:random_pt_scene), epoch image of
every kernel (auto, wave kernel, wave kernel with dead-ray elision, lane per sample, lane per pixel, flattened walk, the
streamed forms with a small slot population on every third seed) against the CPU oracle, bit for bit, plus equal ray counts;
odd seeds build every BVH on the device.  usage: fuzz_pt.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import srt_amd
import _harness as H
from _cases import random_pt_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
H.build_oracle()
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    scene, w, h, depth, use_bvh, spp = random_pt_scene(seed)
    pt = srt_amd.Pathtracer(0)
    pt.set_params(w, h, 1, depth, use_bvh)
    try:
        want = H.OraclePT(scene, w, h, depth, use_bvh).epoch(seed, 3, spp)
    except AssertionError:
        # the reference's BVH build does not terminate on this input (no candidate plane separates the primitives of a
        # node: bvh.inl:100-160 keeps producing a full and an empty child); the product must refuse it too
        try:
            pt.build_scene(scene)
            bad += 1
            print(f"seed {seed}: the oracle refuses the scene, the product builds it", flush=True)
        except srt_amd.SrtError:
            print(f"seed {seed}: non-terminating reference BVH build, refused by oracle and product", flush=True)
        pt.close()
        continue
    if seed & 1:
        pt.set_bvh_builder(True, 1)                            # every tree (the BVH<Object> too) from the device build
    if seed % 3 == 0:
        pt.set_stream_slots(256 * (1 + seed % 5))              # a small population: refills, many generations
    pt.build_scene(scene); pt.set_camera(scene["camera"])
    rays = {}
    for label, mode, elide in (("auto", 0, False), ("wave", 2, False), ("wave+elide", 2, True), ("unit", 4, False), ("unit+elide", 4, True), ("pixel", 1, False),
                               ("flat", 5, False), ("stream", 6, False), ("stream+elide", 6, True), ("stream-sweeps", 7, False), ("stream-sweeps+elide", 7, True)):
        try:
            pt.set_kernel(mode)
        except srt_amd.SrtError:
            continue
        pt.set_elision(elide)
        pt.ray_count(reset=True)
        try:
            img = pt.render_epoch(seed, 3, spp)
        except srt_amd.SrtError as e:
            if "objects" in str(e) or "needs" in str(e) or "does not take" in str(e) or "not apply" in str(e):   # kernel does not take this scene
                continue
            raise
        rays[label] = pt.ray_count()[0]
        ok = np.array_equal(img.view(np.uint32), want.view(np.uint32))
        if not ok:
            bad += 1
            d = np.flatnonzero((img.view(np.uint32) != want.view(np.uint32)).any(axis=2).reshape(-1))
            print(f"MISMATCH seed {seed} kernel {label}: {len(d)} pixels differ, first {d[:4]} ({len(scene['objects'])} objects, {w}x{h}, depth {depth}, bvh {use_bvh}, spp {spp})", flush=True)
    if len(set(rays.values())) > 1:
        bad += 1
        print(f"RAY COUNT seed {seed}: {rays}", flush=True)
    pt.close()
    if (seed - first) % 10 == 9:
        print(f"  {seed - first + 1} scenes, {bad} problems, {time.time() - t0:.0f} s", flush=True)
print("fuzz done:", count, "scenes,", bad, "problems")
sys.exit(1 if bad else 0)

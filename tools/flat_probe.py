#!/usr/bin/env python3
"""Diagnostic: scene.hit through the nested form and through the flattened walk (kernel mode 5) on random rays,
then one small epoch per kernel mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import srt_amd
from _cases import pt_scene, random_rays

name = sys.argv[1]
use_bvh = sys.argv[2] == "1"
modes = [int(m) for m in sys.argv[3].split(",")]
scene = pt_scene(name)
pt = srt_amd.Pathtracer(0)
pt.set_params(32, 32, 4, 8, use_bvh)
pt.build_scene(scene); pt.set_camera(scene["camera"])
org, dirs, bounds = random_rays(3, 4000)
pt.set_kernel(0); a = pt.hit(org, dirs, bounds)
pt.set_kernel(5); b = pt.hit(org, dirs, bounds)
bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(1))[0]
print("hit(): nested vs flat mismatches", len(bad), "of", len(a), "hits", int(a[:, 0].sum()), flush=True)
for i in bad[:5]:
    print(" ", i, a[i], b[i], flush=True)
ref = None
for m in modes:
    print("mode", m, "...", flush=True)
    pt.set_kernel(m)
    img = pt.render_epoch(5, 9, 3)
    if ref is None: ref = img
    print("mode", m, "ok, equal to first:", bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32))), float(np.nanmean(img)), flush=True)
print("--- one ray per launch (slot 0), failing rays first")
pt.set_kernel(5)
for i in list(bad[:6]) + [0, 1, 2]:
    one = pt.hit(org[i:i + 1], dirs[i:i + 1], bounds[i:i + 1])
    print(i, "alone:", "same as nested" if np.array_equal(one.view(np.uint32), a[i:i + 1].view(np.uint32)) else "DIFFERENT", one[0][:2], a[i][:2], flush=True)
print("--- 64 copies of one ray")
i = bad[0] if len(bad) else int(np.nonzero(a[:, 0])[0][0])
rep = pt.hit(np.repeat(org[i:i + 1], 64, 0), np.repeat(dirs[i:i + 1], 64, 0), np.repeat(bounds[i:i + 1], 64, 0))
print("hits per slot-lane:", rep[:, 0].astype(int).tolist(), flush=True)

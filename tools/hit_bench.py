#!/usr/bin/env python3
"""Diagnostic: closest-hit throughput (srt_pt_hit) for incoherent rays in a named scene; run under
rocprofv3 --kernel-trace --stats to read pt_hit_kernel's duration (the wall time here includes the copies)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srt_amd
from soft_rendering_toolsets_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "blob7"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
modes = tuple(int(m) for m in sys.argv[3].split(",")) if len(sys.argv) > 3 else (0, 5)
scene = scenes.cornell_with_mesh(int(name[4:]), "glass") if name.startswith("blob") else scenes.cornell_box(name)
pt = srt_amd.Pathtracer(0)
pt.set_params(64, 64, 1, 8, True)
pt.build_scene(scene); pt.set_camera(scene["camera"])
boxes, links, order = pt.dump_bvh(-1)
lo, hi = boxes[0][:3], boxes[0][3:]
rng = np.random.default_rng(1)
org = (lo + (hi - lo) * (0.05 + 0.9 * rng.random((n, 3)))).astype(np.float32)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
d = d.astype(np.float32)
bounds = np.tile(np.array([1e-5, 3.4e38], np.float32), (n, 1))
outs = []
for mode in modes:
    pt.set_kernel(mode)
    pt.hit(org[:1024], d[:1024], bounds[:1024])
    t = time.perf_counter(); out = pt.hit(org, d, bounds); dt = time.perf_counter() - t
    outs.append(out)
    print(f"mode {mode}: {n} rays, wall {dt*1e3:.1f} ms (with copies), hit fraction {np.mean(out[:, 0] != 0):.3f}")
print("modes agree:", all(np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)) for o in outs[1:]))

#!/usr/bin/env python3
"""Diagnostic: epoch time of a named scene with each kernel (1 = general per-lane, 2 = wave-uniform)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srt_amd
from soft_rendering_toolsets_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "blob7"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
if name.startswith("blob"):
    scene = scenes.cornell_with_mesh(int(name[4:]), "glass")
elif name in ("cbox", "cbox_lambertian"):
    scene = scenes.cornell_box(name)
else:   # any named test scene (tests/_cases.py)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from _cases import pt_scene
    scene = pt_scene(name)
    scene.setdefault("name", name)
pt = srt_amd.Pathtracer(0)
pt.set_params(size, size, spp, 8, True)
t = time.perf_counter(); pt.build_scene(scene); print(f"build_scene {time.perf_counter()-t:.2f} s ({scene['name']})")
pt.set_camera(scene["camera"])
imgs = []
modes = tuple(int(m) for m in sys.argv[4].split(",")) if len(sys.argv) > 4 else (1, 2, 4, 5, 6)
for mode in modes:
    pt.set_kernel(mode)
    pt.render_epoch(0, 0, spp if os.environ.get('SRT_WARM_FULL') else 1)
    pt.ray_count(reset=True)
    t = time.perf_counter(); img = pt.render_epoch(0, 0, spp); dt = time.perf_counter() - t
    rays, cams = pt.ray_count()
    imgs.append(img)
    print(f"mode {mode}: {dt*1e3:.1f} ms, {rays/dt/1e6:.0f} Mrays/s, {rays/cams:.2f} rays/sample")
    if os.environ.get("SRT_STREAM_TIMES") and pt.kernel_form() >= 3:
        pt.stream_times(True); pt.render_epoch(0, 0, spp); ms, g = pt.stream_times(False)
        print(f"   per-kernel (serialised by the timing events): {ms}, {g} generations")
if os.environ.get("SRT_ELIDE"):
    pt.set_elision(True)
    for mode in modes:
        pt.set_kernel(mode)
        pt.render_epoch(0, 0, 1)
        pt.ray_count(reset=True); pt.rays_elided(reset=True)
        t = time.perf_counter(); img = pt.render_epoch(0, 0, spp); dt = time.perf_counter() - t
        rays, cams = pt.ray_count(); el = pt.rays_elided()
        imgs.append(img)
        print(f"mode {mode} + elision: {dt*1e3:.1f} ms, {rays/dt/1e6:.0f} reference-equivalent Mrays/s, {(rays-el)/dt/1e6:.0f} traced, {el/rays:.3f} elided")
    pt.set_elision(False)
print("modes bit-identical:", all(np.array_equal(imgs[0].view(np.uint32), i.view(np.uint32)) for i in imgs[1:]))
rng = np.random.default_rng(0); n = 20000
xs, ys, ss = rng.integers(0, size, n), rng.integers(0, size, n), rng.integers(0, spp, n)
pt.trace_samples(0, xs, ys, ss); c = pt.counters()
print({k: round(v / c["rays"], 2) for k, v in c.items()})

#!/usr/bin/env python3
"""Diagnostic: where does a bounce cycle of the wave kernel spend its shader cycles? (kernel mode 3, stamped build)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srt_amd
from soft_rendering_toolsets_amd import scenes

scene = scenes.cornell_box(sys.argv[1] if len(sys.argv) > 1 else "cbox")
pt = srt_amd.Pathtracer(0)
pt.set_params(1024, 1024, 64, 8, True)
pt.build_scene(scene); pt.set_camera(scene["camera"])
for mode in (2, 3):
    pt.set_kernel(mode)
    pt.render_epoch(0, 0, 16)
    pt.ray_count(reset=True); pt.section_cycles(reset=True)
    t = time.perf_counter(); pt.render_epoch(0, 0, 64); dt = time.perf_counter() - t
    rays, cams = pt.ray_count()
    print(f"mode {mode}: {dt*1e3:.1f} ms (incl. D2H), {rays/dt/1e6:.0f} Mrays/s")
sec = pt.section_cycles()
tot = sum(sec.values())
for k, v in sec.items():
    print(f"  {k:14s} {100.0*v/tot:5.1f} %")

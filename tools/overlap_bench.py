#!/usr/bin/env python3
"""Diagnostic: wall time per 64-spp step of rank 0's share (world 1 and 8) with every step on one stream vs steps
alternating between two streams (the next launch's blocks can fill the CUs the previous launch's tail leaves idle)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import srt_amd
from soft_rendering_toolsets_amd import scenes

scene = scenes.cornell_box("cbox")
pt = srt_amd.Pathtracer(0)
pt.set_params(1024, 1024, 64, 8, True)
pt.build_scene(scene); pt.set_camera(scene["camera"])
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for world in (1, 8):
    pt.set_tiling(32, 32, 0, world)
    local, per_rank, fpt = pt.tile_info()
    tiles = [torch.zeros(per_rank * fpt, dtype=torch.float32, device="cuda") for _ in range(2)]
    for nstreams in (1, 2):
        for warm in (True, False):
            steps = 4 if warm else 16
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                k = i % nstreams
                pt.render_epoch_device(streams[k].cuda_stream, 0, 64 * i, 64, tiles[k].data_ptr())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        print(f"world {world}, {nstreams} stream(s): {dt * 1e3:.3f} ms per step", flush=True)
    a = tiles[0].clone()
    pt.render_epoch_device(streams[0].cuda_stream, 0, 64 * 14, 64, tiles[0].data_ptr()); torch.cuda.synchronize()
    print("  deterministic:", bool(torch.equal(a, tiles[0])))

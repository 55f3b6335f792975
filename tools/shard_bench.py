#!/usr/bin/env python3
"""Diagnostic: device time of rank 0's share of one 64-spp epoch of the bench workload for world sizes 1, 2, 4, 8
(one GPU; what each rank of an N-GPU run executes per step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import srt_amd
from soft_rendering_toolsets_amd import scenes

scene = scenes.cornell_box("cbox")
pt = srt_amd.Pathtracer(0)
pt.set_params(1024, 1024, 64, 8, True)
pt.build_scene(scene); pt.set_camera(scene["camera"])
stream = torch.cuda.current_stream().cuda_stream
base = None
for world in (1, 2, 4, 8):
    pt.set_tiling(32, 32, 0, world)
    local, per_rank, fpt = pt.tile_info()
    tiles = torch.zeros(per_rank * fpt, dtype=torch.float32, device="cuda")
    pt.render_epoch_device(stream, 0, 0, 64, tiles.data_ptr()); torch.cuda.synchronize()
    pt.kernel_time(enable=True)
    for i in range(8):
        pt.render_epoch_device(stream, 0, 64 * (i + 1), 64, tiles.data_ptr())
    torch.cuda.synchronize()
    ms, n = pt.kernel_time(enable=False)
    ms /= n
    base = base or ms
    print(f"world {world}: {ms:.3f} ms per step on rank 0, ideal {base / world:.3f}, efficiency {base / world / ms:.3f}", flush=True)

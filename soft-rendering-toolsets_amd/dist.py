"""Image-tile sharding of the path tracer over the GPUs of one node (one process per GPU).

The arithmetic here mirrors the C ABI (srt_pt_set_tiling / srt_pt_tile_info / pt_untile_kernel in
csrc/pt.hip): tiles of tile_w x tile_h pixels numbered row-major; rank r renders the tiles t with
t % world == r; every rank's tile buffer is padded to tiles_per_rank tiles so that ONE gather per epoch
(RCCL over xGMI on GPUs, gloo in the CPU tests) moves all tile radiance to rank 0.  No reduction collective
is involved: tiles are disjoint.
"""
from __future__ import annotations

import numpy as np


class TileShard:
    def __init__(self, w: int, h: int, tile_w: int = 32, tile_h: int = 32, rank: int = 0, world: int = 1):
        assert tile_w % 8 == 0 and tile_h % 8 == 0 and 0 <= rank < world
        self.w, self.h, self.tile_w, self.tile_h, self.rank, self.world = w, h, tile_w, tile_h, rank, world
        self.tiles_x = (w + tile_w - 1) // tile_w
        self.tiles_y = (h + tile_h - 1) // tile_h
        self.ntiles = self.tiles_x * self.tiles_y
        self.tiles_per_rank = (self.ntiles + world - 1) // world
        self.local = list(range(rank, self.ntiles, world))
        self.floats_per_tile = tile_w * tile_h * 3

    def tile_origin(self, tile: int):
        return (tile % self.tiles_x) * self.tile_w, (tile // self.tiles_x) * self.tile_h

    def pack(self, image: np.ndarray) -> np.ndarray:
        """Full (h, w, 3) image -> this rank's tile-major buffer (tiles_per_rank, tile_h, tile_w, 3), zero padded."""
        out = np.zeros((self.tiles_per_rank, self.tile_h, self.tile_w, 3), np.float32)
        for k, t in enumerate(self.local):
            x0, y0 = self.tile_origin(t)
            blk = image[y0:y0 + self.tile_h, x0:x0 + self.tile_w]
            out[k, : blk.shape[0], : blk.shape[1]] = blk
        return out

    def untile(self, gathered: np.ndarray) -> np.ndarray:
        """(world, tiles_per_rank, tile_h, tile_w, 3) as gathered on the root -> (h, w, 3)."""
        g = np.asarray(gathered, np.float32).reshape(self.world, self.tiles_per_rank, self.tile_h, self.tile_w, 3)
        img = np.zeros((self.h, self.w, 3), np.float32)
        for t in range(self.ntiles):
            x0, y0 = self.tile_origin(t)
            blk = g[t % self.world, t // self.world]
            img[y0:y0 + self.tile_h, x0:x0 + self.tile_w] = blk[: self.h - y0, : self.w - x0]
        return img


def gather_tiles(local, gathered, world: int, rank: int, dst: int = 0) -> None:
    """One collective per epoch: every rank's tile buffer -> `gathered` (world * len(local) elements) on dst.
    `local` / `gathered` are torch tensors on the rank's device (CUDA for RCCL, CPU for gloo)."""
    if world == 1:
        if gathered is not local:
            gathered.copy_(local)
        return
    import torch.distributed as dist

    gather_list = list(gathered.view(world, -1).unbind(0)) if rank == dst else None
    dist.gather(local, gather_list, dst=dst)

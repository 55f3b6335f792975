"""CPU: the N > 1 plumbing (tile sharding + one gather per epoch + untile) with 2 gloo ranks.  The oracle
stands in for the kernel; the tile arithmetic is checked against the C ABI's (host-only context)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _harness as H
from _cases import pt_scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, tw, th, out_path):
    import sys

    sys.path.insert(0, H.ROOT)
    sys.path.insert(0, os.path.join(H.ROOT, "tests"))
    import srt_amd  # noqa: F401
    from soft_rendering_toolsets_amd.dist import TileShard, gather_tiles

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pt_scene("cbox")
    shard = TileShard(w, h, tw, th, rank, world)
    # this rank "renders" only its own tiles: the oracle's epoch image, cropped to the rank's tiles
    full = H.OraclePT(scene, w, h, 4, True).epoch(3, 0, 2)
    local = torch.from_numpy(shard.pack(full).reshape(-1).copy())
    gathered = torch.zeros(world * local.numel()) if rank == 0 else None
    acc = np.zeros((h, w, 3), np.float32)
    for k in (1, 2):                       # two epochs: gather -> untile -> running mean
        gather_tiles(local, gathered, world, rank)
        if rank == 0:
            img = shard.untile(gathered.numpy())
            H.oracle_accumulate(acc, img, k)
    dist.barrier()
    if rank == 0:
        np.save(out_path, acc)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_gather_untile(tmp_path, world):
    w, h, tw, th = 72, 40, 16, 8
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, tw, th, out), nprocs=world, join=True)
    got = np.load(out)
    full = H.OraclePT(pt_scene("cbox"), w, h, 4, True).epoch(3, 0, 2)
    want = np.zeros_like(full)
    for k in (1, 2):
        H.oracle_accumulate(want, full, k)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("w,h,tw,th,world", [(1024, 1024, 32, 32, 8), (72, 40, 16, 8, 3), (33, 17, 32, 32, 2), (64, 64, 8, 8, 5)])
def test_tile_arithmetic_matches_c_abi(w, h, tw, th, world):
    import srt_amd
    from soft_rendering_toolsets_amd.dist import TileShard

    pt = srt_amd.Pathtracer(device=-1)
    pt.set_params(w, h, 1, 8, True)
    total = 0
    for rank in range(world):
        pt.set_tiling(tw, th, rank, world)
        local, per_rank, fpt = pt.tile_info()
        s = TileShard(w, h, tw, th, rank, world)
        assert (local, per_rank, fpt) == (len(s.local), s.tiles_per_rank, s.floats_per_tile)
        total += local
    assert total == ((w + tw - 1) // tw) * ((h + th - 1) // th)
    pt.close()

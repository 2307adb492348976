"""Multi-process path on CPU: world_size 2 and 3 with gloo.

Each rank owns chunks c % world == rank (raytracing_c_amd/multi_gpu.py), fills its compact tile
buffer, ONE all-gather moves the tiles, untile() rebuilds the frame.  The GPU path runs the same
FramePartition / gather_tiles code with RCCL; rt_resolve / rt_untile (HIP) implement the two
layouts that extract_tiles() / untile() state in numpy (checked on the GPU in test_gpu_parity.py).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracing_c_amd.multi_gpu import FramePartition, extract_tiles, gather_tiles, untile
        part = FramePartition(width, height, world)
        full = np.random.default_rng(seed).integers(0, 256, (height, width, 3), dtype=np.uint8)
        # this rank "renders" only its own chunks: everything else stays zero
        mine = np.zeros_like(full)
        for c in part.chunk_ids(rank):
            x0, y0 = part.chunk_origin(c)
            mine[y0:y0 + 32, x0:x0 + 32] = full[y0:y0 + 32, x0:x0 + 32]
        tiles = torch.from_numpy(extract_tiles(mine, rank, world))
        all_tiles = torch.zeros((world, part.max_local, 32 * 32 * 3), dtype=torch.uint8)
        gather_tiles(tiles, all_tiles)
        image = untile(all_tiles.numpy(), width, height, world)
        ok = np.array_equal(image, full)
        # the timing rule of bench.py: MAX over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height", [(2, 100, 70), (3, 1920 // 8, 1080 // 8), (2, 33, 31)])
def test_tile_gather_roundtrip(tmp_path, world, width, height):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, width, height, 1234, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_partition_is_balanced_for_headline_frame():
    from raytracing_c_amd.multi_gpu import FramePartition
    part = FramePartition(1920, 1080, 8)
    counts = [part.n_local(r) for r in range(8)]
    assert sum(counts) == 2040 and max(counts) - min(counts) <= 1
    owners = np.zeros(part.n_chunks, int)
    for r in range(8):
        owners[part.chunk_ids(r)] += 1
    assert (owners == 1).all()

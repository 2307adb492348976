"""Multi-process path on CPU: world_size 2 and 3 with gloo.

Each rank owns the chunks rt_chunk_owner() gives it (lattice partition, include/rt_hip.h), fills its compact tile
buffer, ONE gather moves the tiles to rank 0, untile() rebuilds the frame there.  The GPU path runs the same
FramePartition / gather_tiles code with RCCL; rt_resolve / rt_untile (HIP) implement the two
layouts that extract_tiles() / untile() state in numpy (checked on the GPU in test_gpu_parity.py).
"""
import ctypes as C
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
        all_tiles = gather_tiles(tiles, world, rank)
        ok = (all_tiles is None) == (rank != 0)
        if rank == 0:
            image = untile(all_tiles.numpy(), width, height, world)
            ok = ok and np.array_equal(image, full)
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
    assert sum(counts) == 2040 and max(counts) - min(counts) <= 2 and max(counts) == part.max_local
    owners = np.zeros(part.n_chunks, int)
    for r in range(8):
        owners[part.chunk_ids(r)] += 1
    assert (owners == 1).all()


@pytest.mark.parametrize("world", [2, 3, 4, 5, 6, 7, 8])
def test_lattice_spreads_every_chunk_row_and_column_over_all_ranks(world):
    """rank = (cx + B cy) mod world with B coprime to world: `world` consecutive chunks of any row or column have
    `world` different owners, so a narrow expensive structure (the helmet's centre columns) cannot land on few ranks."""
    import raytracing_c_amd as rt
    w, h = 1920, 1080
    cx_n, cy_n = 60, 34
    own = np.array([rt.lib.rt_chunk_owner(w, h, world, c) for c in range(cx_n * cy_n)]).reshape(cy_n, cx_n)
    assert own.min() == 0 and own.max() == world - 1
    for y in range(cy_n):
        for x0 in range(0, cx_n - world + 1, 7):
            assert len(set(own[y, x0:x0 + world])) == world
    for x in range(cx_n):
        for y0 in range(0, cy_n - world + 1, 5):
            assert len(set(own[y0:y0 + world, x])) == world
    assert rt.lib.rt_chunk_owner(w, h, world, -1) == -1 and rt.lib.rt_chunk_owner(w, h, world, cx_n * cy_n) == -1
    # chunk lists are ascending and consistent with the owner function
    part_lists = []
    for r in range(world):
        n = rt.lib.rt_local_chunk_count(w, h, r, world)
        buf = (C.c_int32 * n)()
        assert rt.lib.rt_local_chunk_list(w, h, r, world, buf, n) == n
        ids = list(buf)
        assert ids == sorted(ids) and all(own.reshape(-1)[c] == r for c in ids)
        part_lists.append(n)
    assert max(part_lists) == rt.lib.rt_max_local_chunk_count(w, h, world)

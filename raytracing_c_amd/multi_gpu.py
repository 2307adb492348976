"""Image partition across the GPUs of one node and the framebuffer tile gather.

The reference work-steals 32x32 chunks between threads (raytracer.c:601-627); here chunk (cx, cy) is
owned by rank (cx + B * cy) % world (a lattice that spreads every chunk column and row over all ranks:
include/rt_hip.h, rt_chunk_owner), every
rank renders its chunks with rt_render_accumulate / rt_resolve into a compact
[max_local][32*32*3] u8 tile buffer, ONE RCCL gather moves the tiles over xGMI to rank 0 (6.2 MB for
1080p in total) and rt_untile scatters them into the row-major image.  Per-path seeds depend
only on (pixel, sample), so the image does not depend on `world`.

extract_tiles()/untile() are numpy statements of the two layouts (what rt_resolve writes and
what rt_untile reads); the CPU tests use them with gloo.
"""
import ctypes as C

import numpy as np

CHUNK = 32


class FramePartition:
    """The library's partition tables (pure host arithmetic in librt_hip.so, no GPU needed)."""

    def __init__(self, width, height, world):
        from .native import lib
        self.width, self.height, self.world = int(width), int(height), int(world)
        self.chunks_x = (self.width + CHUNK - 1) // CHUNK
        self.chunks_y = (self.height + CHUNK - 1) // CHUNK
        self.n_chunks = int(lib.rt_chunk_count(self.width, self.height))
        self.max_local = int(lib.rt_max_local_chunk_count(self.width, self.height, self.world))
        self._lists = []
        for rank in range(self.world):
            n = int(lib.rt_local_chunk_count(self.width, self.height, rank, self.world))
            buf = (C.c_int32 * max(n, 1))()
            assert lib.rt_local_chunk_list(self.width, self.height, rank, self.world, buf, n) == n
            self._lists.append([int(buf[i]) for i in range(n)])

    def chunk_ids(self, rank):
        return self._lists[rank]

    def n_local(self, rank):
        return len(self._lists[rank])

    def owner_slot(self, chunk):
        """(rank, slot) of a global chunk index."""
        from .native import lib
        rank = int(lib.rt_chunk_owner(self.width, self.height, self.world, chunk))
        return rank, self._lists[rank].index(chunk)

    def chunk_origin(self, chunk):
        return (chunk % self.chunks_x) * CHUNK, (chunk // self.chunks_x) * CHUNK


def extract_tiles(image, rank, world):
    """Row-major (H, W, 3) u8 -> this rank's compact tiles (max_local, 32*32*3); pixels of a chunk
    that fall outside the image are 0 (what rt_resolve writes)."""
    h, w, _ = image.shape
    part = FramePartition(w, h, world)
    tiles = np.zeros((part.max_local, CHUNK * CHUNK * 3), np.uint8)
    for l, c in enumerate(part.chunk_ids(rank)):
        x0, y0 = part.chunk_origin(c)
        t = np.zeros((CHUNK, CHUNK, 3), np.uint8)
        blk = image[y0:y0 + CHUNK, x0:x0 + CHUNK]
        t[:blk.shape[0], :blk.shape[1]] = blk
        tiles[l] = t.reshape(-1)
    return tiles


def untile(all_tiles, width, height, world):
    """(world, max_local, 32*32*3) rank-major gathered tiles -> (H, W, 3) image (what rt_untile does)."""
    part = FramePartition(width, height, world)
    image = np.zeros((height, width, 3), np.uint8)
    for rank in range(world):
        for slot, c in enumerate(part.chunk_ids(rank)):
            x0, y0 = part.chunk_origin(c)
            t = np.asarray(all_tiles[rank][slot]).reshape(CHUNK, CHUNK, 3)
            hh, ww = min(CHUNK, height - y0), min(CHUNK, width - x0)
            image[y0:y0 + hh, x0:x0 + ww] = t[:hh, :ww]
    return image


_gather_buffers = {}


def gather_tiles(tiles, world, rank, dst=0):
    """tiles: this rank's (max_local, 3072) u8 tensor.  ONE gather to rank `dst`: returns the
    (world, max_local, 3072) rank-major tensor there (on the device of `tiles`), None on the other ranks.

    With the nccl backend torch issues this as one group of ncclSend / ncclRecv: on the fully connected xGMI
    topology every peer has its own link into rank 0, so the exchange costs one tile buffer per link (0.78 MB at
    1080p u8) instead of the world-1 ring steps of an all-gather that would also deliver the frame to ranks
    that never use it.  gloo in the CPU tests."""
    import torch
    import torch.distributed as dist
    if rank != dst:
        dist.gather(tiles, None, dst=dst)
        return None
    key = (tiles.device, tuple(tiles.shape), world)
    buf = _gather_buffers.get(key)
    if buf is None:
        buf = torch.zeros((world,) + tuple(tiles.shape), dtype=tiles.dtype, device=tiles.device)
        _gather_buffers.clear()
        _gather_buffers[key] = buf
    dist.gather(tiles, list(buf.unbind(0)), dst=dst)
    return buf

"""Image partition across the GPUs of one node and the framebuffer tile gather.

The reference work-steals 32x32 chunks between threads (raytracer.c:601-627); here chunk c is
owned by rank c % world (interleaving balances sky-only against helmet-covered chunks), every
rank renders its chunks with rt_render_accumulate / rt_resolve into a compact
[max_local][32*32*3] u8 tile buffer, ONE RCCL all-gather moves the tiles over xGMI (6.2 MB for
1080p in total) and rt_untile scatters them into the row-major image.  Per-path seeds depend
only on (pixel, sample), so the image does not depend on `world`.

extract_tiles()/untile() are numpy statements of the two layouts (what rt_resolve writes and
what rt_untile reads); the CPU tests use them with gloo.
"""
import numpy as np

CHUNK = 32


class FramePartition:
    def __init__(self, width, height, world):
        self.width, self.height, self.world = int(width), int(height), int(world)
        self.chunks_x = (self.width + CHUNK - 1) // CHUNK
        self.chunks_y = (self.height + CHUNK - 1) // CHUNK
        self.n_chunks = self.chunks_x * self.chunks_y
        self.max_local = (self.n_chunks + self.world - 1) // self.world

    def chunk_ids(self, rank):
        return list(range(rank, self.n_chunks, self.world))

    def n_local(self, rank):
        return len(self.chunk_ids(rank))

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
    for c in range(part.n_chunks):
        x0, y0 = part.chunk_origin(c)
        t = np.asarray(all_tiles[c % world][c // world]).reshape(CHUNK, CHUNK, 3)
        hh, ww = min(CHUNK, height - y0), min(CHUNK, width - x0)
        image[y0:y0 + hh, x0:x0 + ww] = t[:hh, :ww]
    return image


def gather_tiles(tiles, all_tiles):
    """tiles: this rank's (max_local, 3072) u8 tensor; all_tiles: (world, max_local, 3072) tensor on the
    same device.  One all-gather (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(all_tiles.view(-1), tiles.view(-1))
    return all_tiles

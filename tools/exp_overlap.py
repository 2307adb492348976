#!/usr/bin/env python3
"""GPU experiment: frames in flight.  Renders K frames of one rank's share (world = 1, 2, 4, 8) back to back on ONE stream,
then alternating between TWO streams with two device scenes (own work queue / counters each) and two accumulation buffers, so
that the tail of frame k overlaps the head of frame k+1.  Reports ms per frame for both."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                           # noqa: E402
import raytracing_c_amd as rt                          # noqa: E402
from raytracing_c_amd import ctypes_abi as abi         # noqa: E402
from raytracing_c_amd.configs import load_config       # noqa: E402

assert rt.lib.rt_init(0) == 0
hs, cfg = load_config("helmet")
w, h, b = 1920, 1080, 8
ds = [rt.lib.rt_scene_upload(C.byref(hs.scene)) for _ in range(2)]
accum = [torch.zeros((h, w, 3), dtype=torch.int64, device="cuda") for _ in range(2)]
image = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
K = 12


def frame(i, p, two):
    j = i & 1 if two else 0
    with torch.cuda.stream(streams[j]):
        accum[j].zero_()
        assert rt.lib.rt_render_accumulate(ds[j], C.byref(p), accum[j].data_ptr(), streams[j].cuda_stream) == 0, rt.last_error()
        assert rt.lib.rt_resolve(C.byref(p), accum[j].data_ptr(), None, image[j].data_ptr(), None, streams[j].cuda_stream) == 0


for world in (1, 2, 4, 8):
    rank = world // 2
    p = abi.RT_Render_Params(w, h, 256, b, 0x1234ABCD, rank, world, 0, 0)
    res = []
    for two in (False, True):
        for i in range(4):
            frame(i, p, two)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            frame(i, p, two)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / K * 1e3)
    ref = image[0].clone()
    print(f"world {world} rank {rank}: one stream {res[0]:7.3f} ms/frame   two streams {res[1]:7.3f} ms/frame   ({100 * (res[0] / res[1] - 1):+.1f} %)   "
          f"images equal: {bool(torch.equal(image[0], image[1]))}", flush=True)

#!/usr/bin/env python3
"""GPU experiment: frames in flight.  N frames of one view through the blocking rt_render_frame() against the same N through
rt_frame_begin() / rt_frame_end() with two frames on the GPU (begin 0, begin 1, end 0, begin 2, end 1 ...): host wall time per
frame, scene resident, image copied to the host every frame on both sides.  `static` = rt_scene_set_static (no full content check
per frame: 0.9 ms of host time on the helmet, which the blocking path hides behind the kernel and the pipelined one behind the
other frame).  Prints a markdown table.

    python tools/exp_frames.py [frames] [reps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                      # noqa: E402
import raytracing_c_amd as rt                           # noqa: E402
from raytracing_c_amd.configs import load_config        # noqa: E402
from raytracing_c_amd.scene import make_image           # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
depth = int(os.environ.get("EXP_DEPTH", "2"))            # frames in flight (a library built with -DRT_FRAME_LANES=3 takes 3)
if os.environ.get("EXP_TORCH"):                        # a process that also runs torch streams (bench.py): do the lanes still overlap?
    import torch
    _keep_streams = [torch.cuda.Stream() for _ in range(int(os.environ["EXP_TORCH"]))]
    _x = torch.zeros(1 << 20, device="cuda")
    for st in _keep_streams:
        with torch.cuda.stream(st):
            _x += 1
    torch.cuda.synchronize()
assert rt.lib.rt_init(0) == 0, rt.last_error()

JOBS = [("driver default frame (driver.c:733-742) on the helmet", "helmet", 1024, 1024, 16, 8),
        ("helmet 1024^2, 64 spp", "helmet", 1024, 1024, 64, 8),
        ("BASELINE config #1", "spheres", 256, 256, 16, 4),
        ("BASELINE config #2", "quad", 512, 512, 64, 4),
        ("tower 640x360, 16 spp, 12 bounces", "tower", 640, 360, 16, 12),
        ("BASELINE config #3", "helmet", 1920, 1080, 256, 8)]


def blocking(hs, imgs, n, s, b):
    t0 = time.perf_counter()
    for f in range(n):
        img = imgs[f % len(imgs)][1]
        assert rt.lib.rt_render_frame(C.byref(hs.scene), C.byref(img), s, b, None, None) == 0, rt.last_error()
    return (time.perf_counter() - t0) * 1e3 / n


def pipelined(hs, imgs, n, s, b):
    t0 = time.perf_counter()
    pending = []
    for f in range(n):
        if len(pending) == depth:
            assert rt.lib.rt_frame_end(pending.pop(0)) == 0, rt.last_error()
        t = rt.lib.rt_frame_begin(C.byref(hs.scene), C.byref(imgs[f % len(imgs)][1]), s, b)
        assert t >= 0, rt.last_error()
        pending.append(t)
    while pending:
        assert rt.lib.rt_frame_end(pending.pop(0)) == 0, rt.last_error()
    return (time.perf_counter() - t0) * 1e3 / n


print("| frame | blocking, ms per frame | two frames in flight | | blocking, static scene | two in flight, static scene | |")
print("|---|---|---|---|---|---|---|")
if os.environ.get("EXP_ONLY"):
    JOBS = [JOBS[int(k)] for k in os.environ["EXP_ONLY"].split(",")]
for label, cfg, w, h, s, b in JOBS[:int(os.environ.get("EXP_JOBS", "99"))]:
    hs, _ = load_config(cfg)
    n = frames if w * h * s < 2e8 else max(6, frames // 5)
    imgs = []
    for k in range(4):
        out = np.zeros((h, w, 3), np.uint8)
        img, keep = make_image(out)
        img.pixels.data = out.ctypes.data
        imgs.append((out, img, keep))
    cols = []
    for static in (0, 1):
        rt.lib.rt_scene_set_static(C.byref(hs.scene), static)
        blocking(hs, imgs, 4, s, b)
        ref = imgs[1][0].copy()
        pipelined(hs, imgs, 4, s, b)
        assert all(np.array_equal(ref, im[0]) for im in imgs), "pipelined frame differs from the blocking one"
        bl, pl = [], []
        for r in range(reps):
            bl.append(blocking(hs, imgs, n, s, b))
            pl.append(pipelined(hs, imgs, n, s, b))
        cols += [f"{min(bl):.3f} - {max(bl):.3f}", f"{min(pl):.3f} - {max(pl):.3f}", f"{(np.median(pl) / np.median(bl) - 1) * 100:+.1f} %"]
    rt.lib.rt_scene_set_static(C.byref(hs.scene), 0)
    rt.lib.rt_scene_invalidate(C.byref(hs.scene))
    print(f"| {label}: {w}x{h}, {s} spp, {b} bounces, {n} frames | " + " | ".join(cols) + " |", flush=True)

#!/usr/bin/env python3
"""Renders the five BASELINE.json configurations on ONE MI355X and prints a markdown table
(kernel time from HIP events, rays counted in-kernel).  Output committed as profiles/<tag>_configs.md."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                           # noqa: E402
import raytracing_c_amd as rt                          # noqa: E402
from raytracing_c_amd import ctypes_abi as abi         # noqa: E402
from raytracing_c_amd.configs import CONFIGS, load_config   # noqa: E402

assert rt.lib.rt_init(0) == 0
print("| config | scene | frame | kernel ms (mean of 3, after 1 warm-up) | Mray/s | Msample/s | rays/path | nodes/ray | leaves/ray | shades/ray | B/ray |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for i, name in enumerate(["spheres", "quad", "helmet", "tower", "helmet4k"]):
    hs, cfg = load_config(name)
    w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, 0, 0)
    for rep in range(4):
        if rep == 1:
            rt.lib.rt_kernel_timing_reset()
        accum.zero_()
        assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
    torch.cuda.synchronize()
    ms = rt.lib.rt_kernel_timing_mean_ms(None)
    c = rt.render.get_counters()
    print(f"| #{i + 1} | {cfg['asset']} | {w}x{h}, {s} spp, {b} bounces | {ms:.3f} | {c.rays / ms / 1e3:.0f} | {w * h * s / ms / 1e3:.0f} | "
          f"{c.rays / c.paths:.3f} | {c.node_visits / c.rays:.3f} | {c.leaf_visits / c.rays:.3f} | {c.shades / c.rays:.3f} | {c.bytes_per_ray():.0f} |", flush=True)
    rt.lib.rt_scene_release(d)
    del accum

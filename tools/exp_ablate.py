#!/usr/bin/env python3
"""GPU experiment: kernel time of helmet 1080p/256spp under path-length / shader ablations."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                           # noqa: E402
import raytracing_c_amd as rt                          # noqa: E402
from raytracing_c_amd import ctypes_abi as abi         # noqa: E402
from raytracing_c_amd.configs import load_config       # noqa: E402

assert rt.lib.rt_init(0) == 0
for name, shader in (("helmet", "disney"), ("helmet", "debug"), ("tower", "disney"), ("spheres", "disney"), ("quad", "disney")):
    hs, cfg = load_config(name, shader=shader)
    w, h, s = 1920, 1080, 256 if name == "helmet" else 64
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    for b in ((1, 2, 8) if name == "helmet" else (cfg["max_bounces"],)):
        p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, 0, 0)
        rt.lib.rt_kernel_timing_reset()
        for _ in range(2):
            accum.zero_()
            assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
        torch.cuda.synchronize()
        ms = rt.lib.rt_kernel_timing_mean_ms(None)
        c = rt.render.get_counters()
        print(f"{name:8s} {shader:6s} {w}x{h} {s}spp b={b:2d}: {ms:8.2f} ms  rays {c.rays/1e6:8.1f}M  {c.rays/ms/1e3:8.1f} Mray/s  "
              f"N/ray {c.node_visits/c.rays:.2f} L/ray {c.leaf_visits/c.rays:.2f} H/ray {c.shades/c.rays:.3f}", flush=True)
    rt.lib.rt_scene_release(d)

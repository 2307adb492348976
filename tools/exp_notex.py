#!/usr/bin/env python3
"""GPU experiment: helmet frame with and without its textures (upper bound of what texture fetches cost)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracing_c_amd as rt
from raytracing_c_amd import ctypes_abi as abi
from raytracing_c_amd.configs import load_config
assert rt.lib.rt_init(0) == 0
for strip in (False, True):
    hs, cfg = load_config("helmet")
    if strip:
        for m in hs.materials:
            m.texture_albedo = None; m.texture_normal = None; m.texture_metal_roughness = None; m.texture_emission = None
            m.roughness = 0.4; m.metalness = 0.5
    w, h, s, b = 1920, 1080, 256, 8
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, 0, 0)
    for rep in range(4):
        if rep == 1: rt.lib.rt_kernel_timing_reset()
        accum.zero_()
        assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ms = rt.lib.rt_kernel_timing_mean_ms(None); c = rt.render.get_counters()
    print(f"textures stripped={strip}: {ms:.2f} ms rays {c.rays/1e6:.1f}M {c.rays/ms/1e3:.0f} Mray/s shades/ray {c.shades/c.rays:.3f} N/ray {c.node_visits/c.rays:.2f}")
    rt.lib.rt_scene_release(d)

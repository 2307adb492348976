#!/usr/bin/env python3
"""Workgroup-size sweep (diagnostic library: RT_WG_WAVES = 8 / 12 / 16 overrides the library's choice): kernel ms of a list of
launches per workgroup size.  Evidence for the selection rule of rt_api.cpp (profiles/r05_small_launch.md section 3).
    RT_LIB_PATH=raytracing_c_amd/librt_hip_diag.so python tools/exp_wg.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                      # noqa: E402
import torch                                            # noqa: E402
import raytracing_c_amd as rt                           # noqa: E402
from raytracing_c_amd import ctypes_abi as abi          # noqa: E402
from raytracing_c_amd.configs import ASSETS, load_config  # noqa: E402
from raytracing_c_amd.loaders import load_model         # noqa: E402

JOBS = [("quad", 256, 256, 64, 4), ("quad", 512, 512, 64, 4), ("quad", 768, 768, 64, 4), ("quad", 1024, 1024, 64, 4),
        ("fov_test.obj", 512, 512, 16, 8), ("fov_test.obj", 1024, 1024, 64, 8), ("sheen.glb", 512, 512, 16, 8), ("sheen.glb", 1024, 1024, 64, 8),
        ("spheres", 1024, 1024, 16, 4), ("spheres", 1024, 1024, 64, 4), ("tower", 1024, 1024, 16, 12), ("tower", 1920, 1080, 64, 12),
        ("helmet", 768, 768, 16, 8), ("helmet", 1024, 1024, 32, 8)]

assert rt.lib.rt_init(0) == 0
scenes = {}
print("| scene | frame | depth | wave-fulls per slot | 8 waves | 12 | 16 | library's choice |")
print("|---|---|---|---|---|---|---|---|")
for (name, w, h, s, b) in JOBS:
    if name not in scenes:
        hs = load_model(os.path.join(ASSETS, name)) if "." in name else load_config(name)[0]
        scenes[name] = (hs, rt.lib.rt_scene_upload(C.byref(hs.scene)))
    hs, d = scenes[name]
    accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, 0, 0)
    row = []
    for wg in ("8", "12", "16", ""):
        if wg:
            os.environ["RT_WG_WAVES"] = wg
        else:
            os.environ.pop("RT_WG_WAVES", None)
        ms = []
        for i in range(5):
            accum.zero_()
            assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
            torch.cuda.synchronize()
            ms.append(float(rt.lib.rt_last_kernel_ms()))
        row.append(float(np.mean(ms[2:])))
    pslot = w * h * s / (256 * 16 * 64)
    print(f"| {name} | {w}x{h}, {s} spp, {b} bounces | {hs.depth} | {pslot:.0f} | {row[0]:.3f} | {row[1]:.3f} | {row[2]:.3f} | {row[3]:.3f} |", flush=True)

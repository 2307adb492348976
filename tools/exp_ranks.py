#!/usr/bin/env python3
"""GPU experiment: per-rank work of the chunk partition (rays, node/leaf visits, shades, kernel ms)."""
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
hs, cfg = load_config("helmet")
w, h, b = 1920, 1080, 8
d = rt.lib.rt_scene_upload(C.byref(hs.scene))
accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rank in range(world):
    p = abi.RT_Render_Params(w, h, 256, b, 0x1234ABCD, rank, world, 0, 0)
    for rep in range(4):
        if rep == 1:
            rt.lib.rt_kernel_timing_reset()
        accum.zero_()
        assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
    torch.cuda.synchronize()
    ms = rt.lib.rt_kernel_timing_mean_ms(None)
    c = rt.render.get_counters()
    print(f"rank {rank}/{world}: {ms:7.3f} ms  rays {c.rays / 1e6:7.2f} M  nodes {c.node_visits / 1e6:7.1f} M  leaves {c.leaf_visits / 1e6:6.1f} M  "
          f"shades {c.shades / 1e6:6.2f} M  bg {c.backgrounds / 1e6:6.2f} M  -> {c.rays / ms / 1e3:7.0f} Mray/s", flush=True)

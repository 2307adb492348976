#!/usr/bin/env python3
"""GPU experiment (build with -DRT_EXP_NODESTATS): how many node / leaf blocks of the tile-stream kernel are run by
camera rays of ONE pixel pair on ONE node -- the blocks a pixel-pyramid cull could serve."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                            # noqa: E402
import raytracing_c_amd as rt                           # noqa: E402
from raytracing_c_amd import ctypes_abi as abi          # noqa: E402
from raytracing_c_amd.configs import load_config        # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
assert rt.lib.rt_init(0) == 0
hs, cfg = load_config(name)
w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]
d = rt.lib.rt_scene_upload(C.byref(hs.scene))
accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, 0, 0)
for it in range(2):
    accum.zero_()
    assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
    torch.cuda.synchronize()
st = (C.c_uint64 * 32)()
assert rt.lib.rt_get_sched_stats(st) == 0
x = list(st)[:16]
print(f"{name}: kernel {rt.lib.rt_last_kernel_ms():.2f} ms")
blocks, lanes, cam, grp, full, half, nocam, grt, fullt = x[:9]
print(f"  node blocks {blocks/1e6:.2f} M, lanes/block {lanes/max(blocks,1):.1f}, camera-ray lanes {cam/max(lanes,1):.3f} of lanes")
print(f"    first camera lane's (node, pixel pair) group: {grp/max(lanes,1):.3f} of lanes; blocks it fills entirely {full/max(blocks,1):.3f}"
      f"; at least half {half/max(blocks,1):.3f}; blocks without camera rays {nocam/max(blocks,1):.3f}")
print(f"    same with the whole tile as the group: {grt/max(lanes,1):.3f} of lanes; fills the block {fullt/max(blocks,1):.3f}")
print("    pyramid-culled node blocks by surviving children 0..4, then >4 (full block): "
      + " ".join(f"{v/max(blocks,1):.3f}" for v in x[9:15]))

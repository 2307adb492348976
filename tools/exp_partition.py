#!/usr/bin/env python3
"""GPU experiment: per-chunk cost of a frame (one launch per 32x32 chunk: world = n_chunks) and the rank imbalance of
candidate chunk -> rank mappings for 2, 4, 8 GPUs."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                      # noqa: E402
import torch                                            # noqa: E402
import raytracing_c_amd as rt                           # noqa: E402
from raytracing_c_amd import ctypes_abi as abi          # noqa: E402
from raytracing_c_amd.configs import load_config        # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
assert rt.lib.rt_init(0) == 0
hs, cfg = load_config(name)
w, h, b = cfg["width"], cfg["height"], cfg["max_bounces"]
spp = 16
d = rt.lib.rt_scene_upload(C.byref(hs.scene))
accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
n = rt.lib.rt_chunk_count(w, h)
cx_n = (w + 31) // 32
cost = np.zeros(n)
for c in range(n):
    p = abi.RT_Render_Params(w, h, spp, b, 0x1234ABCD, c, n, 0, 0)
    assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
    k = rt.render.get_counters()
    cost[c] = 340 * k.node_visits + 650 * k.leaf_visits + 2200 * k.shades + 450 * k.backgrounds + 220 * k.paths
np.save(os.path.join(ROOT, "gpurun_out", f"chunk_cost_{name}.npy"), cost)
cx, cy = np.arange(n) % cx_n, np.arange(n) // cx_n
print(f"{name}: {n} chunks, cost max/mean {cost.max() / cost.mean():.2f}")
for world in (2, 4, 8):
    res = []
    for label, rank in [("c % world", np.arange(n) % world)] + [(f"(cx + {B} cy) % world", (cx + B * cy) % world) for B in range(1, world)]:
        load = np.bincount(rank, weights=cost, minlength=world)
        res.append((load.max() / load.mean(), label))
    print(f"world {world}: " + "   ".join(f"{lab}: {imb:.3f}" for imb, lab in res))

#!/usr/bin/env python3
"""GPU experiment: kernel time vs amount of work (samples, image partition) -- what an 8-GPU rank will see."""
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


def run(s, rank=0, world=1, slab=0, reps=5):
    p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, rank, world, slab, 0)
    rt.lib.rt_kernel_timing_reset()
    for _ in range(reps):
        accum.zero_()
        assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
    torch.cuda.synchronize()
    ms = rt.lib.rt_kernel_timing_mean_ms(None)
    c = rt.render.get_counters()
    return ms, c.rays


run(256)
if os.environ.get("RT_EXP") == "slab":      # best slab per partition size
    for world in (1, 2, 4, 8):
        for slab in (4, 8, 16, 32, 64):
            if world == 1 and slab < 16 or world >= 4 and slab > 32:
                continue
            ms, rays = run(256, world // 2, world, slab)
            print(f"world {world} rank {world // 2} slab {slab:3d}: {ms:8.3f} ms", flush=True)
    sys.exit(0)
for s in (256, 128, 64, 32, 16, 8):
    ms, rays = run(s)
    print(f"samples {s:4d} world 1: {ms:8.3f} ms  {rays/ms/1e3:8.1f} Mray/s", flush=True)
for world in (2, 4, 8):
    worst = 0
    tot = 0
    per = []
    for rank in range(world):
        ms, rays = run(256, rank, world)
        worst = max(worst, ms)
        tot += rays
        per.append((ms, rays))
    print(f"  world {world} per rank: " + "  ".join(f"{m:.2f} ms/{r / 1e6:.1f} Mray" for m, r in per), flush=True)
    print(f"samples  256 world {world}: slowest rank {worst:8.3f} ms -> {tot/worst/1e3:8.1f} Mray/s aggregate if ranks ran in parallel", flush=True)
for slab in (4, 8, 16, 32):
    ms, rays = run(256, 3, 8, slab)
    print(f"world 8 rank 3 slab {slab:3d}: {ms:8.3f} ms", flush=True)
for wpc in (4, 8, 12, 16):
    os.environ["RT_WAVES_PER_CU"] = str(wpc)
    ms, rays = run(256, 3, 8)
    print(f"world 8 rank 3 waves/CU {wpc:2d}: {ms:8.3f} ms", flush=True)

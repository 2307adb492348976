#!/usr/bin/env python3
"""GPU experiment: when do the 4096 persistent waves start and finish (diagnostic kernel, RT_KERNEL=4)?"""
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

assert rt.lib.rt_init(0) == 0
hs, cfg = load_config("helmet")
w, h, b = 1920, 1080, 8
d = rt.lib.rt_scene_upload(C.byref(hs.scene))
accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
os.environ["RT_KERNEL"] = os.environ.get("RT_EXP_KERNEL", "4")
os.environ["RT_WAVE_TIMES"] = "1"
for (s, rank, world, slab) in ((256, 0, 1, 0), (32, 0, 1, 0), (256, 3, 8, 0), (256, 3, 8, 8)):
    p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, rank, world, slab, 0)
    for rep in range(3):
        accum.zero_()
        assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
        torch.cuda.synchronize()
    ms = rt.lib.rt_last_kernel_ms()
    buf = np.zeros((65536, 3), np.uint64)
    n = rt.lib.rt_get_wave_times(buf.ctypes.data, 65536)
    t = buf[:n].astype(np.float64)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0          # us
    end = (t[:, 1] - t0) / 100.0
    q = np.percentile(end, [0, 1, 10, 50, 90, 99, 100])
    idle = (end.max() - end).sum() / (end.max() * n)        # share of wave-time between a wave's exit and the kernel's end
    if os.environ["RT_KERNEL"] == "5":
        raw = buf[:n, 2]
        t[:, 2] = (raw & np.uint64(0xFFFF)).astype(np.float64)
        grab = (raw >> np.uint64(16)).astype(np.float64) / 100.0 + start       # us: last successful grab of units
        g = np.percentile(grab, [0, 50, 90, 99, 100])
        after = end - grab
        a = np.percentile(after, [0, 50, 90, 99, 100])
        late = np.argsort(end)[-5:]
        print(f"   last grab (us) min {g[0]:.0f} p50 {g[1]:.0f} p90 {g[2]:.0f} p99 {g[3]:.0f} max {g[4]:.0f}; time from last grab to exit: "
              f"min {a[0]:.0f} p50 {a[1]:.0f} p90 {a[2]:.0f} p99 {a[3]:.0f} max {a[4]:.0f}; the 5 last waves: "
              + ", ".join(f"grab {grab[i]:.0f} end {end[i]:.0f} tiles {t[i, 2]:.0f}" for i in late))
    print(f"S={s} rank {rank}/{world} slab {slab}: kernel {ms:.2f} ms, waves {n}; start spread {start.max():.0f} us; "
          f"end percentiles (us) min {q[0]:.0f} p1 {q[1]:.0f} p10 {q[2]:.0f} p50 {q[3]:.0f} p90 {q[4]:.0f} p99 {q[5]:.0f} max {q[6]:.0f}; "
          f"items/wave min {t[:,2].min():.0f} mean {t[:,2].mean():.1f} max {t[:,2].max():.0f}; busy mean {(end-start).mean():.0f} us; "
          f"tail idle {idle * 100:.1f} %", flush=True)

#!/usr/bin/env python3
"""GPU experiment: block-execution statistics of the diagnostic kernel (RT_KERNEL=4) for one config."""
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

NAMES = ["shade", "env", "regen", "leaf_uniform", "leaf_lane", "node_uniform", "node_lane", "pop"]
COST = {"shade": 2200, "env": 480, "regen": 220, "leaf_uniform": 700, "leaf_lane": 640, "node_uniform": 330,
        "node_lane": 330, "pop": 60}       # static VALU instructions per execution (ISA counts)

name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
assert rt.lib.rt_init(0) == 0
hs, cfg = load_config(name)
w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]
d = rt.lib.rt_scene_upload(C.byref(hs.scene))
accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
SLAB = int(os.environ.get("RT_EXP_SLAB", "0"))
for env in ({},) if SLAB else ({}, {"RT_SCHED_THRESH": "32"}, {"RT_SCHED_THRESH": "56"}):
    for k, v in env.items():
        os.environ[k] = v
    os.environ["RT_KERNEL"] = os.environ.get("RT_EXP_KERNEL", "4")
    p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, SLAB, 0)
    accum.zero_()
    assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
    torch.cuda.synchronize()
    st = (C.c_uint64 * 32)()
    assert rt.lib.rt_get_sched_stats(st) == 0
    ms = rt.lib.rt_last_kernel_ms()
    tot = 0
    rows = []
    for i, n in enumerate(NAMES):
        ex, ln = st[2 * i], st[2 * i + 1]
        est = ex * COST[n]
        tot += est
        rows.append((n, ex, ln, ln / max(ex, 1), est))
    print(f"--- {name} {env} kernel {ms:.2f} ms (diagnostic build) est. VALU wave-instr {tot/1e9:.2f} G")
    for n, ex, ln, avg, est in rows:
        print(f"  {n:13s} runs {ex/1e6:9.2f} M  lanes/run {avg:5.1f}  est {est/1e9:6.2f} G ({100*est/tot:4.1f} %)")
    tot_cyc = st[24]
    names = {16: "S blocks with shading", 17: "S blocks (environment + regeneration only)", 19: "leaf blocks", 21: "node blocks", 23: "pop loops"}
    acc = 0
    for i, nm in names.items():
        acc += st[i]
        print(f"  cycles in {nm:45s} {100 * st[i] / max(tot_cyc, 1):5.1f} % of the wave time")
    print(f"  cycles in scheduling / dequeue / flush (rest)            {100 * (tot_cyc - acc) / max(tot_cyc, 1):5.1f} %")
    for k in env:
        os.environ.pop(k)

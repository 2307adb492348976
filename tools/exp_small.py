#!/usr/bin/env python3
"""Small-launch ledger: where the time of a launch goes whose tiles are short (few samples per pixel) or few.

For each job -- the reference driver's default frame (driver.c:733-742: 1024 x 1024, 16 spp, 8 bounces, here on the helmet),
BASELINE config #1, rank 0 of the 8-way partition of config #3, and the same 1024 x 1024 frame at more samples -- it
prints one JSON line with
  * kernel ms of launch 1 (no cost order yet) and the mean of launches 3..5 (ordered by the previous launch's costs),
  * from the wave timeline (RT_WAVE_TIMES=1, diagnostic library): when waves start, take their last units and end; the
    share of all wave-time that lies between a wave's exit and the end of the launch ("tail idle"),
  * from a -DRT_LEDGER=2 library (`DIAG=1 tools/build_variant.sh ledger2 -DRT_LEDGER=2`): shader-clock cycles per kind of
    block incl. the tree copy, the join scans and the DRAIN of a tile (last unit handed out -> last path of it ended).

    RT_LIB_PATH=raytracing_c_amd/librt_hip_diag.so RT_WAVE_TIMES=1 python tools/exp_small.py
    RT_LIB_PATH=tools/exp/librt_ledger2.so python tools/exp_small.py
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                      # noqa: E402

from tools.exp_ledger import LG_NAMES as _OLD          # noqa: E402

LG_NAMES = list(_OLD) + ["CYC_COPY", "CYC_JOIN", "CYC_DRAIN", "DRAIN_X", "DRAIN_L", "CYC_FLUSH"]

# (label, config, width, height, spp, bounces, rank, world)
JOBS = [("driver default frame", "helmet", 1024, 1024, 16, 8, 0, 1),
        ("config #1", "spheres", 256, 256, 16, 4, 0, 1),
        ("config #3 rank 0 of 8", "helmet", 1920, 1080, 256, 8, 0, 8),
        ("config #3 rank 4 of 8", "helmet", 1920, 1080, 256, 8, 4, 8),
        ("1024^2 x 32 spp", "helmet", 1024, 1024, 32, 8, 0, 1),
        ("1024^2 x 64 spp", "helmet", 1024, 1024, 64, 8, 0, 1),
        ("1024^2 x 256 spp", "helmet", 1024, 1024, 256, 8, 0, 1),
        ("config #2", "quad", 512, 512, 64, 4, 0, 1),
        ("config #3", "helmet", 1920, 1080, 256, 8, 0, 1),
        # (only with RT_SMALL_JOBS: frames between config #1 and the default frame, for the workgroup-size crossover)
        ("helmet 256x144 x 16", "helmet", 256, 144, 16, 8, 0, 1),
        ("helmet 512^2 x 16", "helmet", 512, 512, 16, 8, 0, 1),
        ("helmet 512^2 x 64", "helmet", 512, 512, 64, 8, 0, 1),
        ("spheres 512^2 x 16", "spheres", 512, 512, 16, 4, 0, 1),
        ("tower 640x360 x 16", "tower", 640, 360, 16, 12, 0, 1),
        ("config #4", "tower", 1920, 1080, 512, 12, 0, 1),
        ("spheres 1024^2 x 64", "spheres", 1024, 1024, 64, 4, 0, 1)]
EXTRA = 7


def main():
    import torch
    import raytracing_c_amd as rt
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    has_ledger = hasattr(rt.lib, "rt_get_ledger") and "ledger" in os.environ.get("RT_LIB_PATH", "")
    has_waves = hasattr(rt.lib, "rt_get_wave_times") and os.environ.get("RT_WAVE_TIMES")
    only = os.environ.get("RT_SMALL_JOBS")
    scenes = {}
    for (label, name, w, h, s, b, rank, world) in JOBS:
        if only and label not in only.split(";"):
            continue
        if not only and label in [j[0] for j in JOBS[-EXTRA:]]:
            continue
        if name not in scenes:
            hs, _ = load_config(name)
            d = rt.lib.rt_scene_upload(C.byref(hs.scene))
            assert d, rt.last_error()
            scenes[name] = (hs, d)
        hs, d = scenes[name]
        accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
        p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, rank, world, 0, 0)
        ms = []
        for i in range(6):
            accum.zero_()
            assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
            torch.cuda.synchronize()
            ms.append(float(rt.lib.rt_last_kernel_ms()))
        c = rt.render.get_counters()
        out = dict(job=label, frame=[name, w, h, s, b, rank, world], ms_first=round(ms[0], 4), ms_ordered=round(float(np.mean(ms[2:])), 4),
                   ms_all=[round(x, 4) for x in ms], paths=c.paths, rays=c.rays, Mray_s=round(c.rays / np.mean(ms[2:]) / 1e3, 1))
        if has_waves:
            buf = np.zeros((65536, 3), np.uint64)
            n = rt.lib.rt_get_wave_times(buf.ctypes.data, 65536)
            t = buf[:n].astype(np.float64)
            live = t[:, 1] > 0
            t = t[live]
            raw = buf[:n, 2][live]
            t0 = t[:, 0].min()
            start = (t[:, 0] - t0) / 100.0                   # us (100 MHz)
            end = (t[:, 1] - t0) / 100.0
            tiles = (raw & np.uint64(0xFFFF)).astype(np.float64)
            grab = (raw >> np.uint64(16)).astype(np.float64) / 100.0 + start
            grab = np.where(tiles > 0, grab, start)
            pct = lambda a: [round(float(x), 1) for x in np.percentile(a, [0, 10, 50, 90, 99, 100])]
            span = end.max()
            out["waves"] = dict(n=int(len(t)), start_us=pct(start), last_grab_us=pct(grab), end_us=pct(end), after_grab_us=pct(end - grab),
                                tiles_per_wave=pct(tiles), span_us=round(float(span), 1),
                                tail_idle=round(float((span - end).sum() / (span * len(t))), 4),
                                after_last_grab_share=round(float((end - grab).sum() / (span * len(t))), 4),
                                start_share=round(float(start.sum() / (span * len(t))), 4))
        if has_ledger:
            buf = (C.c_uint64 * len(LG_NAMES))()
            assert rt.lib.rt_get_ledger(buf, len(LG_NAMES)) == 0, rt.last_error()
            lg = {n: int(v) for n, v in zip(LG_NAMES, buf)}
            wave = max(lg["CYC_WAVE"], 1)
            out["cycles_share"] = {k[4:].lower(): round(lg[k] / wave, 4) for k in lg if k.startswith("CYC_") and k != "CYC_WAVE"}
            out["counts"] = {k: lg[k] for k in ("TILE_X", "JOIN_X", "FLUSH_X", "GRAB_X", "DRAIN_X", "DRAIN_L", "S_ITER", "SKY_X", "SHADE_X", "SHADE_L",
                                                "ENV_X", "ENV_L", "ROUND_X", "TRAV_CALLS", "LEAF_X", "LEAF_L", "NFULL_X", "NFULL_L", "POP_X", "POP_L")}
            out["wave_cycles_x16"] = lg["CYC_WAVE"]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

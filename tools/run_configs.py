#!/usr/bin/env python3
"""The five BASELINE.json configurations on ONE MI355X, GPU beside CPU: kernel time from HIP events
and rays counted in-kernel for the GPU; for the CPU the oracle (the repo's CPU restatement of the reference path -- its 8-wide
AVX2 form and its scalar form -- rebuilt -O3 -march=native on this host) with all host threads and with one -- configs #4 / #5 at samples / 16, stated in the table,
as BASELINE.md allows.  Prints markdown, committed as profiles/<tag>_configs.md.
    python tools/run_configs.py [--no-cpu]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                           # noqa: E402
import raytracing_c_amd as rt                          # noqa: E402
from raytracing_c_amd import ctypes_abi as abi         # noqa: E402
from raytracing_c_amd.configs import load_config       # noqa: E402
from tests import _oracle                              # noqa: E402

assert rt.lib.rt_init(0) == 0
cpu = "--no-cpu" not in sys.argv
cores = len(os.sched_getaffinity(0))
try:
    q = open("/sys/fs/cgroup/cpu.max").read().split()
    if q[0] != "max":
        cores = max(1, min(cores, int(round(int(q[0]) / int(q[1])))))
except Exception:
    pass
lib = None
if cpu:
    out = os.path.join(tempfile.mkdtemp(prefix="oracle_native_"), "liboracle_native.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "MARCH=native", f"OUT={out}", out], stdout=subprocess.DEVNULL)
    lib = _oracle.load(out)
    assert lib.oracle_have_avx2(), "the host of a GPU box has AVX2 + FMA"
print(f"| config | scene | frame | GPU kernel ms (mean of 3 after 1 warm-up) | GPU Mray/s | GPU Msample/s | CPU sample | CPU s ({cores} threads, AVX2) | "
      f"CPU Mray/s ({cores} threads, AVX2) | CPU Msample/s | CPU Mray/s (1 thread, AVX2) | CPU Mray/s ({cores} threads / 1 thread, scalar form) | GPU / CPU | "
      f"rays/path | nodes/ray | leaves/ray | shades/ray | B/ray |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for i, name in enumerate(["spheres", "quad", "helmet", "tower", "helmet4k"]):
    hs, cfg = load_config(name)
    w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, 0, 0)
    for rep in range(4):
        if rep == 1:
            rt.lib.rt_kernel_timing_reset()
        accum.zero_()
        assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
    torch.cuda.synchronize()
    ms = rt.lib.rt_kernel_timing_mean_ms(None)
    c = rt.render.get_counters()
    rt.lib.rt_scene_release(d)
    del accum
    cpu_cols = "| - | - | - | - | - | - | - "
    if cpu:
        cs = s if i < 3 else max(1, s // 16)            # #4 / #5 at spp / 16, scaled linearly
        t0 = time.perf_counter()
        r = _oracle.render(hs, w, h, cs, b, n_threads=cores, lib=lib)
        dt = time.perf_counter() - t0
        one_s = max(1, min(cs, int(round(cs * 3.0 / max(dt * cores, 1e-3)))))      # about three seconds on one thread
        t0 = time.perf_counter()
        r1 = _oracle.render(hs, w, h, one_s, b, n_threads=1, lib=lib)
        dt1 = time.perf_counter() - t0
        mr = r["counters"]["rays"] / dt / 1e6
        # the scalar form of the checker on a third of the sample, for the table's last CPU column
        lib.oracle_set_simd(0)
        ss = max(1, cs // 3)
        t0 = time.perf_counter()
        rs = _oracle.render(hs, w, h, ss, b, n_threads=cores, lib=lib)
        dts = time.perf_counter() - t0
        t0 = time.perf_counter()
        rs1 = _oracle.render(hs, w, h, max(1, one_s // 2), b, n_threads=1, lib=lib)
        dts1 = time.perf_counter() - t0
        lib.oracle_set_simd(1)
        cpu_cols = (f"| {cs} of {s} spp | {dt:.2f} | {mr:.1f} | {w * h * cs / dt / 1e6:.1f} | {r1['counters']['rays'] / dt1 / 1e6:.2f} ({one_s} spp) | "
                    f"{rs['counters']['rays'] / dts / 1e6:.1f} / {rs1['counters']['rays'] / dts1 / 1e6:.2f} | {c.rays / ms / 1e3 / mr:.0f}x ")
    print(f"| #{i + 1} | {cfg['asset']} | {w}x{h}, {s} spp, {b} bounces | {ms:.3f} | {c.rays / ms / 1e3:.0f} | {w * h * s / ms / 1e3:.0f} "
          f"{cpu_cols}| {c.rays / c.paths:.3f} | {c.node_visits / c.rays:.3f} | {c.leaf_visits / c.rays:.3f} | {c.shades / c.rays:.3f} | "
          f"{c.bytes_per_ray():.0f} |", flush=True)
print(f"\nCPU = oracle/oracle.c, gcc -O3 -march=native, {cores} host threads of the GPU box: kind \"port-avx2\" = the reference's 8-wide AVX2 forms of "
      "ray_aabbs_hit_8 / ray_triangles_hit_8 / min_f32x8 (raytracer.c:15-32,84-230) restated in the checker (the reference itself cannot be built here), "
      "bit-identical to its scalar form (tests/test_oracle_simd.py), whose figures stand in the column beside.  GPU / CPU compares Mray/s at equal work per "
      "sample against the AVX2 figure (a reported baseline, not a target).")

#!/usr/bin/env python3
"""Measures the kernels and host stages around the path kernel on one MI355X and prints a
markdown table: resolve, untile, denoiser (HBM-bound by construction: GB/s against the 8 TB/s roof), lightmap_bake,
scene_init (CPU), scene upload, asset loading.  Output committed as profiles/<tag>_aux.md."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                     # noqa: E402
import torch                                           # noqa: E402
import raytracing_c_amd as rt                          # noqa: E402
from raytracing_c_amd import ctypes_abi as abi         # noqa: E402
from raytracing_c_amd.configs import load_config       # noqa: E402

assert rt.lib.rt_init(0) == 0
HBM = 8.0e12


def gpu_ms(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []
for w, h in ((1920, 1080), (3840, 2160)):
    g = torch.Generator(device="cuda").manual_seed(1)
    src = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    dst = torch.empty_like(src)
    ms = gpu_ms(lambda: rt.lib.rt_denoise(w, h, src.data_ptr(), dst.data_ptr(), None))
    b = w * h * 6
    rows.append((f"rt_denoise_kernel {w}x{h}", f"{ms * 1e3:.1f} us", f"{b / 1e6:.1f} MB (3 B in + 3 B out per pixel)",
                 f"{b / ms / 1e6:.0f} GB/s = {b / ms * 1e3 / HBM:.3f} of 8 TB/s"))
    accum = torch.randint(0, 2 ** 40, (h, w, 3), dtype=torch.int64, device="cuda", generator=g)
    img = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    p = abi.RT_Render_Params(w, h, 256, 8, 1, 0, 1, 0, 0)
    ms = gpu_ms(lambda: rt.lib.rt_resolve(C.byref(p), accum.data_ptr(), None, img.data_ptr(), None, None))
    b = w * h * 27
    rows.append((f"rt_resolve_kernel {w}x{h}", f"{ms * 1e3:.1f} us", f"{b / 1e6:.1f} MB (24 B in + 3 B out per pixel)",
                 f"{b / ms / 1e6:.0f} GB/s = {b / ms * 1e3 / HBM:.3f} of 8 TB/s"))
    world = 8
    n_chunks = rt.lib.rt_chunk_count(w, h)
    max_local = (n_chunks + world - 1) // world
    tiles = torch.randint(0, 256, (world, max_local, 3072), dtype=torch.uint8, device="cuda", generator=g)
    ms = gpu_ms(lambda: rt.lib.rt_untile(w, h, world, tiles.data_ptr(), img.data_ptr(), None))
    b = w * h * 6
    rows.append((f"rt_untile_kernel {w}x{h}, 8 ranks", f"{ms * 1e3:.1f} us", f"{b / 1e6:.1f} MB (3 B in + 3 B out per pixel)",
                 f"{b / ms / 1e6:.0f} GB/s = {b / ms * 1e3 / HBM:.3f} of 8 TB/s"))
    del src, dst, accum, img, tiles

for name in ("spheres", "tower", "helmet"):
    t0 = time.perf_counter()
    hs, cfg = load_config(name)
    t_load = time.perf_counter() - t0
    n_tris = int(np.count_nonzero(np.any(hs.soa_array() != 0, axis=0)))
    t_build = hs.scene_init_seconds
    t0 = time.perf_counter()
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0
    nbytes = rt.lib.rt_scene_device_bytes(d)
    rows.append((f"load + scene_init {cfg['asset']} ({n_tris} triangles)", f"{t_load * 1e3:.0f} ms (Python loader + C builder)"
                 + (f", scene_init alone {t_build * 1e3:.1f} ms" if t_build is not None else ""), "-", "CPU, once per scene"))
    rows.append((f"rt_scene_upload {cfg['asset']}", f"{t_up * 1e3:.1f} ms", f"{nbytes / 1e6:.1f} MB resident", f"{nbytes / t_up / 1e9:.1f} GB/s host->HBM incl. flattening"))
    rt.lib.rt_scene_release(d)

# lightmap_bake over the tower's UV layout
hs, cfg = load_config("tower")          # (helmet.glb keeps its V coordinates in [1, 2): nothing to bake)
lm = np.full((512, 512, 3), 7, np.uint8)
from raytracing_c_amd.scene import make_image          # noqa: E402
img, keep = make_image(lm)
rt.lib.rt_set_seed(0x1234ABCD)
t0 = time.perf_counter()
rt.lib.rt_clear_error()
rt.lib.lightmap_bake(C.byref(img), C.byref(hs.scene), 16)
if rt.last_error():
    print('lightmap_bake error:', rt.last_error(), file=sys.stderr)
t_bake = time.perf_counter() - t0
owned = int(np.count_nonzero((keep != 7).any(axis=2)))
rows.append(("lightmap_bake tower.obj 512x512, 16 samples x 8 bounces", f"{t_bake * 1e3:.1f} ms wall (host call)", f"{owned} baked texels",
             "second caller of the path loop; parity target, not a performance target"))

print("| stage | time | bytes | rate |")
print("|---|---|---|---|")
for r in rows:
    print("| " + " | ".join(r) + " |")

#!/usr/bin/env python3
"""GPU experiment: a camera that moves every frame.  The tile order of a launch comes from the rays per tile of the PREVIOUS launch
(schedule feedback, rt_api.cpp); this renders N frames of the helmet, each from a camera rotated about the model by `step` degrees
more than the last, and prints the path kernel's ms per frame -- with a library that keys the feedback on the exact view the order
of a moved camera is the identity, with one that keys it on the frame's shape alone it is the previous view's.

    RT_LIB_PATH=... python tools/exp_moving.py [step degrees] [frames] [width height spp]"""
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

step = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w, h, s = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080, 256)
assert rt.lib.rt_init(0) == 0, rt.last_error()
hs, _ = load_config("helmet")
d = rt.lib.rt_scene_upload(C.byref(hs.scene))
assert d, rt.last_error()
m0 = np.array([[hs.scene.camera.view_matrix.rows[r][c] for c in range(4)] for r in range(4)], np.float64)
accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
p = abi.RT_Render_Params(w, h, s, 8, 0x1234ABCD, 0, 1, 0, 0)
ms = []
fov = float(hs.scene.camera.fov)
for f in range(frames):
    a = np.radians(step * f)
    rot = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
    hs.set_camera((rot @ m0).astype(np.float32), fov)
    assert rt.lib.rt_set_camera(d, C.byref(hs.scene.camera)) == 0, rt.last_error()
    accum.zero_()
    assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
    torch.cuda.synchronize()
    ms.append(float(rt.lib.rt_last_kernel_ms()))
print(f"step {step} deg, {w}x{h} {s} spp: first {ms[0]:.3f} ms, frames 3.. mean {np.mean(ms[2:]):.3f} ms  ({' '.join(f'{x:.2f}' for x in ms)})")

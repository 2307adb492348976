#!/usr/bin/env python3
"""GPU experiment helper: kernel time of one config under runtime knobs (slab size, blocks per CU).
    python tools/exp_runtime.py [config] [--reps 3]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                           # noqa: E402
import raytracing_c_amd as rt                          # noqa: E402
from raytracing_c_amd import ctypes_abi as abi         # noqa: E402
from raytracing_c_amd.configs import load_config       # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "helmet"
    reps = 6
    assert rt.lib.rt_init(0) == 0
    hs, cfg = load_config(name)
    w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")

    def run(slab, bpc):
        os.environ["RT_BLOCKS_PER_CU"] = str(bpc)
        p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, slab, 0)
        rt.lib.rt_kernel_timing_reset()
        for _ in range(reps):
            accum.zero_()
            assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
        torch.cuda.synchronize()
        n = C.c_int32()
        return rt.lib.rt_kernel_timing_mean_ms(C.byref(n))

    run(64, 4)
    knobs = os.environ.get("RT_EXP", "auto,slab").split(",")
    if "auto" in knobs:     # the library's own choice of the item size (what bench.py runs); used by tools/exp_ab.sh
        print(f"slab auto: {run(0, 4):8.3f} ms", flush=True)
        print(f"slab auto: {run(0, 4):8.3f} ms", flush=True)
    if "kernel" in knobs:
        for kv in (1, 2, 3):
            os.environ["RT_KERNEL"] = str(kv)
            for th in ((32,) if kv == 1 else (32, 40, 48, 56)):
                os.environ["RT_SCHED_THRESH"] = str(th)
                for slab in (8, 16, 32):
                    print(f"kernel {kv} thresh {th:2d} slab {slab:3d}: {run(slab, 4):8.3f} ms", flush=True)
        os.environ.pop("RT_KERNEL")
        os.environ.pop("RT_SCHED_THRESH")
    if "thresh" in knobs:   # lanes waiting for the shade / environment / regenerate block that trigger it
        for th in (32, 40, 48, 56):
            os.environ["RT_SCHED_THRESH"] = str(th)
            print(f"S thresh {th:2d}: {run(0, 4):8.3f} ms", flush=True)
        os.environ.pop("RT_SCHED_THRESH")
    if "ldsn" in knobs:     # how many leading BVH nodes need to be in LDS
        for nl in (0, 9, 73, 105, 150, 200, 290, 100000):
            os.environ["RT_LDS_NODES"] = str(nl)
            print(f"lds nodes {nl:6d}: {run(16, 4):8.3f} ms", flush=True)
        os.environ.pop("RT_LDS_NODES")
    if "slab" in knobs:
        for slab in (4, 8, 16, 32, 64, 128, 256):
            print(f"slab {slab:4d}: {run(slab, 4):8.3f} ms", flush=True)
    c = rt.render.get_counters()
    print("rays", c.rays, "Mray/s at last run n/a")


if __name__ == "__main__":
    main()

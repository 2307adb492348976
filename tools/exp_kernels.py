#!/usr/bin/env python3
"""GPU experiment: A/B of kernel variants / knobs in ONE process on one box (interleaved, so box-to-box and clock
differences cancel).  Each arm = a label and environment settings read per launch by librt_hip.so (RT_KERNEL, ...).
Prints the mean kernel time of the whole helmet frame (config #3), of ranks 0,3,5 of the 8-GPU partition, and the
sha256 of the accumulation buffer (every arm must print the same one).
    python tools/exp_kernels.py "k3:RT_KERNEL=3" "k5:RT_KERNEL=5" ...      [RT_EXP_REPS=4] [RT_EXP_SPP=256]"""
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                           # noqa: E402
import raytracing_c_amd as rt                          # noqa: E402
from raytracing_c_amd import ctypes_abi as abi         # noqa: E402
from raytracing_c_amd.configs import load_config       # noqa: E402


def main():
    arms = []
    for a in sys.argv[1:]:
        label, _, env = a.partition(":")
        arms.append((label, dict(kv.split("=", 1) for kv in env.split(",") if kv)))
    reps = int(os.environ.get("RT_EXP_REPS", "4"))
    name = os.environ.get("RT_EXP_CONFIG", "helmet")
    assert rt.lib.rt_init(0) == 0
    hs, cfg = load_config(name)
    w, h, s, b = cfg["width"], cfg["height"], int(os.environ.get("RT_EXP_SPP", cfg["samples"])), cfg["max_bounces"]
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
    ranks = [int(r) for r in os.environ.get("RT_EXP_RANKS", "0,3,5").split(",") if r != ""]

    def run(rank, world, n):
        p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, rank, world, int(os.environ.get("RT_EXP_SLAB", "0")), 0)
        for i in range(n + 2):
            if i == 2:
                rt.lib.rt_kernel_timing_reset()
            accum.zero_()
            assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
        torch.cuda.synchronize()
        return rt.lib.rt_kernel_timing_mean_ms(None)

    for rnd in range(int(os.environ.get("RT_EXP_ROUNDS", "2"))):
        for label, env in arms:
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            # (the product library reads no experiment knob from the environment: the pipeline is chosen through the API)
            if hasattr(rt.lib, "rt_set_pipeline"):           # diagnostic library only (RT_LIB_PATH=.../librt_hip_diag.so)
                rt.lib.rt_set_pipeline(1 if os.environ.get("RT_PIPELINE") == "wf" else 0)
            full = run(0, 1, reps)
            digest = hashlib.sha256(accum.cpu().numpy().tobytes()).hexdigest()[:12]
            c = rt.render.get_counters()
            per_rank = [run(r, 8, reps) for r in ranks]
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            print(f"round {rnd} {label:14s} full {full:8.3f} ms  {c.rays / full / 1e3:8.0f} Mray/s   rank-of-8 "
                  + " ".join(f"{m:6.3f}" for m in per_rank) + f"   accum {digest} rays {c.rays}", flush=True)


if __name__ == "__main__":
    main()

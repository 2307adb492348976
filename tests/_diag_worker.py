#!/usr/bin/env python3
"""Worker of tests/test_gpu_diag.py and tests/test_gpu_contract_v1.py: runs in its OWN process with RT_LIB_PATH = librt_hip_diag.so
(or librt_hip_v1.so, the product under numeric contract v1) -- the diagnostic build carries the
superseded kernel generations and reads the RT_* experiment knobs from the environment at every launch) and renders a
list of jobs.  stdin: JSON list of {config, w, h, s, b, env: {...}, slabs: [...]}; stdout: one JSON line per job with the
sha256 of the radiance sums per slab and the counters of the last launch."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    jobs = json.load(sys.stdin)
    import numpy as np
    import torch
    import raytracing_c_amd as rt
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    assert os.path.basename(rt.native.LIB_PATH) in ("librt_hip_diag.so", "librt_hip_v1.so", "librt_hip_v2.so"), rt.native.LIB_PATH
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    scenes = {}
    for job in jobs:
        name = job["config"] + ":" + job.get("shader", "disney")
        if name not in scenes:
            hs, _ = load_config(job["config"], shader=job.get("shader", "disney"))
            d = rt.lib.rt_scene_upload(C.byref(hs.scene))
            assert d, rt.last_error()
            scenes[name] = (hs, d)
        hs, d = scenes[name]
        saved = {k: os.environ.get(k) for k in job["env"]}
        os.environ.update(job["env"])
        out = {"digests": [], "error": None}
        try:
            w, h, s, b = job["w"], job["h"], job["s"], job["b"]
            for slab in job.get("slabs", [0]):
                accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
                p = abi.RT_Render_Params(w, h, s, b, job.get("seed", 0x1234ABCD), 0, 1, slab, 0)
                if rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) != 0:
                    raise RuntimeError(rt.last_error())
                torch.cuda.synchronize()
                out["digests"].append(hashlib.sha256(accum.cpu().numpy().tobytes()).hexdigest())
            c = rt.render.get_counters()
            out["counters"] = [c.rays, c.node_visits, c.leaf_visits, c.shades]
            out["contract"] = int(rt.lib.rt_math_contract())
        except Exception as e:           # noqa: BLE001 -- reported to the parent, which fails the test
            out["error"] = repr(e)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

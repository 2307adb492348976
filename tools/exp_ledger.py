#!/usr/bin/env python3
"""Block ledger of the tile-stream path kernel: which block of rt_path_kernel_stream pays for the frame.

Runs a fixed list of launches (scenes x frame sizes x spp x bounce limits) through rt_render_accumulate and prints one JSON
line per launch: the kernel's ray counters and -- when the library is a -DRT_LEDGER build (tools/exp/librt_ledger.so,
`DIAG=1 tools/build_variant.sh ledger -DRT_LEDGER=1`) -- the LG_* slots of csrc/rt_dev.hip.h: executions and lanes of every
block.  The SAME list run with the product library under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU`
gives the VALU wave-instructions of every launch; tools/ledger_fit.py fits instructions per block execution
(non-negative least squares over the launches) and writes profiles/r04_blocks.md.

    RT_LIB_PATH=tools/exp/librt_ledger.so python tools/exp_ledger.py > gpurun_out/ledger/counts.jsonl
    rocprofv3 --pmc SQ_INSTS_VALU ... -- python3 tools/exp_ledger.py > gpurun_out/ledger/product.jsonl
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LG_NAMES = ["S_ITER", "ENV_X", "ENV_L", "SHADE_X", "SHADE_L", "PSTORE_X", "PSTORE_L", "PLOAD_X", "PLOAD_L", "ACCUM_X", "ACCUM_L",
            "REGEN_X", "REGEN_L", "START_X", "START_L", "PRIM_X", "PRIM_L", "TILE_X", "JOIN_X", "FLUSH_X", "GRAB_X", "ROUND_X", "TRAV_CALLS",
            "NFULL_X", "NFULL_L", "NFULL_CAM", "NGLOB_X", "NGLOB_L", "NEXACT_X", "NEXACT_L", "CULLMASK_X", "PYRCHK_X", "NODE_WAIT_L",
            "NFEW0_X", "NFEW1_X", "NFEW2_X", "NFEW3_X", "NFEW4_X", "NFEW0_L", "NFEW1_L", "NFEW2_L", "NFEW3_L", "NFEW4_L",
            "LEAF_X", "LEAF_L", "LEAF_CAM", "POP_X", "POP_L", "POP_UP_L", "POP_RETEST_L", "POP_CAM", "POP_UP_X", "POP_RETEST_X", "POP_DONE_L",
            "CYC_S", "CYC_NODE", "CYC_LEAF", "CYC_POP", "CYC_WAVE", "CYC_TILE", "SKY_X", "SKY_L", "CYC_SKY"]

# (config, shader, width, height, spp, bounces): a spread of block mixes -- no nodes at all (quad), environment-dominated
# (tower), deep bounce chains (helmet at 16 bounces), primary rays only (1 bounce, debug shader), small frames whose launch
# is mostly tile set-up and joins
JOBS = [("helmet", "disney", 1920, 1080, 256, 8),          # BASELINE configs[2]: the frame the table is about
        ("helmet", "disney", 1920, 1080, 64, 8), ("helmet", "disney", 1920, 1080, 64, 1), ("helmet", "disney", 1920, 1080, 64, 2),
        ("helmet", "disney", 1920, 1080, 64, 16), ("helmet", "disney", 960, 540, 256, 8), ("helmet", "disney", 960, 540, 64, 4),
        ("helmet", "debug", 1920, 1080, 64, 8), ("helmet", "disney", 480, 270, 32, 8), ("helmet", "disney", 3840, 2160, 16, 8),
        ("helmet", "disney", 1920, 1080, 16, 8), ("helmet", "disney", 1920, 1080, 8, 3),
        ("tower", "disney", 1920, 1080, 128, 12), ("tower", "disney", 1920, 1080, 64, 1), ("tower", "disney", 960, 540, 128, 4),
        ("tower", "debug", 1920, 1080, 64, 4), ("tower", "disney", 1920, 1080, 32, 24),
        ("spheres", "disney", 1024, 1024, 64, 4), ("spheres", "disney", 1024, 1024, 64, 1), ("spheres", "disney", 256, 256, 256, 8),
        ("spheres", "debug", 1024, 1024, 64, 4), ("spheres", "disney", 2048, 2048, 16, 16), ("spheres", "disney", 512, 512, 128, 2),
        ("quad", "disney", 512, 512, 64, 4), ("quad", "disney", 1024, 1024, 64, 1), ("quad", "disney", 2048, 2048, 16, 8),
        ("quad", "debug", 1024, 1024, 64, 4),
        ("helmet", "disney", 1280, 720, 128, 6), ("helmet", "disney", 640, 360, 512, 8), ("tower", "disney", 640, 360, 256, 12),
        ("spheres", "disney", 1920, 1080, 32, 6), ("helmet", "disney", 1920, 1080, 32, 5), ("helmet", "disney", 1920, 1080, 128, 12)]


def main():
    import torch
    import raytracing_c_amd as rt
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    has_ledger = hasattr(rt.lib, "rt_get_ledger")
    scenes = {}
    jobs = JOBS[:int(os.environ.get("RT_LEDGER_JOBS", len(JOBS)))]
    for (name, shader, w, h, s, b) in jobs:
        key = (name, shader)
        if key not in scenes:
            hs, _ = load_config(name, shader=shader)
            d = rt.lib.rt_scene_upload(C.byref(hs.scene))
            assert d, rt.last_error()
            scenes[key] = (hs, d)
        hs, d = scenes[key]
        accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
        p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, 0, 0)
        # two launches: the second one runs with the tile order the first one's costs produce, like every frame but the first
        for i in range(2):
            accum.zero_()
            assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
        torch.cuda.synchronize()
        c = rt.render.get_counters()
        out = dict(job=[name, shader, w, h, s, b], kernel_ms=float(rt.lib.rt_last_kernel_ms()), paths=c.paths, rays=c.rays,
                   node_visits=c.node_visits, leaf_visits=c.leaf_visits, shades=c.shades, backgrounds=c.backgrounds, textured=c.textured)
        if has_ledger:
            buf = (C.c_uint64 * len(LG_NAMES))()
            assert rt.lib.rt_get_ledger(buf, len(LG_NAMES)) == 0, rt.last_error()
            out["ledger"] = {n: int(v) for n, v in zip(LG_NAMES, buf)}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

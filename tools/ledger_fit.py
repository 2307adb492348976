#!/usr/bin/env python3
"""Fits VALU wave-instructions per block execution of rt_path_kernel_stream from the launches of tools/exp_ledger.py
(gpurun_out/<tag>/counts.jsonl: block executions of the -DRT_LEDGER build; pmc.csv: SQ_INSTS_VALU of the PRODUCT library for
the same launches) and writes profiles/<out>_blocks.md -- the per-block ledger of the product kernel.

A block costs the same number of wave-instructions whatever the number of lanes in it, so instructions = sum over blocks of
(cost x executions) is linear; the costs come from non-negative least squares over ~33 launches with different block mixes.
    python tools/ledger_fit.py ledger_a r04"""
import collections
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (key, label, executions, lanes key, static prior, lower, upper)
#   executions: a ledger slot, or a function of the launch record
#   prior: VALU instructions of the block in the ISA of the product kernel (tools/ledger_static.py: a -DRT_LEDGER_MARKS build has
#   comment markers at the block boundaries and is otherwise the same code); None = glue code without a clean region, fitted freely
def _disney_tex(c):
    return c["ledger"]["SHADE_X"] if (c["job"][1] == "disney" and c["textured"] > 0) else 0


def _disney_plain(c):
    return c["ledger"]["SHADE_X"] if (c["job"][1] == "disney" and c["textured"] == 0) else 0


def _debug(c):
    return c["ledger"]["SHADE_X"] if c["job"][1] == "debug" else 0


def _node_any(c):
    L = c["ledger"]
    return L["NFULL_X"] + L["NGLOB_X"] + L["NEXACT_X"] + sum(L[f"NFEW{i}_X"] for i in range(5))


BLOCKS = [
    ("SKY_X", "sky loop: a batch of up to 64 camera paths of a tile whose pyramid misses the root (primary ray, fast flag, environment, accumulate)", "SKY_X", "SKY_L", 624, 0.85, 1.15),
    ("S_ITER", "S block: entry, park decision, counters, exit into the traversal loop", "S_ITER", None, None, 20, 400),
    ("ENV_X", "S: environment lookup of the misses (atan2, asin, bilinear fetch, 3 x pow)", "ENV_X", "ENV_L", 369, 0.9, 1.1),
    ("SHADE_T", "S: shade_hit, Disney material with 4 textures (helmet)", _disney_tex, "SHADE_L", 1747, 0.75, 1.05),
    ("SHADE_P", "S: shade_hit, Disney material without textures", _disney_plain, "SHADE_L", 1200, 0.6, 1.2),
    ("SHADE_D", "S: shade_hit, debug material (normal -> colour, path ends)", _debug, "SHADE_L", 200, 0.3, 3.0),
    ("PSTORE_X", "S: park hits (18 dwords of path state per hit to memory)", "PSTORE_X", "PSTORE_L", 51, 0.5, 2.0),
    ("PLOAD_X", "S: parked hits back into idle lanes", "PLOAD_X", "PLOAD_L", 58, 0.5, 2.0),
    ("ACCUM_X", "S: finished samples into the LDS tile (quantise, 3 x ds_add_u64)", "ACCUM_X", "ACCUM_L", 42, 0.7, 1.5),
    ("REGEN_X", "S: regeneration loop iteration (unit bookkeeping, lane -> path)", "REGEN_X", None, 90, 0.7, 1.3),
    ("PRIM_X", "S: primary ray (seed, hash12 jitter, camera matrix, normalise)", "PRIM_X", "PRIM_L", 102, 0.8, 1.2),
    ("START_X", "S: ray set-up (3 reciprocals, fast flag, traversal state)", "START_X", "START_L", 54, 0.8, 1.3),
    ("ROUND_X", "traversal round: block choice (2 ballots, popcounts, exit test)", "ROUND_X", None, None, 4, 80),
    ("NODE_ANY", "NODE (any form): perm word to LDS, level / node update, first child", _node_any, None, None, 10, 120),
    ("PYRCHK_X", "NODE: are the block's camera rays about to enter ONE node?", "PYRCHK_X", None, None, 0, 80),
    ("CULLMASK_X", "NODE: cull mask of a (tile, node) on a cache miss (32 lanes x box vs plane)", "CULLMASK_X", None, 64, 0.8, 1.5),
    ("NFULL_X", "NODE full: 8 slab tests from the LDS tree (6 fma per child) + rank sort", "NFULL_X", "NFULL_L", 218, 0.9, 1.15),
    ("NGLOB_X", "NODE full, node through L1 / L2", "NGLOB_X", "NGLOB_L", 274, 0.8, 1.3),
    ("NFEW1_X", "NODE culled, 1 surviving child", "NFEW1_X", "NFEW1_L", 25, 0.6, 2.0),
    ("NFEW2_X", "NODE culled, 2 surviving children", "NFEW2_X", "NFEW2_L", 54, 0.6, 1.6),
    ("NFEW3_X", "NODE culled, 3 surviving children", "NFEW3_X", "NFEW3_L", 120, 0.6, 1.4),
    ("NFEW4_X", "NODE culled, 4 surviving children", "NFEW4_X", "NFEW4_L", 138, 0.6, 1.4),
    ("LEAF_X", "LEAF: 8 Moeller-Trumbore tests (18 x dwordx4, 8 short reciprocals; per triangle, the third dot product and the update -- 11 of 49 -- behind a branch the wave skips when no lane passed u and v)", "LEAF_X", "LEAF_L", 391, 0.70, 1.03),
    ("POP_X", "pop iteration: next child of the current node", "POP_X", "POP_L", 27, 0.4, 2.5),
    ("POP_UP_X", "pop iteration: some lane goes up to the nearest live level (perm word from LDS)", "POP_UP_X", "POP_UP_L", 19, 0.5, 2.0),
    ("POP_RETEST_X", "pop iteration: some lane re-tests its child against a closer hit (3 near planes)", "POP_RETEST_X", "POP_RETEST_L", 25, 0.5, 2.5),
    ("TILE_X", "tile set-up (pyramid planes, root cull) + flush of the LDS tile", "TILE_X", None, 700, 0.5, 2.0),
    ("JOIN_X", "join scan for an open tile", "JOIN_X", None, None, 0, 2000),
    ("GRAB_X", "unit grab (atomic on the tile's counter, unit -> pixels)", "GRAB_X", None, None, 0, 200),
]


def executions(c, spec):
    return spec(c) if callable(spec) else c["ledger"][spec]


def load(tag):
    d = os.path.join(ROOT, "gpurun_out", tag)
    if not os.path.exists(os.path.join(d, "counts.jsonl")):
        d = os.path.join(ROOT, "profiles", tag)           # the committed copy of a run (profiles/r04_ledger_raw)
    counts = [json.loads(l) for l in open(os.path.join(d, "counts.jsonl")) if l.startswith("{")]
    prod = [json.loads(l) for l in open(os.path.join(d, "product.jsonl")) if l.startswith("{")]
    per = collections.OrderedDict()
    for line in open(os.path.join(d, "pmc.csv")).read().splitlines()[1:]:
        did, name, val = line.split(",")
        per.setdefault(int(did), {})[name] = float(val)
    disp = [per[k] for k in sorted(per)]
    assert len(disp) == 2 * len(counts) == 2 * len(prod), (len(disp), len(counts), len(prod))
    pmc = disp[1::2]                      # the second launch of every job (tile order from the first one's costs)
    for c, p in zip(counts, prod):
        assert c["job"] == p["job"] and c["rays"] == p["rays"], (c["job"], p["job"])
    return counts, prod, pmc, d


def main(tag, out):
    from scipy.optimize import lsq_linear
    counts, prod, pmc, src_dir = load(tag)
    X = np.array([[executions(c, b[2]) for b in BLOCKS] for c in counts], float)
    y = np.array([p["SQ_INSTS_VALU"] for p in pmc], float)
    lo = np.array([(b[5] if b[4] is None else b[4] * b[5]) for b in BLOCKS], float)
    hi = np.array([(b[6] if b[4] is None else b[4] * b[6]) for b in BLOCKS], float)
    scale = 1.0 / y                                        # relative residuals: a small launch counts as much as the big frame
    res = lsq_linear(X * scale[:, None], y * scale, bounds=(lo, hi))
    coef = res.x
    prior = np.array([np.nan if b[4] is None else b[4] for b in BLOCKS])
    pred = X @ coef
    rel = pred / y - 1.0
    # the static counts alone (glue terms at their fitted value): how far do the ISA counts get without fitting?
    coef_static = np.where(np.isnan(prior), coef, prior)
    rel_static = (X @ coef_static) / y - 1.0
    head, L, total = counts[0], counts[0]["ledger"], y[0]
    lines = ["# Block ledger of `rt_path_kernel_stream<16, true, 1, true>` -- config #3 (helmet 1920x1080, 256 spp, 8 bounces), 1 x MI355X", "",
             "Where the VALU wave-instructions of the product kernel go, block by block (VERDICT r03 #2).", "",
             "* **Executions and lanes**: counted by a `-DRT_LEDGER=1` build of the same kernel (`tools/exp_ledger.py`, `tools/gpu/ledger.sh`): one row",
             "  of counters per wave in memory, bumped by lane 0 with atomics that return nothing.  (As scalar registers the counters did not fit",
             "  beside the kernel's own: 50-290 spilled VGPRs -- a different kernel.)  Ray / node / leaf / shade counters of that build equal the product's.",
             "* **Static instructions**: VALU instructions of the block in the ISA of a `-DRT_LEDGER_MARKS` build -- the product kernel plus comment",
             "  markers at the block boundaries, same register allocation (22 spilled VGPRs, 72 B scratch) -- `tools/ledger_static.py`.  A block costs the",
             "  same number of wave-instructions whatever the number of lanes in it; both sides of a divergent branch inside a block are issued.",
             f"* **Fitted instructions**: `SQ_INSTS_VALU` of the PRODUCT library, one `rocprofv3 --pmc` value per launch, over {len(counts)} launches with different",
             "  block mixes (4 scenes, 2 shaders, 1-24 bounces, 8-512 spp, 256x256 ... 3840x2160) = sum over blocks of cost x executions; bounded least",
             "  squares on relative residuals, every cost bounded to a band around its static count, glue code without a clean region",
             "  (\"-\" in the static column) free (`tools/ledger_fit.py`).", "",
             f"Fit: predicted / measured - 1 over the {len(counts)} launches: rms {np.sqrt((rel ** 2).mean()):.4f}, worst {rel[np.argmax(np.abs(rel))]:+.4f}; "
             f"config #3: {pred[0] / 1e9:.3f} G predicted vs {y[0] / 1e9:.3f} G measured ({rel[0]:+.4f}).  With the static counts taken as they are "
             f"(only the glue terms fitted): rms {np.sqrt((rel_static ** 2).mean()):.4f}, config #3 {rel_static[0]:+.4f}.", "",
             "| block | executions | mean lanes | static instr | fitted instr | wave-instructions | share | idle-lane loss (share of all lane-slots) |",
             "|---|---|---|---|---|---|---|---|"]
    rows = []
    for b, c in zip(BLOCKS, coef):
        n = executions(head, b[2])
        lanes = None
        if b[3] and n:
            lanes = L[b[3]] / n
        rows.append((b[1], n, lanes, b[4], c, c * n))
    tot_insts = sum(r[5] for r in rows)
    for label, n, lanes, st, c, insts in rows:
        if n == 0:
            continue
        idle = insts * (1.0 - lanes / 64.0) / tot_insts if lanes is not None else None
        lines.append(f"| {label} | {n / 1e6:.2f} M | {('%.1f' % lanes) if lanes is not None else 'wave-level'} | {st if st is not None else '-'} | {c:.0f} | "
                     f"{insts / 1e9:.3f} G | {insts / total:.3f} | {('%.3f' % idle) if idle is not None else '-'} |")
    lines += [f"| **sum of the rows** | | | | | **{tot_insts / 1e9:.3f} G** | {tot_insts / total:.3f} | |",
              f"| `SQ_INSTS_VALU` of the launch (PMC) | | | | | **{total / 1e9:.3f} G** | 1.000 | |", ""]
    lane_insts = sum(r[5] * ((r[2] if r[2] is not None else 64.0) / 64.0) for r in rows)
    meas = pmc[0]["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc[0]["SQ_ACTIVE_INST_VALU"])
    lines += [f"Mean active lanes per VALU instruction implied by the rows (wave-level code = 64 lanes; a pop iteration's lanes = the lanes popping): "
              f"{lane_insts / tot_insts:.3f}; PMC (`SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)`): {meas:.3f}.", ""]
    # groups
    grp = collections.OrderedDict()
    for (key, *_), r in zip(BLOCKS, rows):
        g = ("sky loop (tiles that miss the root)" if key == "SKY_X" else "S block: environment" if key == "ENV_X" else "S block: shading (+ parking)" if key in ("SHADE_T", "SHADE_P", "SHADE_D", "PSTORE_X", "PLOAD_X") else
             "S block: regeneration, primary rays, ray set-up, accumulate, glue" if key in ("S_ITER", "ACCUM_X", "REGEN_X", "PRIM_X", "START_X", "GRAB_X") else
             "NODE blocks (full)" if key in ("NFULL_X", "NGLOB_X") else "NODE blocks (culled) + cull masks + candidate check" if key.startswith(("NFEW", "CULL", "PYR")) else
             "NODE glue" if key == "NODE_ANY" else "LEAF blocks" if key == "LEAF_X" else "pop loops" if key.startswith("POP") else
             "block choice" if key == "ROUND_X" else "tiles, joins")
        grp[g] = grp.get(g, 0.0) + r[5]
    lines += ["## By kind of block", "", "| kind | wave-instructions | share of SQ_INSTS_VALU |", "|---|---|---|"]
    for g, v in grp.items():
        lines.append(f"| {g} | {v / 1e9:.3f} G | {v / total:.3f} |")
    lines.append("")
    cam = dict(node_full=L["NFULL_CAM"] / max(1, L["NFULL_L"]), leaf=L["LEAF_CAM"] / max(1, L["LEAF_L"]), pop=L["POP_CAM"] / max(1, L["POP_L"]))
    nfew_l = sum(L[f"NFEW{i}_L"] for i in range(5))
    lines += ["## Camera rays vs bounce rays", "",
              f"Camera rays are {head['paths'] / head['rays']:.3f} of the rays ({head['paths'] / 1e6:.1f} M of {head['rays'] / 1e6:.1f} M).  Share of the LANES of a block that are camera rays: "
              f"full node blocks {cam['node_full']:.3f}, culled node blocks 1.000 (by construction; {nfew_l / 1e6:.0f} M lane-visits against "
              f"{L['NFULL_L'] / 1e6:.0f} M in full blocks), leaf blocks {cam['leaf']:.3f}, pop iterations {cam['pop']:.3f}.  "
              f"Lanes that WAIT while a culled block runs for the camera rays alone: {L['NODE_WAIT_L'] / 1e6:.1f} M lane-blocks.", ""]
    cyc_file = os.path.join(src_dir, "cycles.jsonl")
    if os.path.exists(cyc_file):
        cy = json.loads(open(cyc_file).readline())
        CL, tot = cy["ledger"], cy["ledger"]["CYC_WAVE"]
        named = [("S blocks (environment, shading, regeneration, ray set-up)", "CYC_S"), ("NODE blocks", "CYC_NODE"), ("LEAF blocks", "CYC_LEAF"),
                 ("pop loops", "CYC_POP"), ("tile set-up, joins", "CYC_TILE")]
        if CL.get("CYC_SKY"):
            named.insert(0, ("sky loop (tiles that miss the root)", "CYC_SKY"))
        lines += ["## Shader-clock cycles per kind of block (`-DRT_LEDGER=2`: `s_memtime` around the blocks, summed over the 4 096 waves)", "",
                  f"(That build runs the frame in {cy['kernel_ms']:.1f} ms -- the timers and counters perturb it; the SHARES are what it is for.)", "",
                  "| kind | share of the waves' cycles | share of the VALU wave-instructions (table above) |", "|---|---|---|"]
        inst_share = {"CYC_S": sum(v for g, v in grp.items() if g.startswith("S block")) / total,
                      "CYC_SKY": grp.get("sky loop (tiles that miss the root)", 0.0) / total,
                      "CYC_NODE": sum(v for g, v in grp.items() if g.startswith("NODE")) / total,
                      "CYC_LEAF": grp.get("LEAF blocks", 0.0) / total, "CYC_POP": grp.get("pop loops", 0.0) / total,
                      "CYC_TILE": grp.get("tiles, joins", 0.0) / total}
        for label, k in named:
            lines.append(f"| {label} | {CL[k] / tot:.3f} | {inst_share[k]:.3f} |")
        rest = 1.0 - sum(CL[k] for _, k in named) / tot
        lines += [f"| block choice, loop glue, waiting at the kernel's start and end | {rest:.3f} | {grp.get('block choice', 0.0) / total:.3f} |", ""]
    ta_file = os.path.join(src_dir, "ta_pmc.txt") if os.path.exists(os.path.join(src_dir, "ta_pmc.txt")) else os.path.join(src_dir, "ta", "pmc.txt")
    if os.path.exists(ta_file):
        ta = {}
        for l in open(ta_file):
            f = l.split()
            if len(f) >= 3 and f[0].startswith("rt_path_kernel"):
                nl = int(l.split("launches=")[1])
                ta[f[-3]] = float(f[-2]) / nl
        if "TA_TA_BUSY_sum" in ta:
            cyc = ta["GRBM_GUI_ACTIVE"] / 8.0
            lines += ["## TA / TCP side of the memory pipe (VERDICT r03 #8; per launch of config #3, counters two per pass)", "",
                      "| counter | per launch | per CU and cycle of the launch |", "|---|---|---|"]
            for k in ("TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum",
                      "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TA_FLAT_READ_WAVEFRONTS_sum"):
                if k in ta:
                    lines.append(f"| `{k}` | {ta[k]:.4g} | {ta[k] / 256.0 / cyc:.3f} |")
            lines += ["", f"(`GRBM_GUI_ACTIVE` / 8 XCDs = {cyc / 1e6:.1f} M cycles; the `_sum` counters add the 256 CUs.)  The texture-address unit is busy "
                      f"{ta['TA_TA_BUSY_sum'] / 256 / cyc:.0%} of the time and stalled by the cache {ta['TA_ADDR_STALLED_BY_TC_CYCLES_sum'] / 256 / cyc:.1%} (address) / "
                      f"{ta['TA_DATA_STALLED_BY_TC_CYCLES_sum'] / 256 / cyc:.1%} (data) of it; the L1 takes {ta['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256 / cyc:.2f} accesses per cycle "
                      f"and sends {ta['TCP_TCC_READ_REQ_sum'] / ta['TCP_TOTAL_CACHE_ACCESSES_sum']:.1%} of them on to the L2.  The memory pipe is not saturated: what a "
                      "leaf or texel fetch costs is its latency (r03's reading, now with the counters the aborted pass did not deliver).", ""]
    open(os.path.join(ROOT, "profiles", f"{out}_blocks.md"), "w").write("\n".join(lines) + "\n")
    json.dump(dict(blocks=[dict(block=b[0], executions=int(executions(head, b[2])), static=b[4], fitted=float(c)) for b, c in zip(BLOCKS, coef)],
                   fit=dict(rms=float(np.sqrt((rel ** 2).mean())), worst=float(rel[np.argmax(np.abs(rel))]), config3=float(rel[0]),
                            static_rms=float(np.sqrt((rel_static ** 2).mean())), static_config3=float(rel_static[0])),
                   ledger_config3=L), open(os.path.join(ROOT, "profiles", f"{out}_blocks.json"), "w"), indent=1)
    print("\n".join(lines))
    for c, r, rs in zip(counts, rel, rel_static):
        print(c["job"], f"{r:+.4f} static {rs:+.4f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

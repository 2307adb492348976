#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>_* (rocprofv3 csv output of tools/profile_gpu.sh) into
profiles/<tag>_summary.md + profiles/<tag>_traffic.json (committed, cited by DESIGN.md / bench.py)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    lines = [f"# rocprofv3 summary `{tag}` -- bench.py {os.environ.get('BENCH_ARGS', '(helmet 1920x1080, 256 spp, 8 bounces)')}, 1 x MI355X", ""]
    stats = sorted(glob.glob(os.path.join(src, f"prof_{tag}_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]
    kernel_avg_ms = None
    if stats:
        lines += [f"## `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps {os.environ.get('STEPS', '9')} --warmup 1 {os.environ.get('BENCH_ARGS', '')}`", "",
                  "| kernel | calls | total ms | average ms | % | min ms | max ms |", "|---|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(stats[0])):
            name = r["Name"].split("(")[0][:60]
            lines.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | "
                         f"{float(r['Percentage']):.3f} | {float(r['MinNs'])/1e6:.4f} | {float(r['MaxNs'])/1e6:.4f} |")
            if "rt_path_kernel" in name:
                kernel_avg_ms = float(r["AverageNs"]) / 1e6
        lines.append("")
    bench_json = os.path.join(src, f"prof_{tag}_stats.json")
    bench = None
    if os.path.exists(bench_json):
        txt = [l for l in open(bench_json).read().splitlines() if l.startswith("{")]
        if txt:
            bench = json.loads(txt[-1])
            lines += ["bench.py line of the same (profiled) run: "
                      f"value {bench['value']:.1f} Mray/s, ms_per_step {bench['ms_per_step']:.2f}, "
                      f"kernel_ms (HIP events, mean per launch) {bench['kernel_ms']:.3f}"
                      + (f" vs rocprof average {kernel_avg_ms:.3f}" if kernel_avg_ms else ""), ""]
    counters = collections.OrderedDict()
    meta = {}
    newest = {}
    for f in glob.glob(os.path.join(src, f"prof_{tag}_pmc_*", "*", "*_counter_collection.csv")):
        key = f.split(os.sep)[-3]            # one pass directory may hold the files of several gpurun calls: newest wins
        if key not in newest or os.path.getmtime(f) > os.path.getmtime(newest[key]):
            newest[key] = f
    for f in sorted(newest.values()):
        for r in csv.DictReader(open(f)):
            if "rt_path_kernel" in r["Kernel_Name"]:
                counters[r["Counter_Name"]] = counters.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
    if counters:
        lines += ["## PMC counters of ONE `rt_path_kernel` launch (separate `rocprofv3 --pmc ...` passes, "
                  "`bench.py --steps 1 --warmup 0`)", "", f"dispatch: {meta}", "", "| counter | value |", "|---|---|"]
        for k, v in counters.items():
            lines.append(f"| {k} | {v:.6g} |")
        lines.append("")
    traffic = None
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        fetch_kb, write_kb = counters["FETCH_SIZE"], counters["WRITE_SIZE"]
        # MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads half
        # the bytes of 16-B-per-lane reads -> doubled; WRITE_SIZE is exact.
        hbm = (2.0 * fetch_kb + write_kb) * 1024.0
        sys.path.insert(0, ROOT)
        from raytracing_c_amd.buildinfo import hipflags, kernel_source_hash
        traffic = {"tag": tag, "kernel_hash": kernel_source_hash(), "hipflags": hipflags(), "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "hbm_bytes_per_launch": hbm,
                   "correction": "2 x FETCH_SIZE (gfx950 half-count of 16 B/lane reads) + WRITE_SIZE, x 1024",
                   "workload": bench["config"]["workload"] if bench else None,
                   "rays_per_launch": bench["rays_per_frame"] if bench else None}
        lines += [f"HBM-side traffic per launch = (2 x {fetch_kb:.0f} + {write_kb:.0f}) KB = **{hbm/1e9:.2f} GB** "
                  "(FETCH_SIZE doubled per MI355X_MICROARCH.md; it counts L2 fabric requests, Infinity-Cache hits included).", ""]
        if bench:
            b_ray = bench["roofline"]["algorithmic"]["bytes_per_ray"]
            alg = bench["rays_per_frame"] * b_ray
            lines += [f"Algorithmic scene bytes per launch = {bench['rays_per_frame']} rays x {b_ray:.1f} B "
                      f"= {alg/1e9:.1f} GB -> traffic/algorithmic = {hbm/alg:.4f}: the ~60 MB scene is served by L1/L2/Infinity Cache, "
                      "the kernel is not HBM bound.", ""]
        if "TCC_HIT_sum" in counters:
            hit, miss = counters["TCC_HIT_sum"], counters["TCC_MISS_sum"]
            lines += [f"L2 hit rate = {hit/(hit+miss):.3f}", ""]
    if "SQ_WAVE_CYCLES" in counters:
        wc = counters["SQ_WAVE_CYCLES"]
        lines += ["Wave-cycle split: "
                  f"ACTIVE_INST_ANY {counters.get('SQ_ACTIVE_INST_ANY', 0)/wc:.2f}, WAIT_ANY {counters.get('SQ_WAIT_ANY', 0)/wc:.2f}, "
                  f"WAIT_INST_ANY {counters.get('SQ_WAIT_INST_ANY', 0)/wc:.2f}; VALU share of wave cycles "
                  f"{counters.get('SQ_ACTIVE_INST_VALU', 0)/wc:.2f}", ""]
    if "SQ_THREAD_CYCLES_VALU" in counters and "SQ_ACTIVE_INST_VALU" in counters:
        lines += [f"Mean active lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) "
                  f"= {counters['SQ_THREAD_CYCLES_VALU']/(64*counters['SQ_ACTIVE_INST_VALU']):.2f}", ""]
    if traffic and "SQ_INSTS_VALU" in counters and "GRBM_GUI_ACTIVE" in counters:
        # VALU issue: a wave64 fp32 instruction occupies its SIMD for 2 cycles (128 fp32 lanes per CU = 4 SIMDs x 32);
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
        simd_cycles = counters["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        traffic["gui_active_cycles_per_xcd"] = counters["GRBM_GUI_ACTIVE"] / 8.0
        traffic["valu"] = {"wave_insts": counters["SQ_INSTS_VALU"],
                           "issue_frac": 2.0 * counters["SQ_INSTS_VALU"] / simd_cycles,
                           "active_lane_frac": counters["SQ_THREAD_CYCLES_VALU"] / (64 * counters["SQ_ACTIVE_INST_VALU"])
                           if "SQ_THREAD_CYCLES_VALU" in counters and "SQ_ACTIVE_INST_VALU" in counters else None,
                           "waves_per_simd": 4}
        lines += [f"VALU issue = 2 cycles x SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = "
                  f"**{traffic['valu']['issue_frac']:.2f}** of the SIMD cycles of the launch", ""]
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
    if traffic:
        json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    for f in stats:
        import shutil
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")

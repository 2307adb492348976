#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on N MI355X of one node.

A "step" is one full frame of the hot path (per-pixel sample loop -> BVH traversal ->
ray/triangle tests -> BSDF -> bounce loop) on BASELINE.json's metric configuration:
helmet, 1920x1080, 256 spp, 8 bounces (configs[2]; scene = assets/helmet.glb, the
self-contained form of models/helmet.gltf).  Scene, textures and background are resident in
HBM before the timed region; the region covers accumulation-buffer clear, the path kernel,
resolve to u8, the framebuffer tile gather over RCCL (N > 1), untile and the D2H copy of the
finished image on rank 0 (on a copy stream, double-buffered: it overlaps the next frame's kernel;
the final synchronisation waits for the last copy) -- the reference's own timed region is thread
spawn -> finish (driver.c:791-821).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: starts the N ranks itself, see launch_ranks())

Rank 0 prints ONE JSON line.  value = rays traced by all ranks per second / 1e6 (Mray/s,
rays counted in-kernel), weak/strong: the frame is fixed, so scaling is "strong".
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="helmet", choices=["spheres", "quad", "helmet", "tower", "helmet4k"],
                    help="BASELINE.json configs[0..4]; the default is the configuration the metric is quoted on")
    ap.add_argument("--pipeline", default="stream", choices=["stream", "wavefront"],
                    help="stream = the product's path kernel; wavefront = the split camera / shade / trace pipeline (measurement only)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=0)
    ap.add_argument("--slab", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    ap.add_argument("--save", default="", help="write the last frame as PNG (rank 0)")
    ap.add_argument("--no-bvh-compare", action="store_true",
                    help="skip the side measurement of the opt-in SAH builder (config.bvh.sah)")
    ap.add_argument("--dry-run-sleep", type=float, default=0.0, help=argparse.SUPPRESS)   # tests: ranks linger this long
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: ranks rendezvous (gloo), exchange an empty tile buffer, rank 0 prints a "
                         "stub line -- covers the launcher and the N-rank plumbing on a CPU-only machine")
    return ap.parse_args()


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes through
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line.
    This parent has not imported torch and never touches the GPU; it is not replaced by an exec, it waits
    for the children and exits with their status.  The children run in a process group of their own: when the
    parent is told to stop (SIGTERM from a `timeout`, the driver's time limit, Ctrl-C) or fails, the whole group --
    torchrun and every rank, including one that hangs on the GPU -- is terminated, SIGKILLed after a grace period."""
    import signal
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, start_new_session=True)
    pgid = proc.pid                                  # start_new_session: the child leads a new session and group

    def group_alive():
        try:
            os.killpg(pgid, 0)
            return True
        except (ProcessLookupError, PermissionError):
            return False

    def kill_group(grace=float(os.environ.get("RT_BENCH_KILL_GRACE", "5"))):
        for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 2.0)):
            try:
                os.killpg(pgid, sig)
            except (ProcessLookupError, PermissionError):
                return
            t0 = time.time()
            while time.time() - t0 < wait:
                proc.poll()                          # reap torchrun so that the group can disappear
                if not group_alive():
                    return
                time.sleep(0.05)

    def on_signal(signum, _frame):
        raise KeyboardInterrupt(f"signal {signum}")

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    line, rc = None, 1
    try:
        for raw in proc.stdout:
            txt = raw.decode("utf-8", "replace").rstrip("\n")
            if txt.startswith("{") and line is None:
                line = txt
            else:
                print(txt, file=sys.stderr)
        rc = proc.wait()
    except KeyboardInterrupt as e:
        print(f"bench.py: stopping the ranks ({e})", file=sys.stderr)
        line, rc = None, 130
    finally:
        if proc.poll() is None or group_alive():
            kill_group()
        for sig, h in old.items():
            signal.signal(sig, h)
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 3
    sys.exit(rc)


def cpu_baseline(hs, cfg, target_seconds):
    """The CPU checker (oracle/oracle.c) as the measured CPU baseline, on a bounded sample of the same workload: the full
    frame at (possibly) reduced spp.  kind "port-avx2": the reference's 8-wide AVX2 forms of ray_aabbs_hit_8 /
    ray_triangles_hit_8 / min_f32x8 (raytracer.c:15-32,84-230) restated in the checker and proven bit-identical to its
    scalar form (tests/test_oracle_simd.py) -- the reference's SIMD CPU path, which north_star asks to be timed on the same
    box's host cores; the scalar form's figures stand beside it (`scalar`).  T = all host threads and T = 1.
    Test infrastructure used as the measured baseline only -- never on the product path."""
    import subprocess
    import tempfile
    from tests import _oracle

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a container may see every host CPU but be limited by a cgroup quota (a 1-GPU box gets a 16-CPU share)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = max(1, min(cores, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = max(1, min(cores, int(round(q / p_))))
            break
        except Exception:
            continue
    if "RT_CPU_THREADS" in os.environ:
        cores = int(os.environ["RT_CPU_THREADS"])
    lib = None
    # rebuild the checker for this host's ISA (-march=native) when a compiler is present
    try:
        tmp = tempfile.mkdtemp(prefix="oracle_native_")
        out = os.path.join(tmp, "liboracle_native.so")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "MARCH=native", f"OUT={out}", out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = _oracle.load(out)
    except Exception:
        lib = _oracle.load()
    w, h, b = cfg["width"], cfg["height"], cfg["max_bounces"]
    simd = bool(lib.oracle_have_avx2())

    def timed(spp, threads):
        t0 = time.perf_counter()
        r = _oracle.render(hs, w, h, spp, b, n_threads=threads, lib=lib)
        return r["counters"]["rays"], time.perf_counter() - t0

    def measure(mode, seconds):
        lib.oracle_set_simd(mode)
        _, t1 = timed(1, cores)
        spp = int(max(1, min(cfg["samples"], round(seconds / max(t1, 1e-3)))))
        rays, dt = timed(spp, cores)
        rays1, dt1 = timed(2, 1)           # per-core figure (T = all threads and T = 1)
        return dict(mray_per_s=rays / dt / 1e6, single_thread_mray_per_s=rays1 / dt1 / 1e6, spp=spp, seconds=dt,
                    msample_per_s=w * h * spp / dt / 1e6)

    try:
        main = measure(1 if simd else 0, target_seconds)
        scalar = measure(0, target_seconds / 3.0) if simd else main
    finally:
        lib.oracle_set_simd(1)
    form = "8-wide AVX2 forms of the reference's SIMD routines" if simd else "scalar restatement (no AVX2 on this host)"
    return {"value": main["mray_per_s"], "unit": "Mray/s", "cores": cores, "kind": "port-avx2" if simd else "port",
            "single_thread_mray_per_s": main["single_thread_mray_per_s"],
            "sample": f"{cfg['asset']} {w}x{h}, {main['spp']} of {cfg['samples']} spp, {b} bounces, {main['seconds']:.1f} s, "
                      f"oracle -O3 -march=native ({form}), {cores} threads",
            "msample_per_s": main["msample_per_s"],
            "scalar": {"kind": "port", "value": scalar["mray_per_s"], "single_thread_mray_per_s": scalar["single_thread_mray_per_s"],
                       "sample": f"{scalar['spp']} spp, {scalar['seconds']:.1f} s"}}


def measured_profile(workload):
    """PMC summary of this workload (profiles/*_traffic.json, newest tag wins): HBM-side bytes, VALU
    wave-instructions, active lanes, SIMD cycles of ONE rt_path_kernel launch, collected by separate
    `rocprofv3 --pmc` passes of this same command (tools/profile_gpu.sh + tools/summarize_profile.py).
    bench.py cannot read PMC counters itself: these values are REPLAYED and labelled as such -- and only while the
    profile's `kernel_hash` (sources of the path kernel + compiler flags, raytracing_c_amd/buildinfo.py) equals the
    tree's: returns (profile, file name, stale)."""
    import glob
    from raytracing_c_amd.buildinfo import kernel_source_hash
    best = None
    # newest = last in name order (r01 < r01f < ... < r02): file times mean nothing after a fresh clone
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("workload") == workload:
            best = (t, os.path.basename(f))
    if best is None:
        return None
    return best[0], best[1], best[0].get("kernel_hash") != kernel_source_hash()


# MI355X: 256 CUs x 4 SIMD-32; a wave64 VALU instruction holds its SIMD for 2 cycles (MI355X_MICROARCH.md,
# Wave scheduling); 2.4 GHz is the chip's maximum clock -- the clock under load is lower, so `frac` against this
# peak is a lower bound of the issue-slot use at the real clock (`valu.issue_frac_profiled_clock`).
SIMDS = 256 * 4
MAX_CLOCK_HZ = 2.4e9
VALU_PEAK_GINST = SIMDS * MAX_CLOCK_HZ / 2.0 / 1e9      # G wave-instructions / s
LDS_PEAK_GBS = 256 * 128 * MAX_CLOCK_HZ / 1e9           # 128 B / clk / CU


FP32_VECTOR_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 32 lanes x 2 flop (FMA) x 2.4 GHz


def roofline_block(tot, rays_per_launch, kernel_ms, n_launches, prof, variant_name, kernel_ms_source, profile_scale=1.0,
                   skipped_root_visits=None):
    """SURVEY 8d prices a ray at 200 flop per node visit + 480 per leaf visit.  The scene is cache resident (the 8d HBM byte
    model gives a rate ABOVE the HBM peak: listed under `algorithmic`, labelled), and there is no dense contraction for
    MFMA, so the roof this path can be priced against is the fp32 VECTOR peak -- `frac` = algorithmic flops / launch time /
    157.3 TFLOP/s, computed LIVE from the in-kernel counters and the HIP-event launch time.  `decomposition` factors it:
    VALU issue x active lanes x algorithmic flops per executed lane-instruction x 1/2 (no FMA contraction: one flop per
    instruction against a 2-flop peak; bit parity with the CPU oracle forbids contraction).  Issue and lanes need PMC
    counters: replayed from the newest committed profile of this workload, and only while its kernel hash matches the tree."""
    b_ray = tot.bytes_per_ray()
    ksec = kernel_ms * 1e-3 if kernel_ms and kernel_ms > 0 else None
    alg_bytes = rays_per_launch * b_ray
    rays = max(tot.rays, 1)
    node_bytes = 192.0 * tot.node_visits / rays * rays_per_launch
    flops_per_ray = (200.0 * tot.node_visits + 480.0 * tot.leaf_visits) / rays
    flops = flops_per_ray * rays_per_launch
    tflops = flops / ksec / 1e12 if ksec else None
    out = {"bound": "fp32_vector", "achieved": tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": tflops / FP32_VECTOR_PEAK_TFLOPS if tflops else None,
           "traffic": None, "kernel": variant_name, "launches_averaged": int(n_launches), "kernel_ms": kernel_ms,
           "kernel_ms_source": kernel_ms_source,
           "flops": {"per_ray": flops_per_ray, "per_launch": flops,
                     "model": "SURVEY 8d: 200 x node visits + 480 x leaf visits, counters from the kernel (equal to the oracle's). "
                              "FOOTNOTE: `culled_root_visits` of the node visits are COUNTED BUT NOT EXECUTED -- the one root visit of "
                              "every camera path whose 8x8 tile's pixel pyramid misses every child of the root; the reference spends "
                              "it (raytracer.c:459-472), the kernel proves its outcome per tile and skips it.  A flops-model fraction "
                              "is flattered by that share (`culled_flops_share`); `frac_executed_only` removes it"},
           "peak_note": "fp32 vector peak; MFMA does not apply (branchy fp32, no contraction), the HBM byte model of SURVEY 8d "
                        "is served on chip (see algorithmic.ratio_to_hbm_peak)",
           "algorithmic": {"bytes_per_ray": b_ray, "bytes_per_launch": alg_bytes,
                           "rate_GBps": alg_bytes / ksec / 1e9 if ksec else None,
                           "ratio_to_hbm_peak": alg_bytes / ksec / 1e9 / HBM_PEAK_GBS if ksec else None,
                           "note": "SURVEY 8d scene bytes (192 N + 288 L + 112 H + 48 X + 12 M per ray, counters from "
                                   "the kernel); served by LDS / L1 / L2 / Infinity Cache, NOT HBM traffic -- a ratio "
                                   "above 1 only says the scene is cache resident"},
           "lds": {"node_bytes_per_launch": node_bytes,
                   "rate_GBps": node_bytes / ksec / 1e9 if ksec else None, "peak_GBps": LDS_PEAK_GBS,
                   "frac": node_bytes / ksec / 1e9 / LDS_PEAK_GBS if ksec else None,
                   "note": "BVH node reads come from the workgroup's LDS copy of the tree; 192 B per node visit is the "
                           "algorithmic figure and an upper bound: pyramid-culled node blocks read 24 B per surviving child"},
           "l1_l2": {"bytes_per_launch": alg_bytes - node_bytes,
                     "rate_GBps": (alg_bytes - node_bytes) / ksec / 1e9 if ksec else None,
                     "note": "leaf tiles, shading records, texels through L1 / L2"}}
    if skipped_root_visits is not None:
        culled = float(skipped_root_visits) / max(tot.rays, 1) * rays_per_launch
        out["flops"]["culled_root_visits"] = culled
        out["flops"]["culled_flops_share"] = 200.0 * culled / flops if flops else None
        out["flops"]["culled_share_of_rays"] = culled / rays_per_launch if rays_per_launch else None
        out["flops"]["executed_node_visits"] = tot.node_visits / rays * rays_per_launch - culled
        out["frac_executed_only"] = (flops - 200.0 * culled) / ksec / 1e12 / FP32_VECTOR_PEAK_TFLOPS if ksec else None
    if prof and ksec:
        t, fname, stale = prof
        out["replayed_from"] = f"profiles/{fname}"
        out["replayed_stale"] = bool(stale)
        if stale:
            out["replayed_note"] = ("the kernel sources or compiler flags changed since this profile was taken (kernel_hash "
                                    "differs): its counters are NOT used; traffic, valu_issue and decomposition are null")
            out["valu_issue"] = None
            out["decomposition"] = None
            return out
        out["replayed_note"] = ("traffic, hbm.*, valu_issue.* and the issue / lanes factors of `decomposition` use PMC counters of "
                                "one launch of this same workload and kernel (separate rocprofv3 --pmc passes, committed "
                                "summary, kernel_hash equal to the tree's); launch time and flops are live")
        if profile_scale != 1.0:
            out["replayed_scaled"] = (f"counters of the single-GPU launch scaled by {profile_scale:.4f} = this rank's share of "
                                      "the frame's rays")
        hbm = t.get("hbm_bytes_per_launch")
        if hbm:
            hbm *= profile_scale
            out["traffic"] = hbm
            out["hbm"] = {"achieved": hbm / ksec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": hbm / ksec / 1e9 / HBM_PEAK_GBS, "traffic_over_algorithmic": hbm / alg_bytes,
                          "correction": t.get("correction")}
        v = t.get("valu") or {}
        if v.get("wave_insts"):
            wi = v["wave_insts"] * profile_scale
            ach = wi / ksec / 1e9
            lanes = v.get("active_lane_frac")
            issue = ach / VALU_PEAK_GINST
            out["valu_issue"] = {"achieved": ach, "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s", "frac": issue,
                                 "wave_insts_per_launch": wi, "issue_frac_profiled_clock": v.get("issue_frac"),
                                 "active_lane_frac": lanes, "waves_per_simd": v.get("waves_per_simd"),
                                 "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction; "
                                              "issue_frac_profiled_clock = the same at the clock of the profiled launch"}
            if lanes:
                per_lane_inst = flops / (wi * 64.0 * lanes)
                out["decomposition"] = {"valu_issue": issue, "active_lanes": lanes,
                                        "flops_per_executed_lane_instruction": per_lane_inst, "fma_factor": 0.5,
                                        "product": issue * lanes * per_lane_inst * 0.5,
                                        "valu_issue_ceiling_measured": [0.67, 0.70],
                                        "note": "product = frac (up to the ratio of the nominal 2.4 GHz to itself): every factor is a lever, and none alone the bound "
                                                "(DESIGN.md 4.1: 4 % fewer instructions returned 0.6 %). "
                                                "valu_issue_ceiling_measured = what a straight-line stream of independent v_fma_f32 (plain / mixed "
                                                "with mul and sub) reaches at 4 waves per SIMD on this chip in the same units "
                                                "(tools/exp/pk_issue_bench.hip, profiles/r04ad_pk_issue_bench.log: 2.98 / 2.85 cycles per "
                                                "wave-instruction)"}
    return out


def bvh_compare(rt, abi, args, cfg, image_ref, steps=3):
    """Side measurement for config.bvh: the same frame on a scene built by the opt-in scene_init_sah() (same layout,
    tighter boxes).  The headline stays on the reference's split (scene.c:311-414); GPU == oracle parity is untouched
    because both traverse whatever Scene they are given.  Returns node / leaf visits per ray, Mray/s and the number of
    pixels that differ from the reference-split image (only exact-distance ties may)."""
    import numpy as np
    import torch
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config(args.config, builder="sah")
    w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    if not d:
        return {"error": rt.last_error()}
    try:
        dev = torch.device("cuda", torch.cuda.current_device())
        accum = torch.zeros((h, w, 3), dtype=torch.int64, device=dev)
        image = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        params = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, args.slab, 0)
        stream = torch.cuda.current_stream().cuda_stream

        def frame():
            accum.zero_()
            if rt.lib.rt_render_accumulate(d, C.byref(params), accum.data_ptr(), stream) != 0:
                raise RuntimeError(rt.last_error())
            if rt.lib.rt_resolve(C.byref(params), accum.data_ptr(), None, image.data_ptr(), None, stream) != 0:
                raise RuntimeError(rt.last_error())

        frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            frame()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        cnt = rt.render.get_counters()
        differing = int((image.cpu().numpy() != image_ref).any(axis=2).sum()) if image_ref is not None else None
        return {"builder": "scene_init_sah (opt-in; same implicit 8-ary layout)", "mray_per_s": cnt.rays / dt / 1e6,
                "ms_per_step": dt * 1e3, "node_visits_per_ray": cnt.node_visits / max(cnt.rays, 1),
                "leaf_visits_per_ray": cnt.leaf_visits / max(cnt.rays, 1), "rays_per_frame": cnt.rays,
                "pixels_differing_from_reference_split": differing, "pixels": w * h,
                "build_ms": hs.scene_init_seconds * 1e3}
    finally:
        rt.lib.rt_scene_release(d)


def boundary_cost(rt, abi, hs, frames=8):
    """Side measurement for config.boundary: frames through the reference's own entry point render_thread_proc (the protocol of
    driver.c:793-818: Rendering_Context, a worker thread, polling) at the reference driver's DEFAULT frame -- 1024 x 1024, 16 spp,
    8 bounces (driver.c:733-742) -- on this scene, split into host and GPU phases by rt_get_frame_timing().  Medians over the
    frames after the first two (upload, no schedule feedback yet).  The same through a C host: profiles/r03_boundary.md."""
    w, h, s, b = 1024, 1024, 16, 8
    rows = []
    for _ in range(frames + 2):
        r = rt.render_context(hs, w, h, s, b, n_threads=1)
        if not r["finished"]:
            return {"error": rt.last_error()}
        t = abi.RT_Frame_Timing()
        if rt.lib.rt_get_frame_timing(C.byref(t)) != 0:
            return {"error": rt.last_error()}
        rows.append([getattr(t, f[0]) for f in t._fields_])
    rows = rows[2:]
    med = [sorted(col)[len(col) // 2] for col in zip(*rows)]
    names = [f[0] for f in abi.RT_Frame_Timing._fields_]
    out = {"frame": f"{w}x{h}, {s} spp, {b} bounces (driver.c:733-742), via render_thread_proc", "frames": frames}
    out.update({n: v for n, v in zip(names, med)})
    out["non_kernel_share"] = (out["total_ms"] - out["gpu_path_ms"]) / out["total_ms"] if out["total_ms"] > 0 else None
    # the same frames for a host that has a NEXT frame: rt_frame_begin / rt_frame_end keep two on the GPU (the second one's
    # workgroups take the CUs the first one's thinning bounce chains leave); host wall per frame against the blocking rt_render_frame
    if hasattr(rt.lib, "rt_frame_begin"):
        import numpy as np
        img = np.zeros((h, w, 3), np.uint8)
        image, _keep = rt.scene.make_image(img)
        image.pixels.data = img.ctypes.data
        n = 4 * frames

        def run(pipelined):
            t0 = time.perf_counter()
            pending = []
            for _ in range(n):
                if pipelined:
                    if len(pending) == 2 and rt.lib.rt_frame_end(pending.pop(0)) != 0:
                        return None
                    pending.append(rt.lib.rt_frame_begin(C.byref(hs.scene), C.byref(image), s, b))
                    if pending[-1] < 0:
                        return None
                elif rt.lib.rt_render_frame(C.byref(hs.scene), C.byref(image), s, b, None, None) != 0:
                    return None
            for t in pending:
                if rt.lib.rt_frame_end(t) != 0:
                    return None
            return (time.perf_counter() - t0) * 1e3 / n
        run(True)
        blocking_ms, pipelined_ms = run(False), run(True)
        if blocking_ms is None or pipelined_ms is None:
            return {"error": rt.last_error()}
        out["frames_in_flight"] = {"blocking_ms_per_frame": blocking_ms, "two_in_flight_ms_per_frame": pipelined_ms, "frames": n,
                                   "note": "host wall per frame incl. the copy to the host; rt_frame_begin / rt_frame_end (rt_hip.h), "
                                           "same pixels as the blocking call (tests/test_gpu_frames_in_flight.py)"}
    return out


def dry_run(args, world, rank):
    """CPU-only rehearsal of the N-rank plumbing: rendezvous, partition tables, tile exchange, MAX-over-ranks
    timing, one line from rank 0.  No kernel runs and the line says so."""
    import torch
    import torch.distributed as dist
    from raytracing_c_amd.multi_gpu import FramePartition, gather_tiles
    if world > 1:
        dist.init_process_group("gloo")
    part = FramePartition(args.width or 1920, args.height or 1080, world)
    if args.dry_run_sleep > 0:
        print(f"rank {rank} pid {os.getpid()} sleeping", file=sys.stderr, flush=True)
        time.sleep(args.dry_run_sleep)
    tiles = torch.full((part.max_local, 1024 * 3), rank, dtype=torch.uint8)
    t0 = time.perf_counter()
    got = gather_tiles(tiles, world, rank) if world > 1 else tiles[None]
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = rank != 0 or all(int(got[r][0, 0]) == r for r in range(world))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise RuntimeError("dry run: gathered tiles are not rank-major")
    if rank == 0:
        return {"metric": "Mray/s", "value": None, "unit": "Mray/s", "n_gpus": world, "steps": 0, "warmup": 0,
                "dry_run": True, "chunks_per_rank": [part.n_local(r) for r in range(world)]}
    return None


def main():
    args = parse_args()
    # Environment of the HIP / RCCL runtimes is fixed BEFORE anything imports torch or touches the GPU: the host
    # driver of this pool only supports dmabuf IPC, and the runtime reads the variable once at initialisation.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)            # does not return
    # stdout carries exactly ONE line (the JSON): libraries that print banners to fd 1 (RCCL prints its
    # version block there at communicator creation) are sent to stderr for the whole run.
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    if args.dry_run:
        out = dry_run(args, world, rank)
        if out is not None:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        return

    import numpy as np
    import torch
    import raytracing_c_amd as rt
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    from raytracing_c_amd.multi_gpu import FramePartition, gather_tiles

    # Rehearsal knobs (one-GPU boxes): RT_BENCH_DEVICE pins every rank to one device and
    # RT_BENCH_BACKEND=gloo moves the tile gather through host memory; the driver's runs use
    # neither (one rank per GPU, RCCL).
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if "RT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["RT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if rt.lib.rt_init(local_rank) != 0:
        raise RuntimeError("rt_init: " + rt.last_error())
    wavefront = False
    if args.pipeline == "wavefront":
        # measurement only: the wavefront pipeline exists in the diagnostic library, not in the product (include/rt_hip_diag.h)
        if not rt.native.LIB_PATH.endswith("librt_hip_diag.so"):
            raise SystemExit("--pipeline wavefront needs RT_LIB_PATH=raytracing_c_amd/librt_hip_diag.so (make -C raytracing_c_amd/csrc diag)")
        rt.lib.rt_set_pipeline(1)
        wavefront = True
    dist = None
    force_dist = world == 1 and os.environ.get("RT_BENCH_FORCE_DIST") == "1"     # rehearsal: RCCL path with one rank
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist_
        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    hs, cfg = load_config(args.config)
    if args.width:
        cfg["width"] = args.width
    if args.height:
        cfg["height"] = args.height
    if args.samples:
        cfg["samples"] = args.samples
    if args.bounces:
        cfg["max_bounces"] = args.bounces
    w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]

    dscene = rt.lib.rt_scene_upload(C.byref(hs.scene))
    if not dscene:
        raise RuntimeError("rt_scene_upload: " + rt.last_error())

    dev = torch.device("cuda", local_rank)
    max_local = FramePartition(w, h, world).max_local
    accum = torch.zeros((h, w, 3), dtype=torch.int64, device=dev)
    # the finished frame leaves through a copy stream (double-buffered), so the 6 MB device->host copy of frame k
    # overlaps the path kernel of frame k+1 instead of sitting between two kernels
    images = [torch.zeros((h, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    host_images = [torch.zeros((h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    ev_frame = [torch.cuda.Event() for _ in range(2)]
    ev_copied = [torch.cuda.Event() for _ in range(2)]
    frame_no = [0]
    tiles = torch.zeros((max_local, 1024 * 3), dtype=torch.uint8, device=dev)
    multi = world > 1 or force_dist
    params = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, rank, world, args.slab, 0)

    # Multi-GPU frames are pipelined over two streams and two sets of buffers: the tile exchange of frame k (gather to
    # rank 0 over xGMI, untile, D2H) runs on `post_stream` beside the path kernel of frame k + 1, so a frame costs
    # max(kernel, exchange) instead of their sum (the exchange is ~0.3 ms of a ~6.7 ms rank launch at 8 GPUs).
    accums = [accum, torch.zeros_like(accum)] if multi else [accum]
    tile_bufs = [tiles, torch.zeros_like(tiles)] if multi else [tiles]
    post_stream = torch.cuda.Stream(device=dev) if multi else None
    # ... and the path kernels of consecutive frames run on two streams with a device scene each (launch state -- work
    # counters, schedule feedback, parked hits -- belongs to the device scene): a launch ends ~0.7 ms after its last unit
    # is handed out, while the bounce chains of its last paths finish; that is 2 % of a whole frame but 13 % of a rank's
    # launch at 8 GPUs, and the next frame's workgroups take the CUs as they fall idle.  RT_BENCH_OVERLAP=0: one stream.
    overlap = multi and os.environ.get("RT_BENCH_OVERLAP", "1") != "0"
    dscenes = [dscene]
    render_streams = [None, None]
    if overlap:
        second = rt.lib.rt_scene_upload(C.byref(hs.scene))
        if not second:
            raise RuntimeError("rt_scene_upload: " + rt.last_error())
        dscenes.append(second)
        # two streams on ONE hardware queue run their kernels one after the other, and the runtime multiplexes a process's
        # streams onto GPU_MAX_HW_QUEUES (4) queues per priority: the two render streams take different priorities, so they
        # never share one whatever else (copy / exchange / RCCL streams) the process creates (profiles/r05_frames_in_flight.md)
        prio = [0, -1] if os.environ.get("RT_BENCH_STREAM_PRIO", "1") != "0" else [0, 0]
        render_streams = [torch.cuda.Stream(device=dev, priority=prio[0]), torch.cuda.Stream(device=dev, priority=prio[1])]
    ev_rendered = [torch.cuda.Event() for _ in range(2)]
    ev_posted = [torch.cuda.Event() for _ in range(2)]

    def step():
        cur = torch.cuda.current_stream()
        stream = cur.cuda_stream
        buf = frame_no[0] & 1
        frame_no[0] += 1
        image = images[buf]
        if not multi:
            if frame_no[0] > 2:
                cur.wait_event(ev_copied[buf])      # the image buffer's previous contents have reached the host
            accum.zero_()
            if rt.lib.rt_render_accumulate(dscene, C.byref(params), accum.data_ptr(), stream) != 0:
                raise RuntimeError(rt.last_error())
            if rt.lib.rt_resolve(C.byref(params), accum.data_ptr(), None, image.data_ptr(), None, stream) != 0:
                raise RuntimeError(rt.last_error())
            ev_frame[buf].record()
            copy_stream.wait_event(ev_frame[buf])
            with torch.cuda.stream(copy_stream):
                host_images[buf].copy_(image, non_blocking=True)
                ev_copied[buf].record()
            return
        acc, tl = accums[buf], tile_bufs[buf]
        rs = render_streams[buf] if overlap else cur
        with torch.cuda.stream(rs):
            if frame_no[0] > 2:
                rs.wait_event(ev_posted[buf])       # frame k - 2 has left these buffers
            elif overlap:
                rs.wait_stream(cur)                 # (the first use of the stream: after everything enqueued so far)
            acc.zero_()
            if rt.lib.rt_render_accumulate(dscenes[buf if overlap else 0], C.byref(params), acc.data_ptr(), rs.cuda_stream) != 0:
                raise RuntimeError(rt.last_error())
            if rt.lib.rt_resolve(C.byref(params), acc.data_ptr(), tl.data_ptr(), None, None, rs.cuda_stream) != 0:
                raise RuntimeError(rt.last_error())
            ev_rendered[buf].record(rs)
        post_stream.wait_event(ev_rendered[buf])
        with torch.cuda.stream(post_stream):
            # framebuffer tiles of every rank -> rank 0 (ONE gather: grouped send / recv over xGMI, every peer on its
            # own link into rank 0; 6 MB in total at 1080p)
            if backend == "nccl":
                all_tiles = gather_tiles(tl, world, rank)
            else:
                all_tiles = gather_tiles(tl.cpu(), world, rank)
                if rank == 0:
                    all_tiles = all_tiles.to(dev)
            if rank == 0:
                if rt.lib.rt_untile(w, h, world, all_tiles.data_ptr(), image.data_ptr(), post_stream.cuda_stream) != 0:
                    raise RuntimeError(rt.last_error())
                host_images[buf].copy_(image, non_blocking=True)
            ev_posted[buf].record(post_stream)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    rt.lib.rt_kernel_timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0

    # mean path-kernel time per launch over the timed region (HIP events recorded on the launch
    # stream around every launch) and the in-kernel counters of the last frame
    n_launches = C.c_int32(0)
    last_ms = float(rt.lib.rt_kernel_timing_mean_ms(C.byref(n_launches)))
    kernel_ms_source = "HIP events around every launch of the timed region"
    if overlap:
        # with consecutive frames' kernels on two streams an event interval also holds the time the OTHER frame's kernel kept
        # the CUs: the per-launch time the roofline uses comes from two launches on one stream after the timed region
        rt.lib.rt_kernel_timing_reset()
        for _ in range(2):
            accums[0].zero_()
            if rt.lib.rt_render_accumulate(dscenes[0], C.byref(params), accums[0].data_ptr(), torch.cuda.current_stream().cuda_stream) != 0:
                raise RuntimeError(rt.last_error())
            torch.cuda.synchronize()
        last_ms = float(rt.lib.rt_kernel_timing_mean_ms(C.byref(n_launches)))
        kernel_ms_source = "2 launches on ONE stream after the timed region (the timed region overlaps consecutive frames' kernels)"
    cnt = rt.render.get_counters()
    skipped = C.c_uint64(0)
    if rt.lib.rt_get_skipped_root_visits(C.byref(skipped)) != 0:
        raise RuntimeError(rt.last_error())
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    stats = torch.tensor([cnt.paths, cnt.rays, cnt.node_visits, cnt.leaf_visits, cnt.shades, cnt.backgrounds,
                          cnt.textured, int(skipped.value)], dtype=torch.int64, device=dev)
    kms = torch.tensor([last_ms], dtype=torch.float64, device=dev)
    if dist is not None:
        if backend != "nccl":
            t, stats, kms = t.cpu(), stats.cpu(), kms.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    tot = rt.render.Counters(*[int(v) for v in stats.tolist()[:7]])
    skipped_total = int(stats.tolist()[7])
    last_ms = float(kms.item())

    # N > 1: what rank 0 holds after the gather must BE the frame -- the same bytes as one rank rendering every chunk (per-path
    # seeds depend on pixel and sample only).  Checked after the timed region, reported in the line, and fatal when it fails.
    dist_check = None
    if rank == 0 and multi:
        import hashlib
        gathered = host_images[(frame_no[0] - 1) & 1].numpy()
        whole = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, 0, 1, args.slab, 0)
        acc1, img1 = torch.zeros_like(accums[0]), torch.zeros_like(images[0])
        st = torch.cuda.current_stream().cuda_stream
        if rt.lib.rt_render_accumulate(dscenes[0], C.byref(whole), acc1.data_ptr(), st) != 0 or \
                rt.lib.rt_resolve(C.byref(whole), acc1.data_ptr(), None, img1.data_ptr(), None, st) != 0:
            raise RuntimeError(rt.last_error())
        torch.cuda.synchronize()
        sha_n = hashlib.sha256(gathered.tobytes()).hexdigest()
        sha_1 = hashlib.sha256(img1.cpu().numpy().tobytes()).hexdigest()
        dist_check = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "devices": sorted({local_rank}) if "RT_BENCH_DEVICE" in os.environ else "one per rank (LOCAL_RANK)",
                      "image_sha256": sha_n, "one_rank_image_sha256": sha_1, "equal": sha_n == sha_1}
        del acc1, img1

    if rank == 0:
        sec_per_step = elapsed / max(args.steps, 1)
        rays = tot.rays
        mrays = rays / sec_per_step / 1e6
        # the dominant kernel (rt_path_kernel): at N > 1 every rank launches one kernel on its share of
        # the chunks, the slowest rank's time is used.
        rays_per_launch = rays / world
        workload = f"{cfg['asset']} {w}x{h}, {s} spp, {b} bounces"
        from raytracing_c_amd.configs import CONFIGS
        names = ["spheres", "quad", "helmet", "tower", "helmet4k"]
        if args.config in names and tuple(CONFIGS[args.config][1:5]) == (w, h, s, b):
            workload += f" (BASELINE.json configs[{names.index(args.config)}])"
        prof = measured_profile(workload)
        pipeline = "wavefront pipeline (rt_wf_camera / shade / trace kernels; diagnostic library)" if wavefront else \
            "rt_path_kernel_stream<16, true, 1, short reciprocal> (the product's one path kernel)"
        if prof is not None and wavefront:
            prof = None                                 # the committed profiles are of the tile-stream kernel
        out = {
            "metric": "Mray/s", "value": mrays, "unit": "Mray/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": sec_per_step * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "the reference's own model file (geometry + textures) under a procedural environment map; no dataset involved",
            "config": {"workload": workload,
                       "scene": f"assets/{cfg['asset']}" + (" = self-contained models/helmet.gltf" if "helmet" in cfg["asset"] else "")
                                + "; procedural 2048x1024 equirect background (background.png is a missing blob); seed 0x1234ABCD",
                       "bvh": {"headline": "scene_init (the reference's fixed-capacity split, scene.c:311-414)",
                               "reference": {"mray_per_s": mrays, "node_visits_per_ray": tot.node_visits / max(rays, 1),
                                             "leaf_visits_per_ray": tot.leaf_visits / max(rays, 1),
                                             "build_ms": hs.scene_init_seconds * 1e3}},
                       "partition": f"32x32 chunks dealt to {world} GPU(s) by the (cx + B cy) mod world lattice, "
                                    f"{'RCCL' if backend == 'nccl' else backend} gather of u8 tiles to rank 0"
                                    + ("; consecutive frames' path kernels on two streams" if overlap else "")
                                    if world > 1 else "single GPU"},
            "fps": 1.0 / sec_per_step,
            "msample_per_s": w * h * s / sec_per_step / 1e6,
            "rays_per_frame": rays,
            "rays_per_path": rays / max(tot.paths, 1),
            "node_visits_per_ray": tot.node_visits / max(rays, 1),
            "leaf_visits_per_ray": tot.leaf_visits / max(rays, 1),
            "shades_per_ray": tot.shades / max(rays, 1),
            "kernel_ms": last_ms,
            "roofline": roofline_block(tot, rays_per_launch, last_ms, n_launches.value, prof, pipeline, kernel_ms_source,
                                       profile_scale=(rays_per_launch / prof[0]["rays_per_launch"])
                                       if prof and world > 1 and prof[0].get("rays_per_launch") else 1.0,
                                       skipped_root_visits=skipped_total),
        }
        if world == 1 and not args.no_bvh_compare:
            out["config"]["bvh"]["sah"] = bvh_compare(rt, abi, args, cfg, host_images[(frame_no[0] - 1) & 1].numpy())
        if world == 1 and not args.no_bvh_compare:
            out["config"]["boundary"] = boundary_cost(rt, abi, hs)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hs, cfg, args.cpu_seconds)
        if dist_check is not None:
            out["distributed"] = dist_check
        if args.save:
            from PIL import Image
            Image.fromarray(host_images[(frame_no[0] - 1) & 1].numpy()).save(args.save)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    for dsc in dscenes:
        rt.lib.rt_scene_release(dsc)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if dist_check is not None and not dist_check["equal"]:
        print(f"bench.py: the gathered frame of {world} ranks differs from the one-rank frame: {dist_check}", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()

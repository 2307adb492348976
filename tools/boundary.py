#!/usr/bin/env python3
"""Cost of the drop-in boundary: frames through render_thread_proc from a C host (examples/driver_min, the reference's own
protocol, driver.c:793-818) at the reference's default frame (driver.c:733-742: 1024 x 1024, 16 spp, 8 bounces, here on the
helmet) and at BASELINE config #1 (spheres 256 x 256, 16 spp, 4 bounces), split into host and GPU phases by
rt_get_frame_timing().  Prints markdown (committed as profiles/<tag>_boundary.md).
    python tools/boundary.py [frames]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_c_driver import _build, _dump          # noqa: E402


def main():
    import pathlib
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    exe = _build()
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="boundary_"))
    rows = []
    for label, cfg, w, h, s, b, threads, devices in (("driver defaults (driver.c:733-742) on the helmet", "helmet", 1024, 1024, 16, 8, 1, 1),
                                                     ("the same, `-T 8`", "helmet", 1024, 1024, 16, 8, 8, 1),
                                                     ("the same, `-T 8`, `RT_DEVICES=2` rehearsed on this one GPU", "helmet", 1024, 1024, 16, 8, 8, 2),
                                                     ("the same, `-T 8`, `RT_DEVICES=4` rehearsed", "helmet", 1024, 1024, 16, 8, 8, 4),
                                                     ("the same, `-T 8`, `RT_DEVICES=8` rehearsed", "helmet", 1024, 1024, 16, 8, 8, 8),
                                                     ("BASELINE config #1", "spheres", 256, 256, 16, 4, 1, 1),
                                                     ("BASELINE config #3", "helmet", 1920, 1080, 256, 8, 1, 1),
                                                     ("BASELINE config #3, `RT_DEVICES=8` rehearsed", "helmet", 1920, 1080, 256, 8, 8, 8)):
        scene = _dump(tmp, cfg)
        env = dict(os.environ, DRIVER_MIN_FRAMES=str(frames))
        if devices > 1:
            env.update(RT_DEVICES=str(devices), RT_DEVICES_REHEARSE="1")
        r = subprocess.run([exe, scene, str(w), str(h), str(s), str(b), str(threads), str(tmp / "o.ppm")], capture_output=True,
                           text=True, env=env, timeout=600)
        if r.returncode != 0:
            print(r.stdout, r.stderr, file=sys.stderr)
            raise SystemExit(1)
        pat = re.compile(r"frame (\d+): host wall ([\d.]+) ms .* library total ([\d.]+) = stamp ([\d.]+) \+ upload ([\d.]+) \+ enqueue ([\d.]+)"
                         r".*clear\+prepare ([\d.]+), path kernel ([\d.]+), resolve ([\d.]+), copy to host ([\d.]+) \| verify ([\d.]+) .*"
                         r"devices (\d+), slowest (\d+), gather ([\d.]+)")
        vals = [[float(x) for x in m.groups()[1:]] for m in (pat.search(l) for l in r.stdout.splitlines()) if m]
        first, rest = vals[0], vals[2:]                # frame 0 uploads the scene, frame 1 has no schedule feedback yet
        med = [sorted(col)[len(col) // 2] for col in zip(*rest)]
        rows.append((label, f"{w}x{h}, {s} spp, {b} bounces", first, med))
    print("| workload | frame | host wall | library total | stamp | enqueue | GPU clear + prepare | GPU path kernel | GPU resolve | GPU copy (to host; N devices: a device's tiles to device 0) | "
          "full scene check (host, while the GPU renders) | gather + untile + copy out (N devices) | first frame (uploads the scene): total / upload |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for label, frame, first, m in rows:
        wall, total, stamp, upload, enq, prep, path, res, copy, verify, ndev, slowest, gather = m
        print(f"| {label} | {frame} | {wall:.3f} | {total:.3f} | {stamp:.3f} | {enq:.3f} | {prep:.3f} | {path:.3f} | {res:.3f} | {copy:.3f} | "
              f"{verify:.3f} | {gather:.3f} | {first[1]:.1f} / {first[3]:.1f} |")
    print(f"\n(median of frames 2 .. {frames - 1} of one process, milliseconds; `host wall` = thread start -> rendering_context_is_finished as "
          "the C host sees it, polling every 20 us; `library total` = the owner's call into render_thread_proc.  Rows with RT_DEVICES: N logical devices "
          "rehearsed on this box's ONE GPU -- their kernels share it, so the GPU columns are the slowest device's SHARE of the frame, and the row shows "
          "what N devices cost the library on top of the one-device frame, not a speed-up)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU fuzz (one-off assurance; a script, not collected by pytest): random triangle soups, random cameras (far, near, inside, looking
away), random frame shapes / sample counts / bounce limits / builders / scene scales; the HIP path through the C-ABI must give
the oracle's radiance sums and counters bit for bit.    python tests/fuzz_parity.py [seconds] [first seed]   (RT_FUZZ_LARGE=1: frames of 200 ... 900 pixels; RT_FUZZ_LANES=1: every frame also through the two frame lanes)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                     # noqa: E402
import raytracing_c_amd as rt                          # noqa: E402
from tests import _oracle                              # noqa: E402
from tests.test_gpu_random_scenes import make_scene, _look_at   # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
assert rt.lib.rt_init(0) == 0, rt.last_error()
_oracle.load()
t0 = time.time()
n = bad = 0
seed = seed0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    n_tris = int(rng.choice([3, 9, 40, 200, 700, 2500, 6000]))
    scale = float(rng.choice([1.0, 1.0, 1.0, 1e-3, 37.0, 2.0 ** 20]))
    builder = str(rng.choice(["reference", "reference", "sah"]))
    hs = make_scene(seed, n_tris, builder=builder, scale=scale)
    mode = int(rng.integers(0, 4))
    c = rng.uniform(-1, 1, 3) * scale
    if mode == 0:      # far, looking at the soup
        eye = c + rng.normal(size=3) * scale * rng.uniform(2, 8)
        tgt = rng.uniform(-0.5, 0.5, 3) * scale
    elif mode == 1:    # inside
        eye = rng.uniform(-0.6, 0.6, 3) * scale
        tgt = eye + rng.normal(size=3) * scale
    elif mode == 2:    # grazing past the soup
        eye = c + rng.normal(size=3) * scale * 3
        tgt = eye + np.cross(eye, rng.normal(size=3))
    else:              # looking away
        eye = c + rng.normal(size=3) * scale * 4
        tgt = eye * 2.0
    hs.set_camera(_look_at(eye, tgt), float(rng.uniform(0.2, 2.2)))
    if os.environ.get("RT_FUZZ_LARGE"):     # frames whose 8x8 tiles are small against the geometry: pyramid culling at every level
        w, h = int(rng.integers(200, 900)), int(rng.integers(120, 500))
        s = int(rng.choice([4, 8, 32]))
        b = int(rng.choice([1, 4, 8]))
        cap = 3000000
    else:
        w, h = int(rng.integers(9, 140)), int(rng.integers(9, 90))
        s = int(rng.choice([1, 2, 5, 16, 33, 64]))
        b = int(rng.choice([0, 1, 3, 8, 40]))
        cap = 400000
    if w * h * s > cap:
        s = max(1, cap // (w * h))
    want = _oracle.render(hs, w, h, s, b, seed=seed)
    got = rt.render_frame(hs, w, h, s, b, seed=seed, want_accum=True)
    ok = np.array_equal(want["accum"], got["accum"])
    cn = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        ok = ok and want["counters"][k] == getattr(cn, k)
    if os.environ.get("RT_FUZZ_LANES"):     # ... and the same frame twice through rt_frame_begin / rt_frame_end, both lanes busy
        ta, outa, ka = rt.frame_begin(hs, w, h, s, b, seed=seed)
        tb, outb, kb = rt.frame_begin(hs, w, h, s, b, seed=seed ^ 0x5555)
        ca = rt.frame_end(ta)
        rt.frame_end(tb)
        ok = ok and np.array_equal(outa, got["image"]) and ca.rays == cn.rays and ca.node_visits == cn.node_visits
        ok = ok and np.array_equal(outb, rt.render_frame(hs, w, h, s, b, seed=seed ^ 0x5555)["image"])
    n += 1
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed}: tris {n_tris} scale {scale} builder {builder} mode {mode} {w}x{h} s{s} b{b}", flush=True)
    if n % 25 == 0:
        print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    seed += 1
print(f"done: {n} cases (seeds {seed0}..{seed - 1}), {bad} mismatches")
sys.exit(1 if bad else 0)

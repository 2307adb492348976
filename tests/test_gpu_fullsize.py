"""Parity at BASELINE.json's full sizes.

Config #3 (the bench workload) is compared with the oracle in full; the bigger configs (#4, #5: 1 and 8.5 G paths)
are checked through properties that do not depend on size: bit-exact parity on WINDOWS of the full frame (the oracle renders only the window,
with the full frame's width/height/spp so rays, seeds and jitter are the full frame's), additivity of
the fixed-point accumulation over sample ranges, independence from the GPU partition and from
scheduling, exact path counts, and a checksum of checksums.
"""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import raytracing_c_amd as rt
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    return rt


class Frame:
    """rt_render_accumulate on an uploaded scene into a torch buffer."""

    def __init__(self, rt, name):
        import torch
        from raytracing_c_amd.configs import load_config
        self.rt, self.torch = rt, torch
        self.hs, self.cfg = load_config(name)
        self.d = rt.lib.rt_scene_upload(C.byref(self.hs.scene))
        assert self.d, rt.last_error()

    def accumulate(self, w, h, s, b, accum=None, rank=0, world=1, first=0, count=0, seed=0x1234ABCD, slab=0):
        from raytracing_c_amd import ctypes_abi as abi
        torch = self.torch
        if accum is None:
            accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
        p = abi.RT_Render_Params(w, h, s, b, seed, rank, world, slab, 0, first, count)
        assert self.rt.lib.rt_render_accumulate(self.d, C.byref(p), accum.data_ptr(), None) == 0, self.rt.last_error()
        torch.cuda.synchronize()
        return accum

    def close(self):
        self.rt.lib.rt_scene_release(self.d)


@pytest.fixture(scope="module")
def helmet(rt):
    f = Frame(rt, "helmet")
    yield f
    f.close()


def _np(t):
    return t.cpu().numpy().view(np.uint64)


def test_config3_full_frame_windows_match_oracle(rt, helmet):
    """helmet 1920x1080, 256 spp, 8 bounces: four windows (sky, silhouette, visor, ground) bit-exact."""
    from tests import _oracle
    w, h, s, b = 1920, 1080, 256, 8
    acc = _np(helmet.accumulate(w, h, s, b))
    c = rt.render.get_counters()
    assert c.paths == w * h * s
    assert c.rays == c.backgrounds + (c.rays - c.backgrounds) and c.rays >= c.paths
    assert c.backgrounds <= c.paths and c.shades <= c.rays and c.textured == c.shades      # helmet: one textured material
    for (x0, y0, x1, y1) in [(0, 0, 24, 16), (700, 300, 724, 316), (1000, 500, 1016, 524), (1900, 1064, 1920, 1080)]:
        want = _oracle.render(helmet.hs, w, h, s, b, window=(x0, y0, x1, y1))["accum"]
        assert np.array_equal(acc[y0:y1, x0:x1], want[y0:y1, x0:x1]), (x0, y0)
    # the frame is not blank: sky rows differ from ground rows, and the centre (helmet) from both
    lum = acc.astype(np.float64).sum(-1)
    assert lum[:100].mean() > 1.5 * lum[-100:].mean()
    assert abs(lum[450:650, 900:1000].mean() / lum[:100].mean() - 1) > 0.2


def _host_threads():
    import os
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            threads = max(1, min(threads, int(round(int(q[0]) / int(q[1])))))
    except OSError:
        pass
    return min(threads, 64)


def test_config3_entire_frame_matches_oracle(rt, helmet):
    """BASELINE.json configs[2] in full -- 1920x1080, 256 spp, 8 bounces, 530.8 M paths, 642.5 M rays: every one of the
    6.2 M fixed-point radiance sums, the u8 image and all seven counters equal the CPU oracle's.  (The oracle needs
    about 8 s for the frame on the 16 host CPUs of a GPU box; threads = the CPUs this process may use.)"""
    from tests import _oracle
    w, h, s, b = 1920, 1080, 256, 8
    want = _oracle.render(helmet.hs, w, h, s, b, n_threads=_host_threads())
    got = rt.render_frame(helmet.hs, w, h, s, b, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"])
    assert np.array_equal(want["image"], got["image"])
    c = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        assert want["counters"][k] == getattr(c, k), k
    assert c.paths == w * h * s == 530841600 and c.rays == 642493048      # (contract v1: 642492860)


def test_config3_additivity_partition_and_determinism(rt, helmet):
    """Fixed-point sums are exact: sample ranges add up, 8 ranks' chunks add up, reruns are identical."""
    w, h, s, b = 1920, 1080, 32, 8          # full frame, reduced spp: the properties are size independent
    full = _np(helmet.accumulate(w, h, s, b))
    again = _np(helmet.accumulate(w, h, s, b, slab=4))
    assert hashlib.sha256(full.tobytes()).hexdigest() == hashlib.sha256(again.tobytes()).hexdigest()
    # progressive: [0,10) then [10,32) into the same buffer
    acc = helmet.accumulate(w, h, s, b, first=0, count=10)
    acc = helmet.accumulate(w, h, s, b, accum=acc, first=10, count=22)
    assert np.array_equal(_np(acc), full)
    # the 8-GPU partition of the scaling bench, executed rank after rank on this GPU
    acc = None
    paths = 0
    for rank in range(8):
        acc = helmet.accumulate(w, h, s, b, accum=acc, rank=rank, world=8)
        paths += rt.render.get_counters().paths
    assert paths == w * h * s
    assert np.array_equal(_np(acc), full)
    # checksum of checksums: per-chunk sums of the partitioned run equal those of the single run
    a, f = _np(acc), full
    assert a.sum(dtype=np.uint64) == f.sum(dtype=np.uint64)


def test_config4_tower_window(rt):
    """tower 1920x1080, 512 spp, 12 bounces (BASELINE.json configs[3]): one window bit-exact, path count exact."""
    from tests import _oracle
    f = Frame(rt, "tower")
    try:
        w, h, s, b = 1920, 1080, 512, 12
        acc = _np(f.accumulate(w, h, s, b))
        assert rt.render.get_counters().paths == w * h * s
        x0, y0, x1, y1 = 940, 520, 956, 536
        want = _oracle.render(f.hs, w, h, s, b, window=(x0, y0, x1, y1))["accum"]
        assert np.array_equal(acc[y0:y1, x0:x1], want[y0:y1, x0:x1])
    finally:
        f.close()


def test_config4_entire_frame_matches_oracle(rt):
    """BASELINE.json configs[3] in full on one GPU -- tower 1920x1080, 512 spp, 12 bounces, 1.06 G paths: all radiance
    sums, the image and the counters equal the oracle's."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("tower")
    w, h, s, b = 1920, 1080, 512, 12
    want = _oracle.render(hs, w, h, s, b, n_threads=_host_threads())
    got = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"])
    assert np.array_equal(want["image"], got["image"])
    c = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        assert want["counters"][k] == getattr(c, k), k
    assert c.paths == w * h * s


def test_config5_4k_window(rt, helmet):
    """helmet 3840x2160, 16 bounces (BASELINE.json configs[4]) at 64 of its 1024 spp: window bit-exact.
    (The sample range API renders [0, 64) of 1024; seeds and jitter are those of the 1024-spp frame.)"""
    from tests import _oracle
    w, h, s, b = 3840, 2160, 1024, 16
    acc = _np(helmet.accumulate(w, h, s, b, first=0, count=64))
    assert rt.render.get_counters().paths == w * h * 64
    x0, y0, x1, y1 = 1700, 900, 1716, 912
    want = _oracle.render(helmet.hs, w, h, s, b, window=(x0, y0, x1, y1), sample_range=(0, 64))["accum"]
    assert np.array_equal(acc[y0:y1, x0:x1], want[y0:y1, x0:x1])


def test_config5_entire_frame_full_spp(rt, helmet):
    """BASELINE.json configs[4] IN FULL on one GPU -- helmet 3840x2160, 1024 spp, 16 bounces: 8 493 465 600 paths in one
    launch (render_thread_proc's loop of raytracer.c:596-720 over the whole frame, about 0.75 s).  Checked through
    size-independent properties: exact path count; two oracle windows at the full 1024 spp, bit-exact (the oracle renders
    only the window, with the full frame's size and spp, so seeds, jitter and rays are the full frame's); the sum of 16
    sample ranges of 64 equals the single launch on every one of the 24.9 M radiance sums; rays/paths/backgrounds
    consistent."""
    from tests import _oracle
    torch = helmet.torch
    w, h, s, b = 3840, 2160, 1024, 16
    full_t = helmet.accumulate(w, h, s, b)
    c = rt.render.get_counters()
    assert c.paths == w * h * s == 8493465600
    assert c.rays >= c.paths and c.backgrounds <= c.paths and c.textured == c.shades
    assert c.rays == c.paths + (c.rays - c.paths) and c.node_visits >= c.rays          # every ray enters the root
    rays_full = c.rays
    full = _np(full_t)
    for (x0, y0, x1, y1) in [(1700, 900, 1712, 908), (2300, 1300, 2308, 1308)]:        # visor / lower shell: deep paths
        want = _oracle.render(helmet.hs, w, h, s, b, window=(x0, y0, x1, y1), n_threads=_host_threads())["accum"]
        assert np.array_equal(full[y0:y1, x0:x1], want[y0:y1, x0:x1]), (x0, y0)
    del full
    # 16 launches of 64 samples each into one buffer == the single launch (integer accumulation is exact)
    acc = None
    rays = 0
    for k in range(16):
        acc = helmet.accumulate(w, h, s, b, accum=acc, first=64 * k, count=64)
        rays += rt.render.get_counters().rays
    assert rays == rays_full
    assert bool(torch.equal(acc, full_t))


def test_config2_quad_full(rt):
    """quad 512x512, 64 spp, 4 bounces (BASELINE.json configs[1], depth-0 BVH): the WHOLE frame bit-exact."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, cfg = load_config("quad")
    w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]
    want = _oracle.render(hs, w, h, s, b, n_threads=16)
    got = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"])
    assert np.array_equal(want["image"], got["image"])


def test_config1_spheres_full(rt):
    """spheres 256x256, 16 spp, 4 bounces (BASELINE.json configs[0]): the WHOLE frame bit-exact."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, cfg = load_config("spheres")
    w, h, s, b = cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"]
    want = _oracle.render(hs, w, h, s, b, n_threads=16)
    got = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"])
    assert np.array_equal(want["image"], got["image"])

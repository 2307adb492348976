"""GPU parity: the HIP path (through the C-ABI of librt_hip.so) against the CPU oracle.

Bar: BIT-EXACT.  The kernels and the oracle share include/rt_math.h, both are compiled with
-ffp-contract=off, and radiance is accumulated in 32.32 fixed point, so the u64 sums, the fp32
means, the u8 image and the ray counters must be identical (north_star's 1e-4 RMS is implied).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


@pytest.fixture(scope="module")
def rt():
    import raytracing_c_amd as rt
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    return rt


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---------------------------------------------------------------------------------------
# numeric contract

MATH_CASES = [
    (0, "log", lambda r: r.uniform(1e-6, 1e3, 200000), None),
    (1, "exp", lambda r: r.uniform(-80, 80, 200000), None),
    (2, "pow", lambda r: r.uniform(0, 1.2, 200000), lambda r: r.choice([2.4, 1 / 2.4, 5.0, 0.5], 200000)),
    (3, "sin", lambda r: r.uniform(0, 6.2832, 200000), None),
    (4, "cos", lambda r: r.uniform(0, 6.2832, 200000), None),
    (5, "atan2", lambda r: r.uniform(-1, 1, 200000), lambda r: r.uniform(-1, 1, 200000)),
    (6, "asin", lambda r: r.uniform(-1.01, 1.01, 200000), None),
    (7, "srgb_to_linear", lambda r: r.uniform(0, 1, 200000), None),
    (8, "linear_to_srgb", lambda r: r.uniform(0, 1, 200000), None),
    (9, "sqrt", lambda r: np.concatenate([r.uniform(0, 1e-30, 1000), r.uniform(0, 1e6, 199000)]), None),
    (10, "rcp", lambda r: np.concatenate([r.uniform(-1e-38, 1e-38, 1000), r.uniform(-1e3, 1e3, 199000)]), None),
]


@pytest.mark.parametrize("op,name,gx,gy", MATH_CASES, ids=[c[1] for c in MATH_CASES])
def test_math_bit_exact(rt, oracle, op, name, gx, gy, diag):
    from tests import _oracle
    rng = np.random.default_rng(100 + op)
    x = gx(rng).astype(np.float32)
    y = gy(rng).astype(np.float32) if gy else None
    if op == 5:   # axis cases of atan2
        x[:8] = [0, 0, 0, 1, -1, 0.5, -0.5, 0]
        y[:8] = [0, 1, -1, 0, 0, 0, 0, 0.25]
    want = _oracle.math(op, x, y)
    got = np.zeros_like(x)
    rc = rt.diag.rt_test_math(op, x.size, x.ctypes.data, y.ctypes.data if y is not None else None, got.ctypes.data)
    assert rc == 0, rt.last_error()
    bad = np.nonzero(_bits(want) != _bits(got))[0]
    assert bad.size == 0, f"{name}: {bad.size} mismatches, first x={x[bad[0]]!r} want={want[bad[0]]!r} got={got[bad[0]]!r}"


def test_short_reciprocal_equals_the_division(rt, oracle, diag):
    """Leaf blocks of the tile-stream kernel take 1 / det from rcp_exact() (v_rcp_f32 + one Newton step, scaled by 2^24
    on the way in and out) instead of hipcc's IEEE division sequence.  The claim is equality, not accuracy: every bit
    pattern with |x| < 2^102, both infinities and every NaN is compared on the device with 1.0f / x, and a sample --
    denormals, zeros, infinities, powers of two, all-ones mantissas -- with the CPU's division (the oracle's)."""
    from tests import _oracle
    out = (C.c_uint64 * 6)()
    assert rt.diag.rt_test_rcp_sweep(out) == 0, rt.last_error(rt.diag)
    inside_bad, outside_n, outside_bad, first, leaf_bad, leaf_special_not_nan = (int(v) for v in out)
    assert inside_bad == 0, f"first differing pattern {first - 1:#010x}"
    # the leaf blocks' form without v_div_fixup_f32 (rcp_leaf): the quotient for every finite non-zero |x| < 2^102, NaN for 0 / inf / NaN
    assert leaf_bad == 0 and leaf_special_not_nan == 0
    assert outside_n == 2 * 26 * (1 << 23)           # 2^102 <= |x| < infinity: exponent fields 229..254, both signs
    assert outside_bad > 0                            # ... where it really does differ: the host's bound is needed
    rng = np.random.default_rng(11)
    bits = rng.integers(0, 1 << 32, 400000, dtype=np.uint64).astype(np.uint32)
    special = np.array([0, 0x80000000, 1, 0x007FFFFF, 0x00800000, 0x7F800000, 0xFF800000, 0x7FC00000, 0x3F800000,
                        0x3FFFFFFF, 0x00FFFFFF, 0x727FFFFF, 0x72800000, 0x807FFFFF, 0x80000001], np.uint32)
    bits[:special.size] = special
    bits[special.size:special.size + 50000] &= np.uint32(0x807FFFFF)          # denormals of both signs
    x = bits.view(np.float32)
    inside = ~((np.abs(x) >= np.float32(2.0 ** 102)) & np.isfinite(x))
    want = _oracle.math(10, x, None)
    got = np.zeros_like(x)
    assert rt.diag.rt_test_math(11, x.size, x.ctypes.data, None, got.ctypes.data) == 0, rt.last_error(rt.diag)
    same = (_bits(want) == _bits(got)) | (np.isnan(want) & np.isnan(got))
    assert same[inside].all(), f"x = {x[inside][~same[inside]][:4]!r}"
    assert inside.sum() > 300000 and (~inside).sum() > 1000


def test_shift_quantisation_equals_the_double_one(rt, diag):
    """The tile-stream kernel turns a sample into 32.32 fixed point by shifting the float's mantissa instead of going through
    double (rt_accum_quantize, rt_math.h): the same integer for every one of the 2^32 bit patterns, NaN and negatives (0),
    infinity and anything above 2^20 (clamped) included."""
    out = (C.c_uint64 * 2)()
    assert rt.diag.rt_test_quantize_sweep(out) == 0, rt.last_error(rt.diag)
    assert int(out[0]) == 0, f"first differing pattern {int(out[1]) - 1:#010x}"


def test_texture_srgb_decode_equals_the_division(rt, oracle, diag):
    """The kernels decode a texture sample with (x + 0.055f) * RN(1 / 1.055f) corrected by the exact residual instead of the
    division by 1.055f of common.h:84-91.  Every float in [0, 2] and in [-0.046875, -0.03125] -- a sample lies in [0, 0.9961] --
    gives the same bits as rt_srgb_to_linear1() on the device; a sample of them is compared with the CPU's."""
    from tests import _oracle
    out = (C.c_uint64 * 3)()
    assert rt.diag.rt_test_srgb_sweep(out) == 0, rt.last_error(rt.diag)
    n, bad, first = (int(v) for v in out)
    assert n == (1 << 30) + (1 << 22)
    assert bad == 0, f"first differing pattern {first - 1:#010x}"
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.uniform(0, 1, 200000), rng.uniform(0, 1e-6, 1000), [0.0, 1.0, 0.9961, np.nan]]).astype(np.float32)
    want = _oracle.math(7, x, None)
    got = np.zeros_like(x)
    assert rt.diag.rt_test_math(13, x.size, x.ctypes.data, None, got.ctypes.data) == 0, rt.last_error(rt.diag)
    assert np.array_equal(_bits(want)[:-1], _bits(got)[:-1]) and np.isnan(got[-1]) == np.isnan(want[-1])


# ---------------------------------------------------------------------------------------
# traversal + intersection

def _rays_for(hs, n, rng):
    """Rays from around the scene towards it, plus axis-aligned and degenerate ones."""
    soa = hs.soa_array()
    used = np.any(soa != 0, axis=0)
    pts = np.stack([soa[0][used], soa[3][used], soa[6][used]], 1)
    lo, hi = pts.min(0), pts.max(0)
    c, ext = (lo + hi) / 2, max(float((hi - lo).max()), 1e-3)
    o = c + rng.normal(size=(n, 3)) * ext * 1.5
    tgt = pts[rng.integers(0, len(pts), n)] + rng.normal(size=(n, 3)) * ext * 0.02
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], 1).astype(np.float32)
    # axis-aligned directions (zero components -> inf reciprocals) and origins inside the scene
    k = min(n // 8, 512)
    axes = np.eye(3, dtype=np.float32)
    for i in range(k):
        rays[i, 3:] = axes[i % 3] * (1 if (i // 3) % 2 == 0 else -1)
    rays[k:2 * k, :3] = (c + rng.uniform(-0.4, 0.4, (k, 3)) * ext).astype(np.float32)
    return np.ascontiguousarray(rays)


@pytest.mark.parametrize("asset", ["quad.obj", "fov_test.obj", "sheen.glb", "spheres.glb", "tower.obj", "helmet.glb"])
def test_trace_bit_exact(rt, oracle, asset, diag):
    from raytracing_c_amd.loaders import load_model
    hs = load_model(os.path.join(ASSETS, asset))
    rng = np.random.default_rng(7)
    n = 20000
    rays = _rays_for(hs, n, rng)
    wt, wtri, wuv = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
    oracle.oracle_trace_rays(C.byref(hs.scene), n, rays.ctypes.data, wt.ctypes.data, wtri.ctypes.data, wuv.ctypes.data)
    d = rt.diag.rt_scene_upload(C.byref(hs.scene))
    assert d, rt.last_error()
    try:
        gt, gtri, guv = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
        rc = rt.diag.rt_test_trace(d, n, rays.ctypes.data, gt.ctypes.data, gtri.ctypes.data, guv.ctypes.data)
        assert rc == 0, rt.last_error()
    finally:
        rt.diag.rt_scene_release(d)
    assert (wtri >= 0).sum() > n // 10, "test rays must actually hit the scene"
    assert np.array_equal(wtri, gtri)
    assert np.array_equal(_bits(wt), _bits(gt))
    assert np.array_equal(_bits(wuv), _bits(guv))


def test_leaf_known_answers_on_gpu(rt, diag):
    """The hand-derived cases of tests/test_oracle_kat.py::test_ray_triangles_hit_8_cases on the device:
    closest of several, equal t -> LOWEST slot wins (min_f32x8, raytracer.c:27-29), t < eps and epsilon-padded
    barycentric bounds (raytracer.c:137-149).  A <= 8 triangle scene is a depth-0 BVH: one leaf group, input order."""
    from raytracing_c_amd.background import procedural_background
    from raytracing_c_amd.loaders import camera_from_trs
    from raytracing_c_amd.scene import Material, build_scene
    tri = lambda z: [[0, 0, z], [1, 0, z], [0, 1, z]]          # noqa: E731
    P = np.array([tri(5), tri(3), tri(3), tri(-1), tri(7), tri(4)], np.float32)
    N = np.tile(np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32), (6, 1, 1))
    UV = np.tile(np.array([[0, 0], [1, 0], [0, 1]], np.float32), (6, 1, 1))
    hs = build_scene(P, N, UV, np.zeros(6, int), [Material()], [], camera_from_trs((0, 0, 9)), 1.0, procedural_background(16, 8))
    assert hs.depth == 0 and hs.n_slots == 8
    rays = np.array([[0.25, 0.25, 0, 0, 0, 1],          # hits z=3 twice (slots 1, 2), z=4, z=5, z=7 -> slot 1, t = 3
                     [0.25, 0.25, 7.5, 0, 0, 1],        # everything behind the origin
                     [-0.5e-4, 0.25, 0, 0, 0, 1],       # u = -0.5e-4: inside the epsilon padding
                     [-2e-4, 0.25, 0, 0, 0, 1],         # u = -2e-4: outside
                     [0.25, 0.25, 3.5, 0, 0, 1]], np.float32)   # starts between z=3 and z=4 -> slot 5 (z=4), t = 0.5
    n = len(rays)
    d = rt.diag.rt_scene_upload(C.byref(hs.scene))
    assert d, rt.last_error()
    try:
        t, tri_i, uv = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
        assert rt.diag.rt_test_trace(d, n, rays.ctypes.data, t.ctypes.data, tri_i.ctypes.data, uv.ctypes.data) == 0
    finally:
        rt.diag.rt_scene_release(d)
    assert tri_i.tolist() == [1, -1, 1, -1, 5]
    assert t[0] == 3.0 and np.isinf(t[1]) and t[2] == 3.0 and np.isinf(t[3]) and t[4] == 0.5
    assert uv[0].tolist() == [0.25, 0.25]


def test_texture_bit_exact(rt, oracle, diag):
    from raytracing_c_amd.loaders import load_model
    hs = load_model(os.path.join(ASSETS, "helmet.glb"))
    rng = np.random.default_rng(3)
    n = 20000
    uv = rng.uniform(-3, 3, (n, 2)).astype(np.float32)
    uv[:6] = [[0, 0], [1, 1], [-1, -1], [0.99999994, 0.5], [-1e-9, 0.25], [2.5, -0.5]]
    d = rt.diag.rt_scene_upload(C.byref(hs.scene))
    assert d, rt.last_error()
    try:
        cands = [hs.background_image] + list(hs.images)
        wants = []
        for cimg in cands:
            w = np.zeros((n, 3), np.float32)
            for i in range(n):
                oracle.oracle_sample_texture_bilinear(C.byref(cimg), uv[i, 0], uv[i, 1], w[i].ctypes.data)
            wants.append(w)
        seen = set()
        for tex in [-1] + list(range(len(hs.images) + 1)):
            got = np.zeros((n, 3), np.float32)
            rc = rt.diag.rt_test_texture(d, tex, n, uv.ctypes.data, got.ctypes.data)
            assert rc == 0, rt.last_error()
            # textures are uploaded in first-use order: identify the host image by content
            match = [k for k, w in enumerate(wants) if np.array_equal(_bits(w), _bits(got))]
            assert match, f"texture {tex}: no host image reproduces the device fetch bit-exactly"
            seen.add(match[0])
        assert seen == set(range(len(cands)))
    finally:
        rt.diag.rt_scene_release(d)


# ---------------------------------------------------------------------------------------
# whole frames

FRAMES = [
    # asset, camera config name or None, width, height, samples, bounces, shader
    ("quad", 96, 96, 16, 4, "disney"),
    ("spheres", 96, 96, 16, 4, "disney"),
    ("spheres", 64, 64, 8, 3, "debug"),
    ("helmet", 160, 90, 8, 8, "disney"),
    ("tower", 96, 54, 8, 12, "disney"),
    ("helmet", 70, 45, 3, 5, "disney"),         # ragged: not a multiple of 8 or 32, odd sample count
]


@pytest.mark.parametrize("name,w,h,s,b,shader", FRAMES, ids=[f"{f[0]}-{f[1]}x{f[2]}-{f[3]}spp-{f[5]}" for f in FRAMES])
def test_frame_bit_exact(rt, oracle, name, w, h, s, b, shader):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config(name, shader=shader)
    want = _oracle.render(hs, w, h, s, b)
    got = rt.render_frame(hs, w, h, s, b, want_linear=True, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"]), "fixed-point radiance sums differ"
    assert np.array_equal(_bits(want["linear"]), _bits(got["linear"]))
    assert np.array_equal(want["image"], got["image"])
    c = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        assert want["counters"][k] == getattr(c, k), k
    assert want["image"].std() > 1.0, "frame must not be blank"


def test_seed_changes_image_and_is_reproducible(rt):
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config("spheres")
    a = rt.render_frame(hs, 64, 64, 8, 4, seed=1, want_accum=True)["accum"]
    b = rt.render_frame(hs, 64, 64, 8, 4, seed=1, want_accum=True)["accum"]
    c = rt.render_frame(hs, 64, 64, 8, 4, seed=2, want_accum=True)["accum"]
    assert np.array_equal(a, b)
    assert not np.array_equal(a, c)


def test_render_thread_proc_protocol(rt, oracle):
    """driver.c:793-818: N threads on one context; n_threads reaches 0 only with a complete image."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    want = _oracle.render(hs, 80, 48, 4, 4)["image"]
    for n_threads in (1, 4):
        r = rt.render_context(hs, 80, 48, 4, 4, n_threads=n_threads)
        assert r["finished"] and r["n_threads"] == 0
        assert r["current_chunk"] >= rt.lib.rt_chunk_count(80, 48)
        assert np.array_equal(r["image"], want)


def test_partition_and_untile_match_single_gpu(rt):
    """Chunks interleaved over `world` ranks (all run here on one GPU), compact tiles, untile."""
    import torch
    from raytracing_c_amd.configs import load_config
    from raytracing_c_amd import ctypes_abi as abi
    hs, _ = load_config("spheres")
    w, h, s, b, world = 100, 70, 4, 4, 3
    full = rt.render_frame(hs, w, h, s, b)["image"]
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    assert d, rt.last_error()
    try:
        max_local = rt.lib.rt_max_local_chunk_count(w, h, world)
        all_tiles = torch.zeros((world, max_local, 1024 * 3), dtype=torch.uint8, device="cuda")
        for rank in range(world):
            accum = torch.zeros((h, w, 3), dtype=torch.int64, device="cuda")
            p = abi.RT_Render_Params(w, h, s, b, 0x1234ABCD, rank, world, 0, 0)
            assert rt.lib.rt_render_accumulate(d, C.byref(p), accum.data_ptr(), None) == 0, rt.last_error()
            assert rt.lib.rt_resolve(C.byref(p), accum.data_ptr(), all_tiles[rank].data_ptr(), None, None, None) == 0
            torch.cuda.synchronize()
        image = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        assert rt.lib.rt_untile(w, h, world, all_tiles.data_ptr(), image.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(image.cpu().numpy(), full)
    finally:
        rt.lib.rt_scene_release(d)


def test_unknown_shader_proc_fails_loudly(rt):
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config("quad")
    hs.scene.triangles.aos[0].shader.proc = 0x1234
    rt.lib.rt_clear_error()
    d = rt.lib.rt_scene_upload(C.byref(hs.scene))
    assert not d
    assert "shader proc" in rt.last_error()

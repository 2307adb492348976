"""Known-answer tests that pin the CPU oracle (oracle/oracle.c) and include/rt_math.h.

The reference ships no tests or golden vectors, so every expected value
here is derived independently of the oracle: from the reference's source text with Python
integers / numpy float32 (RNG, hash12, textures, slab and triangle tests), from libm in float64
(elementary functions), or from SURVEY.md / BASELINE.md's probe of the unmodified reference
(BVH shape table, per-ray traversal statistics).
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

from raytracing_c_amd import ctypes_abi as abi

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
F = np.float32


def fma32(a, b, c):
    """fmaf(a, b, c): a * b + c with ONE rounding to float32 (numeric contract v2, include/rt_math.h), exactly: the
    product and the sum in rational arithmetic, then round to nearest even."""
    from fractions import Fraction
    a, b, c = F(a), F(b), F(c)
    if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
        with np.errstate(invalid="ignore", over="ignore"):
            return F(np.float64(a) * np.float64(b) + np.float64(c))      # inf / NaN propagate like the hardware's
    exact = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    if exact == 0:
        return F(float(a) * float(b) + float(c))                          # (sign of zero as IEEE: exact sum of the two)
    g = F(float(exact))                                                    # double rounding can be off by one ulp: fix up
    best = None
    for cand in (np.nextafter(g, F(-np.inf)), g, np.nextafter(g, F(np.inf))):
        if not np.isfinite(cand):
            continue
        err = abs(Fraction(float(cand)) - exact)
        even = (int(np.array(cand, F).view(np.uint32)) & 1) == 0
        if best is None or err < best[0] or (err == best[0] and even and not best[2]):
            best = (err, cand, even)
    return F(best[1])


# --- common.h:13-24 -----------------------------------------------------------------------

def py_rand_u32(state):
    state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def test_rand_u32_sequence(oracle):
    out = np.zeros(16, np.uint32)
    oracle.oracle_rand_u32_seq(1, 16, out.ctypes.data)
    s, want = 1, []
    for _ in range(16):
        s = py_rand_u32(s)
        want.append(s)
    assert out.tolist() == want
    # first value by hand: state=1 -> 747796405+2891336453 mod 2^32 = 3639132858
    assert want[0] == py_rand_u32(1)


def test_rand_f32_is_u32_over_2_pow_32_inclusive_one(oracle):
    out = np.zeros(64, np.float32)
    oracle.oracle_rand_f32_seq(12345, 64, out.ctypes.data)
    s = 12345
    for i in range(64):
        s = py_rand_u32(s)
        assert out[i] == F(F(s) / F(4294967296.0))
    assert ((out >= 0) & (out <= 1)).all()


# --- raytracer.c:584-594 ------------------------------------------------------------------

def np_hash12(px, py):
    def fract(v):
        return F(v - np.floor(v))
    k, add = F(0.1031), F(33.33)
    p3x, p3y, p3z = fract(F(px * k)), fract(F(py * k)), fract(F(px * k))
    d = F(F(F(p3x * F(p3y + add)) + F(p3y * F(p3z + add))) + F(p3z * F(p3x + add)))
    return fract(F(F(F(p3x + p3y) + F(d * F(2.0))) * F(p3z + d)))


def test_hash12_grid(oracle):
    for x in range(4):
        for y in range(4):
            for s in range(8):
                px, py = F(F(x) * F(50.0) + F(s)), F(y)
                got = oracle.oracle_hash12(px, py)
                assert F(got) == np_hash12(px, py)
                assert 0.0 <= got < 1.0


# --- rt_math.h elementary functions vs libm (float64) --------------------------------------

def test_elementary_functions_accuracy():
    from tests import _oracle
    rng = np.random.default_rng(0)
    x = rng.uniform(0.05, 1.0, 50000).astype(F)
    assert np.max(np.abs(_oracle.math(2, x, np.full_like(x, 2.4)) / np.power(x.astype(np.float64), 2.4) - 1)) < 3e-6
    x = rng.uniform(0.003, 1.0, 50000).astype(F)
    assert np.max(np.abs(_oracle.math(2, x, np.full_like(x, F(1 / 2.4))) / np.power(x.astype(np.float64), 1 / 2.4) - 1)) < 2e-6
    a = rng.uniform(0, 2 * np.pi, 50000).astype(F)
    assert np.max(np.abs(_oracle.math(3, a) - np.sin(a.astype(np.float64)))) < 3e-7
    assert np.max(np.abs(_oracle.math(4, a) - np.cos(a.astype(np.float64)))) < 3e-7
    yy, xx = rng.uniform(-1, 1, 50000).astype(F), rng.uniform(-1, 1, 50000).astype(F)
    assert np.max(np.abs(_oracle.math(5, yy, xx) - np.arctan2(yy.astype(np.float64), xx.astype(np.float64)))) < 6e-7
    v = rng.uniform(-1, 1, 50000).astype(F)
    assert np.max(np.abs(_oracle.math(6, v) - np.arcsin(v.astype(np.float64)))) < 6e-7
    assert _oracle.math(6, np.array([1.5, -1.5], F)).tolist() == [F(np.pi / 2), F(-np.pi / 2)]   # clamp, SURVEY H6
    lg = rng.uniform(1e-4, 100, 50000).astype(F)
    assert np.max(np.abs(_oracle.math(0, lg) - np.log(lg.astype(np.float64)))) < 1e-6
    # sqrt and 1/x are the correctly rounded IEEE operations
    assert np.array_equal(_oracle.math(9, lg), np.sqrt(lg))
    assert np.array_equal(_oracle.math(10, lg), F(1.0) / lg)


def test_srgb_curves(oracle):
    from tests import _oracle
    x = np.linspace(0, 1, 257).astype(F)
    lin = _oracle.math(7, x)
    assert np.max(np.abs(lin - ((x.astype(np.float64) + 0.055) / 1.055) ** 2.4)) < 2e-6   # no toe, common.h:82-88
    back = _oracle.math(8, x)
    want = np.where(x <= 0.0031308, 12.92 * x.astype(np.float64), 1.055 * x.astype(np.float64) ** (1 / 2.4) - 0.055)
    assert np.max(np.abs(back - want)) < 2e-6
    assert oracle.oracle_encode_u8(F(0.0)) == 0 and oracle.oracle_encode_u8(F(1.0)) == 255
    assert oracle.oracle_encode_u8(F(7.5)) == 255 and oracle.oracle_encode_u8(F(-1.0)) == 0     # clamp, raytracer.c:702-706


def test_srgb_decode_power_is_within_4_ulp_for_every_float_of_its_domain(oracle):
    """Contract v3 (include/rt_math.h, rt_pow24): pow_f32((x + 0.055) / 1.055, 2.4) of common.h:84-91 through a degree-6
    polynomial in the mantissa times one of six constants.  Pinned against the power in double for EVERY float b in
    [2^-5, 2) -- the six binades a texture sample can reach -- by choosing x so that (x + 0.055f) / 1.055f visits them: here
    every x = b * 1.055f - 0.055f on a stride of 5 mantissa steps (10 M values; the C sweep of all 50 M gave 3.63 ulp worst,
    profiles/r05_pow24.md), and the decode of all 256 x 4 u8-lerp corner values exactly as a texture yields them."""
    from tests import _oracle
    bits = np.arange(np.float32(2.0 ** -5).view(np.uint32), np.float32(2.0).view(np.uint32), 5, dtype=np.uint32)
    b = bits.view(F)
    x = (b * F(1.055) - F(0.055)).astype(F)
    got = _oracle.math(7, x).astype(np.float64)
    bb = ((x + F(0.055)).astype(F) / F(1.055)).astype(F).astype(np.float64)          # the b the decode really saw
    rel = np.abs(got / bb ** 2.4 - 1.0)
    assert rel.max() < 4 * 2.0 ** -24, rel.max() / 2.0 ** -24
    u8 = (np.arange(256, dtype=F) / F(255.999)).astype(F)
    lin = _oracle.math(7, u8).astype(np.float64)
    want = ((u8.astype(np.float64) + np.float64(F(0.055))) / np.float64(F(1.055))) ** 2.4
    assert np.max(np.abs(lin / want - 1.0)) < 6 * 2.0 ** -24      # (+ 2.4 x the two roundings of (x + 0.055f) / 1.055f)
    assert _oracle.math(7, np.array([0.9999999], F))[0] <= 1.0 and _oracle.math(7, np.array([1.0], F))[0] == F(1.0)
    # outside the core domain (no texture gets there) the general power answers: 0 for b <= 0 and NaN, finite above 2
    out = _oracle.math(7, np.array([-1.0, -0.055, np.nan, 3.0, -0.03], F))
    assert out[0] == 0 and out[1] == 0 and out[2] == 0 and abs(out[3] / ((3.055 / 1.055) ** 2.4) - 1) < 3e-6 and out[4] > 0


# --- raytracer.c:190-230 ------------------------------------------------------------------

def np_slab(o, d, mn, mx, t_min, t_max):
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = F(1.0) / d
        bias = -(o * inv)
        if np.all(np.isfinite(inv)) and np.all(np.isfinite(bias)):
            # contract v2, NaN-free ray: every plane distance is ONE fused multiply-add (rt_slab_t_fast)
            t0 = np.array([fma32(mn[i], inv[i], bias[i]) for i in range(3)], F)
            t1 = np.array([fma32(mx[i], inv[i], bias[i]) for i in range(3)], F)
        else:
            # the reference's own form (raytracer.c:203-208), also what contract v1 computes for every ray
            t0 = (mn - o) * inv
            t1 = (mx - o) * inv

    def mn2(a, b):
        return a if a < b else b

    def mx2(a, b):
        return a if a > b else b
    s = [mn2(t0[i], t1[i]) for i in range(3)]
    b = [mx2(t0[i], t1[i]) for i in range(3)]
    tmin = mx2(F(t_min), mx2(s[0], mx2(s[1], s[2])))
    tmax = mn2(F(t_max), mn2(b[0], mn2(b[1], b[2])))
    return F(np.inf) if tmin >= tmax else tmin


def test_ray_aabbs_hit_8_cases(oracle):
    node = abi.BVH_Node()
    boxes = [((-1, -1, -1), (1, 1, 1)), ((2, -1, -1), (3, 1, 1)), ((-1, 5, -1), (1, 6, 1)),
             ((0, 0, 0), (0, 0, 0)),              # unpopulated child: always a miss
             ((-1, -1, 4), (1, 1, 6)), ((-10, -10, -10), (10, 10, 10)), ((-1, -1, -9), (1, 1, -7)),
             ((0.5, 0.5, -1), (1.5, 1.5, 1))]
    for k, (mn, mx) in enumerate(boxes):
        node.min_x[k], node.min_y[k], node.min_z[k] = mn
        node.max_x[k], node.max_y[k], node.max_z[k] = mx
    rays = [((0, 0, -5), (0, 0, 1)), ((-5, 0, 0), (1, 0, 0)), ((0, 0, 0), (0, 1, 0)),
            ((0.25, 0.3, -5), (0.1, 0.05, 0.99)), ((0, 0, 0), (0.57735026, 0.57735026, 0.57735026)),
            ((0.37, -0.21, -4.9), (0.123, 0.077, 0.989)), ((7.3, 3.1, -2.2), (-0.81, -0.33, 0.48)),
            ((1e30, 0, 0), (-1e-9, 0.3, 0.95))]          # bias overflows: this ray keeps the unfused form
    for (o, d) in rays:
        for t_max in (np.inf, 4.5):
            ray = abi.Ray(abi.Vec3(*o), abi.Vec3(*d))
            out = np.zeros(8, F)
            oracle.oracle_ray_aabbs_hit_8(C.byref(ray), F(1e-4), F(t_max), C.byref(node), out.ctypes.data)
            for k, (mn, mx) in enumerate(boxes):
                want = np_slab(np.array(o, F), np.array(d, F), np.array(mn, F), np.array(mx, F), 1e-4, t_max)
                assert out[k] == want or (np.isnan(out[k]) and np.isnan(want)), (o, d, k)
            assert not (out[3] < np.inf), "zero box: +inf, or NaN for a ray through the origin; never '<'"


# --- raytracer.c:84-188 -------------------------------------------------------------------

def _leaf(tris8):
    """Triangles struct with one leaf group from 8 (a, b, c) vertex triples (or None = zero slot)."""
    data = np.zeros((9, 8), F)
    aos = (abi.Triangle_AOS * 8)()
    for k, t in enumerate(tris8):
        if t is None:
            continue
        for v in range(3):
            data[0 + v, k], data[3 + v, k], data[6 + v, k] = t[v]
        aos[k].normal_a = abi.Vec3(1, 0, 0)
        aos[k].normal_b = abi.Vec3(0, 1, 0)
        aos[k].normal_c = abi.Vec3(0, 0, 1)
        aos[k].tex_coords_b = abi.Vec2(1, 0)
        aos[k].tex_coords_c = abi.Vec2(0, 1)
        aos[k].normal = abi.Vec3(0, 0, float(k))
    T = abi.Triangles()
    fp = C.POINTER(C.c_float)
    for v in range(3):
        T.x[v] = data[0 + v].ctypes.data_as(fp)
        T.y[v] = data[3 + v].ctypes.data_as(fp)
        T.z[v] = data[6 + v].ctypes.data_as(fp)
    T.aos = C.cast(aos, C.POINTER(abi.Triangle_AOS))
    T.len = 8
    return T, (data, aos)


def _hit(distance=np.inf):
    h = abi.Hit()
    h.distance = distance
    return h


def test_ray_triangles_hit_8_cases(oracle):
    tri = lambda z: ((0, 0, z), (1, 0, z), (0, 1, z))           # noqa: E731
    T, keep = _leaf([tri(5), tri(3), None, tri(3), tri(-1), None, tri(7), tri(4)])
    lane = C.c_int32(-1)

    # closest of several; equal t on lanes 1 and 3 -> LOWEST lane wins (min_f32x8, raytracer.c:27-29)
    ray = abi.Ray(abi.Vec3(0.25, 0.25, 0), abi.Vec3(0, 0, 1))
    h = _hit()
    assert oracle.oracle_ray_triangles_hit_8(C.byref(ray), C.byref(T), 0, C.byref(h), C.byref(lane))
    assert lane.value == 1 and h.distance == F(3.0)
    assert (h.point.x, h.point.y, h.point.z) == (F(0.25), F(0.25), F(3.0))
    # barycentric interpolation t0 = 1-u-v on a, t1 = u on b, t2 = v on c (raytracer.c:164-177)
    assert (h.normal.x, h.normal.y, h.normal.z) == (F(0.5), F(0.25), F(0.25))
    assert (h.tex_coords.x, h.tex_coords.y) == (F(0.25), F(0.25))
    assert h.normal_geo.z == 1.0

    # a hit that is not closer than hit.distance is rejected (strict <, raytracer.c:159)
    h = _hit(3.0)
    assert not oracle.oracle_ray_triangles_hit_8(C.byref(ray), C.byref(T), 0, C.byref(h), C.byref(lane))
    assert h.distance == 3.0

    # triangle behind the origin (t < eps) and zero triangles (NaN path) never hit
    ray2 = abi.Ray(abi.Vec3(0.25, 0.25, 7.5), abi.Vec3(0, 0, 1))
    h = _hit()
    assert not oracle.oracle_ray_triangles_hit_8(C.byref(ray2), C.byref(T), 0, C.byref(h), C.byref(lane))
    assert h.distance == np.inf

    # epsilon-padded barycentric bounds: u = -0.5e-4 accepted, u = -2e-4 rejected (raytracer.c:137-147)
    for x, accept in ((-0.5e-4, True), (-2e-4, False), (1.00005 - 0.25, True)):
        r = abi.Ray(abi.Vec3(x, 0.25, 0), abi.Vec3(0, 0, 1))
        h = _hit()
        got = oracle.oracle_ray_triangles_hit_8(C.byref(r), C.byref(T), 0, C.byref(h), C.byref(lane))
        assert bool(got) == accept, x


# --- driver.c:49-93 -----------------------------------------------------------------------

def np_bilinear(pix, tx, ty):
    h, w, comp = pix.shape
    tx, ty = F(tx), F(ty)
    if tx < 0:
        tx = F(tx + F(-int(tx) + 1))
    if ty < 0:
        ty = F(ty + F(-int(ty) + 1))
    tx, ty = F(tx - np.floor(tx)), F(ty - np.floor(ty))
    px, py = F(tx * F(w)), F(ty * F(h))
    u, v = int(px), int(py)
    a, b = F(px - F(u)), F(py - F(v))
    u2 = u + 1 if u + 1 < w else u
    v2 = v + 1 if v + 1 < h else v

    def tex(uu, vv):
        return (pix[vv, uu, :3].astype(F) / F(255.999)).astype(F)

    def lerp(p, q, t):              # rt_lerpf: fma(q, t, p * (1 - t))
        return np.array([fma32(q[i], t, F(p[i] * F(F(1.0) - t))) for i in range(3)], F)
    c0 = lerp(tex(u, v), tex(u2, v), a)
    c1 = lerp(tex(u, v2), tex(u2, v2), a)
    return lerp(c0, c1, b)


def test_sample_texture_bilinear(oracle):
    from raytracing_c_amd.scene import make_image
    rng = np.random.default_rng(5)
    for shape in ((2, 2, 3), (5, 3, 4), (16, 16, 3)):
        pix = rng.integers(0, 256, shape, dtype=np.uint8)
        img, keep = make_image(pix)
        uvs = [(0, 0), (0.5, 0.5), (0.999, 0.999), (-0.25, 0.75), (-1.0, -2.0), (3.25, -0.125), (1.0, 1.0)]
        uvs += [tuple(p) for p in rng.uniform(-2, 2, (200, 2))]
        for (tu, tv) in uvs:
            out = np.zeros(3, F)
            oracle.oracle_sample_texture_bilinear(C.byref(img), F(tu), F(tv), out.ctypes.data)
            assert np.array_equal(out, np_bilinear(keep, tu, tv)), (shape, tu, tv)


def test_texel_scale_multiplication_is_exact_on_u8():
    """The device computes u8 * RN(1/255.999f); the reference divides (driver.c:70-87).  Equal for all 256 inputs."""
    i = np.arange(256).astype(F)
    assert np.array_equal(i / F(255.999), i * (F(1.0) / F(255.999)))


def test_sample_background_orientation(oracle):
    """u = 0.5 + atan2(z, x)/2pi, v = 0.5 - asin(y)/pi (driver.c:95-104): +y looks at row 0."""
    from raytracing_c_amd.scene import make_image
    pix = np.zeros((8, 16, 3), np.uint8)
    pix[0, :, 0] = 255       # top row red
    pix[-1, :, 2] = 255      # bottom row blue
    img, keep = make_image(pix)
    out = np.zeros(3, F)
    oracle.oracle_sample_background(C.byref(img), np.array([0, 1, 0], F).ctypes.data, out.ctypes.data)
    assert out[0] > 0.9 and out[2] < 0.01
    # straight down gives v = 1.0, which fract() wraps to row 0 (reference behaviour); just off-axis hits the last row
    oracle.oracle_sample_background(C.byref(img), np.array([0.02, -0.9998, 0], F).ctypes.data, out.ctypes.data)
    assert out[2] > 0.9 and out[0] < 0.01


# --- driver.c:287-348 ---------------------------------------------------------------------

def test_sample_disney_brdf_lobes_and_rng_draws(oracle):
    """RNG draws per shaded bounce: 3 (specular) or 5 (diffuse) -- SURVEY.md section 3.5."""
    base = np.array([0.8, 0.4, 0.2], F)
    in_dir = np.array([0.3, 0.2, 0.93273791], F)
    n_diff = n_spec = 0
    for seed in range(200):
        for metal in (0.0, 0.5, 1.0):
            st = C.c_uint32(seed * 7919 + 1)
            out_dir, brdf = np.zeros(3, F), np.zeros(4, F)
            oracle.oracle_sample_disney_brdf(F(0.4), F(metal), F(0.0), F(0.0), F(0.0), base.ctypes.data,
                                             in_dir.ctypes.data, C.byref(st), out_dir.ctypes.data, brdf.ctypes.data)
            s, draws = seed * 7919 + 1, 0
            while s != st.value:
                s = py_rand_u32(s)
                draws += 1
                assert draws <= 5
            assert draws in (3, 5)
            if draws == 5:
                n_diff += 1
                assert metal < 1.0
            else:
                n_spec += 1
            if brdf[3] > 0:
                assert abs(np.linalg.norm(out_dir) - 1) < 1e-5 and out_dir[2] > 0
                assert np.all(brdf[:3] >= 0)
    assert n_diff > 100 and n_spec > 100


# --- BVH shape: scene.c:224-242,311-414 against the survey's probe of the reference -----------

BVH_SHAPES = [   # asset, triangles, depth, node slots, populated nodes, triangle slots, populated leaves
    ("helmet.glb", 15452, 4, 585, 278, 32768, 1932),
    ("tower.obj", 4320, 4, 585, 80, 32768, 540),
    ("spheres.glb", 4800, 4, 585, 88, 32768, 600),
    ("sheen.glb", 1920, 3, 73, None, 4096, 240),
    ("quad.obj", 2, 0, 0, 0, 8, 1),
    ("fov_test.obj", 72, 2, 9, None, 512, 9),
]


@pytest.mark.parametrize("asset,ntri,depth,slots,populated,tslots,leaves", BVH_SHAPES, ids=[b[0] for b in BVH_SHAPES])
def test_bvh_shape_table(asset, ntri, depth, slots, populated, tslots, leaves):
    from raytracing_c_amd.loaders import load_model
    hs = load_model(os.path.join(ASSETS, asset))
    assert hs.n_input_triangles == ntri
    assert hs.depth == depth
    assert hs.n_nodes == slots
    assert hs.n_slots == tslots
    assert hs.populated_leaves() == leaves
    if populated is not None:
        assert hs.populated_nodes() == populated


@pytest.mark.parametrize("asset", ["spheres.glb", "fov_test.obj", "sheen.glb", "quad.obj"])
def test_bvh_invariants(asset):
    """Every input triangle is stored exactly once; every child box contains its subtree (eps padded)."""
    from raytracing_c_amd.loaders import load_model, load_model_data
    hs = load_model(os.path.join(ASSETS, asset))
    d = load_model_data(os.path.join(ASSETS, asset))
    soa = hs.soa_array()                                         # (9, slots)
    stored = np.stack([soa[0], soa[3], soa[6], soa[1], soa[4], soa[7], soa[2], soa[5], soa[8]], 1)   # v0 v1 v2 xyz
    used = np.any(stored != 0, axis=1)
    assert used.sum() == hs.n_input_triangles
    a = np.sort(stored[used].view([("", F)] * 9).ravel())
    b = np.sort(np.ascontiguousarray(d["positions"].reshape(-1, 9)).view([("", F)] * 9).ravel())
    assert np.array_equal(a, b)
    if hs.depth == 0:
        return
    nodes = hs.nodes_array()                                     # (n, 6, 8)
    depth, last = hs.depth, int(hs.scene.bvh.last_row_offset)

    def subtree_bounds(index, level):
        """(lo, hi) over all triangles below node/leaf `index`; None if empty."""
        if level == depth:                                       # leaf group
            g = index - last
            sl = stored[g * 8:(g + 1) * 8]
            m = np.any(sl != 0, axis=1)
            if not m.any():
                return None
            p = sl[m].reshape(-1, 3)
            return p.min(0), p.max(0)
        lo = hi = None
        for j in range(8):
            r = subtree_bounds(8 * index + 1 + j, level + 1)
            box_lo, box_hi = nodes[index, 0:3, j], nodes[index, 3:6, j]
            if r is None:
                assert not box_lo.any() and not box_hi.any(), "unpopulated child must be the zero box"
                continue
            assert np.all(box_lo <= r[0] - F(0.9e-4)) and np.all(box_hi >= r[1] + F(0.9e-4))
            assert np.all(box_lo >= r[0] - F(1.1e-4)) and np.all(box_hi <= r[1] + F(1.1e-4)), "box is tight + eps"
            lo = r[0] if lo is None else np.minimum(lo, r[0])
            hi = r[1] if hi is None else np.maximum(hi, r[1])
        return None if lo is None else (lo, hi)

    assert subtree_bounds(0, 0) is not None


def test_traversal_equals_brute_force(oracle):
    """ray_bvh_node_hit must return the same closest distance as the reference's disabled brute-force
    path (raytracer.c:491-499: every leaf group in order)."""
    from raytracing_c_amd.loaders import load_model
    from tests.test_gpu_parity import _rays_for
    hs = load_model(os.path.join(ASSETS, "spheres.glb"))
    rng = np.random.default_rng(11)
    n = 300
    rays = _rays_for(hs, n, rng)
    wt, wtri, wuv = np.zeros(n, F), np.zeros(n, np.int32), np.zeros((n, 2), F)
    oracle.oracle_trace_rays(C.byref(hs.scene), n, rays.ctypes.data, wt.ctypes.data, wtri.ctypes.data, wuv.ctypes.data)
    groups = np.nonzero(np.any(hs.soa_array().reshape(9, -1, 8) != 0, axis=(0, 2)))[0]
    lane = C.c_int32(0)
    for i in range(n):
        ray = abi.Ray(abi.Vec3(*rays[i, :3]), abi.Vec3(*rays[i, 3:]))
        h = _hit()
        for g in groups:
            oracle.oracle_ray_triangles_hit_8(C.byref(ray), C.byref(hs.scene.triangles), int(g) * 8, C.byref(h), C.byref(lane))
        assert F(h.distance) == wt[i], i
    assert (wtri >= 0).sum() > 30


# --- whole-path statistics against BASELINE.md's probe of the unmodified reference -----------

def test_path_statistics_match_reference_probe():
    """BASELINE.md section 2, config #1 (spheres 256x256, 16 spp, 4 bounces), measured on the unmodified
    reference: 1.275 rays/path, 3.561 node visits/ray, 1.129 leaf visits/ray, 0.221 shades/ray.  The
    oracle uses a different RNG seeding rule and background, so agreement is statistical (1 %)."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, cfg = load_config("spheres")
    c = _oracle.render(hs, cfg["width"], cfg["height"], cfg["samples"], cfg["max_bounces"])["counters"]
    assert c["paths"] == 256 * 256 * 16
    assert abs(c["rays"] / c["paths"] - 1.275) < 0.013
    assert abs(c["node_visits"] / c["rays"] - 3.561) < 0.036
    assert abs(c["leaf_visits"] / c["rays"] - 1.129) < 0.012
    assert abs(c["shades"] / c["rays"] - 0.221) < 0.003


def test_fixed_point_accumulation_matches_fp32_running_sum():
    """Deviation D6: 32.32 fixed-point sums vs the reference's fp32 `color += cast_ray` (raytracer.c:695-700)."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    a = _oracle.render(hs, 64, 64, 16, 4, accum_mode=0)
    b = _oracle.render(hs, 64, 64, 16, 4, accum_mode=1)
    err = np.abs(a["linear"].astype(np.float64) - b["linear"]) / np.maximum(1.0, np.abs(b["linear"]))
    assert err.max() < 2e-6
    assert np.sqrt(np.mean((a["linear"].astype(np.float64) - b["linear"]) ** 2)) < 1e-6     # << 1e-4 RMS
    # u8: the sRGB round trip of a flat background texel lands on an integer boundary of `* 255.999`
    # (raytracer.c:707-716 truncates), so last-ulp differences flip it by one code -- never more
    assert np.abs(a["image"].astype(int) - b["image"].astype(int)).max() <= 1
    # resolve(accum) is exactly the stored linear value
    mean = (a["accum"].astype(np.float64) / (16 * 4294967296.0)).astype(F)
    assert np.array_equal(mean, a["linear"])


def test_oracle_is_thread_count_invariant():
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    a = _oracle.render(hs, 48, 40, 4, 4, n_threads=1)
    b = _oracle.render(hs, 48, 40, 4, 4, n_threads=7)
    assert np.array_equal(a["accum"], b["accum"]) and a["counters"] == b["counters"]


def test_helmet_matches_reference_sample_render_in_framing_and_orientation():
    """Visual pin: the reference's own sample render output.png (helmet, 1024x1024, its own
    environment map) kept as a 128x128 box-filtered fixture.  Lighting differs (our background is procedural), so
    only view-independent landmarks are compared: the emissive cyan ring of the visor display (UV orientation +
    camera convention) and the dark visor glass must sit where the reference has them."""
    from PIL import Image
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_output_png_128.npz"))["image"]
    hs, _ = load_config("helmet")
    img = _oracle.render(hs, 256, 256, 16, 8)["image"]
    mine = np.asarray(Image.fromarray(img).resize((128, 128), Image.BOX))

    def cyan(a):
        r, g, b = [a[..., i].astype(int) for i in range(3)]
        return (g > 120) & (b > 120) & (r < g - 40)

    def centroid(m):
        ys, xs = np.nonzero(m)
        return np.array([xs.mean(), ys.mean()])

    cr, cm = cyan(ref), cyan(mine)
    assert cr.sum() > 40 and cm.sum() > 40
    assert np.abs(centroid(cr) - centroid(cm)).max() < 7.0            # 128-pixel scale: within 5 % of the frame
    dr, dm = ref.astype(int).sum(-1) < 120, mine.astype(int).sum(-1) < 120
    assert np.abs(centroid(dr) - centroid(dm)).max() < 8.0
    # ... to the pixel: left and bottom edge of the ring (its right/top side merges with HUD lines that our brighter
    # environment lights up), and the three small yellow lamps of the upper shell, which are emissive
    def ring_box(a):
        m = cyan(a)
        m[:, 60:] = False
        ys, xs = np.nonzero(m)
        return xs.min(), ys.max()
    assert np.abs(np.array(ring_box(ref)) - np.array(ring_box(mine))).max() <= 1

    def lamps(a):
        r, g, b = [a[..., i].astype(int) for i in range(3)]
        m = (r > 150) & (g > 140) & (b < 110)
        m[64:] = False
        return np.argwhere(m)
    lr, lm = lamps(ref), lamps(mine)
    assert len(lm) >= 3
    matched = sum(1 for q in lm if np.abs(lr - q).max(axis=1).min() <= 1)
    assert matched >= 3, (lr.tolist(), lm.tolist())                    # (y, x) at 128 x 128: (34, 101), (34, 102), (35, 37)
    # a vertical or horizontal flip of our render would move the ring far away
    assert np.abs(centroid(cr) - centroid(cyan(mine[::-1]))).max() > 15 and np.abs(centroid(cr) - centroid(cyan(mine[:, ::-1]))).max() > 15

"""The checker's two forms of the reference's SIMD routines: the 8-wide AVX2 forms of min_f32x8 /
ray_triangles_hit_8 / ray_aabbs_hit_8 (raytracer.c:15-32,84-230; _mm256_min_ps / _mm256_max_ps / _mm256_cmp_ps /
_mm256_blendv_ps in the reference's operand order, _mm256_fmadd_ps where numeric contract v2 has an rt_madd) against the
scalar restatement: the same bits, ray by ray and frame by frame, under both numeric contracts."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracing_c_amd import ctypes_abi as abi

F = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
LIBS = ["liboracle.so", "liboracle_v1.so"]


@pytest.fixture(scope="module", params=LIBS)
def lib(request):
    from tests import _oracle
    _oracle.load()                                          # (builds both if missing)
    d = _oracle.load(os.path.join(ROOT, "oracle", request.param))
    if not d.oracle_have_avx2():
        pytest.skip("checker built without AVX2")
    yield d
    d.oracle_set_simd(1)


def _rays(hs, n, rng):
    soa = hs.soa_array()
    used = np.any(soa != 0, axis=0)
    pts = np.stack([soa[0][used], soa[3][used], soa[6][used]], 1)
    lo, hi = pts.min(0), pts.max(0)
    c, ext = (lo + hi) / 2, max(float((hi - lo).max()), 1e-3)
    o = c + rng.normal(size=(n, 3)) * ext * 1.5
    tgt = pts[rng.integers(0, len(pts), n)] + rng.normal(size=(n, 3)) * ext * 0.02
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], 1).astype(F)
    k = n // 16
    rays[:k, 3:] = 0
    rays[:k, 3 + np.arange(k) % 3] = np.where(np.arange(k) % 2, 1.0, -1.0)          # axis-aligned: 0 * inf in the slab test
    rays[k:2 * k, 3 + np.arange(k) % 3] = 0.0                                        # one zero component
    rays[2 * k:2 * k + 8, rng.integers(0, 6, 8)] = np.nan
    rays[2 * k + 8:2 * k + 16, 3] = np.inf
    rays[2 * k + 16:2 * k + 24, 0] = 1e30                                            # slab bias overflows: the unfused form
    return np.ascontiguousarray(rays)


def _trace(lib, hs, rays):
    n = len(rays)
    t, tri, uv = np.zeros(n, F), np.zeros(n, np.int32), np.zeros((n, 2), F)
    visits = (C.c_uint64 * 2)()
    lib.oracle_trace_rays_counted(C.byref(hs.scene), n, rays.ctypes.data, t.ctypes.data, tri.ctypes.data, uv.ctypes.data, visits)
    return t.view(np.uint32), tri, uv.view(np.uint32), (int(visits[0]), int(visits[1]))


@pytest.mark.parametrize("asset", ["quad.obj", "fov_test.obj", "sheen.glb", "spheres.glb", "tower.obj", "helmet.glb"])
def test_avx2_traversal_equals_the_scalar_one(lib, asset):
    from raytracing_c_amd.loaders import load_model
    hs = load_model(os.path.join(ASSETS, asset))
    rays = _rays(hs, 20000, np.random.default_rng(17))
    assert lib.oracle_set_simd(1) == 1
    a = _trace(lib, hs, rays)
    assert lib.oracle_set_simd(0) == 0
    b = _trace(lib, hs, rays)
    assert (a[1] >= 0).sum() > 1000
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


def test_avx2_box_and_triangle_tests_equal_the_scalar_ones_on_random_and_special_operands(lib):
    rng = np.random.default_rng(5)
    node = abi.BVH_Node()
    specials = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38, 3e38, 1.0]
    for it in range(400):
        lo = rng.normal(size=(3, 8)).astype(F)
        hi = lo + np.abs(rng.normal(size=(3, 8))).astype(F)
        if it % 5 == 0:
            lo[:, it % 8] = hi[:, it % 8] = 0                                   # an unpopulated child
        if it % 7 == 0:
            lo[rng.integers(0, 3), rng.integers(0, 8)] = specials[it % len(specials)]
        for k in range(8):
            node.min_x[k], node.min_y[k], node.min_z[k] = (float(v) for v in lo[:, k])
            node.max_x[k], node.max_y[k], node.max_z[k] = (float(v) for v in hi[:, k])
        o = rng.normal(size=3) * 3
        d = rng.normal(size=3)
        if it % 3 == 0:
            d[it % 3] = specials[(it // 3) % len(specials)]
        if it % 11 == 0:
            o[it % 3] = specials[(it // 11) % len(specials)]
        ray = abi.Ray(abi.Vec3(*[float(F(v)) for v in o]), abi.Vec3(*[float(F(v)) for v in d]))
        out = [np.zeros(8, F), np.zeros(8, F)]
        for mode in (1, 0):
            lib.oracle_set_simd(mode)
            lib.oracle_ray_aabbs_hit_8(C.byref(ray), F(1e-4), F(np.inf if it % 2 else 2.5), C.byref(node), out[mode].ctypes.data)
        assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32)), (it, out)


@pytest.mark.parametrize("name,w,h,s,b", [("helmet", 96, 54, 6, 8), ("spheres", 64, 64, 8, 4), ("quad", 48, 48, 8, 4)])
def test_avx2_frames_equal_the_scalar_ones(lib, name, w, h, s, b):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config(name)
    lib.oracle_set_simd(1)
    a = _oracle.render(hs, w, h, s, b, n_threads=4, lib=lib)
    lib.oracle_set_simd(0)
    c = _oracle.render(hs, w, h, s, b, n_threads=4, lib=lib)
    assert np.array_equal(a["accum"], c["accum"]) and np.array_equal(a["image"], c["image"]) and a["counters"] == c["counters"]

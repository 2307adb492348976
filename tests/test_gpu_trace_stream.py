"""Unit-level parity of the PRODUCTION traversal (reference raytracer.c:443-483): rt_test_trace_stream() sends arbitrary
rays through traversal_blocks() -- the one function that holds the NODE / LEAF / pop code of the tile-stream path kernel,
of the wavefront kernels and of this test kernel (csrc/rt_dev.hip.h) -- in the path kernel's launch geometry, and every ray's
(t, triangle, u, v) and the TOTAL node / leaf visits must equal oracle_trace_rays_counted() bit for bit.  Covered forms:
the LDS node block with planes picked by address (NODE_LDS_ORDERED), the min / max form for rays that are not NaN-free,
nodes through L1 / L2, both reciprocals of the leaf block, the address-picked pop re-test, and -- with a pyramid -- the
culled node block node_enter_few() with 1-4 surviving children."""
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


def _bbox(hs):
    T = hs.scene.triangles
    n = int(T.len)
    pts = np.concatenate([np.ctypeslib.as_array(T.x[k], (n,)) for k in range(3)]), \
        np.concatenate([np.ctypeslib.as_array(T.y[k], (n,)) for k in range(3)]), \
        np.concatenate([np.ctypeslib.as_array(T.z[k], (n,)) for k in range(3)])
    lo = np.array([p.min() for p in pts], np.float64)
    hi = np.array([p.max() for p in pts], np.float64)
    return lo, hi


def _rays(hs, n, rng, special=True):
    """Rays from outside and inside the scene box, plus the awkward ones: axis-aligned (0 * inf in the slab test), starting on a
    box face, zero components, NaN, huge / tiny magnitudes."""
    lo, hi = _bbox(hs)
    c, e = (lo + hi) / 2, np.maximum(hi - lo, 1e-3)
    rays = np.zeros((n, 6), np.float32)
    k = n // 2
    o = c + rng.normal(size=(k, 3)) * e * 1.5
    t = c + rng.uniform(-0.5, 0.5, (k, 3)) * e
    d = t - o
    rays[:k, :3] = o
    rays[:k, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    o = c + rng.uniform(-0.5, 0.5, (n - k, 3)) * e                       # inside: incoherent, what bounce rays look like
    d = rng.normal(size=(n - k, 3))
    rays[k:, :3] = o
    rays[k:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    if special:
        m = min(600, n // 8)
        idx = rng.choice(n, m, replace=False)
        for j, i in enumerate(idx):
            kind = j % 6
            if kind == 0:                                                  # axis-aligned
                ax = j % 3
                rays[i, 3:] = 0
                rays[i, 3 + ax] = 1.0 if (j // 3) % 2 else -1.0
            elif kind == 1:                                                # one zero component
                rays[i, 3 + j % 3] = 0.0
            elif kind == 2:                                                # origin exactly on a face of the scene box
                rays[i, j % 3] = np.float32(lo[j % 3] if (j // 3) % 2 else hi[j % 3])
            elif kind == 3:                                                # NaN somewhere
                rays[i, rng.integers(0, 6)] = np.nan
            elif kind == 4:                                                # unnormalised, 2^+-30
                rays[i, 3:] *= np.float32(2.0 ** (30 if (j // 6) % 2 else -30))
            else:                                                          # negative zero / infinity in the direction
                rays[i, 3 + j % 3] = -0.0 if (j // 6) % 2 else np.inf
    return np.ascontiguousarray(rays)


def _oracle_trace(oracle, hs, rays):
    n = len(rays)
    t, tri, uv = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
    visits = (C.c_uint64 * 2)()
    oracle.oracle_trace_rays_counted(C.byref(hs.scene), n, rays.ctypes.data, t.ctypes.data, tri.ctypes.data, uv.ctypes.data, visits)
    return t, tri, uv, (int(visits[0]), int(visits[1]))


def _gpu_trace(rt, d, rays, pyramid=None, exit_lanes=48, mode=0):
    n = len(rays)
    t, tri, uv = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
    visits = (C.c_uint64 * 2)()
    pyr = None if pyramid is None else np.ascontiguousarray(pyramid, np.float32)
    rc = rt.diag.rt_test_trace_stream(d, n, rays.ctypes.data, None if pyr is None else pyr.ctypes.data, exit_lanes, mode,
                                     t.ctypes.data, tri.ctypes.data, uv.ctypes.data, visits)
    assert rc == 0, rt.last_error(rt.diag)
    return t, tri, uv, (int(visits[0]), int(visits[1]))


def _same(want, got):
    wt, wtri, wuv, wv = want
    gt, gtri, guv, gv = got
    assert np.array_equal(wtri, gtri)
    assert np.array_equal(_bits(wt), _bits(gt))
    assert np.array_equal(_bits(wuv), _bits(guv))
    assert wv == gv, (wv, gv)


@pytest.mark.parametrize("asset", ["quad.obj", "fov_test.obj", "sheen.glb", "spheres.glb", "tower.obj", "helmet.glb"])
def test_production_traversal_is_bit_exact(rt, oracle, asset, diag):
    from raytracing_c_amd.loaders import load_model
    hs = load_model(os.path.join(ASSETS, asset))
    rng = np.random.default_rng(11)
    rays = _rays(hs, 30000, rng)
    want = _oracle_trace(oracle, hs, rays)
    assert (want[1] >= 0).sum() > len(rays) // 10, "test rays must actually hit the scene"
    d = rt.diag.rt_scene_upload(C.byref(hs.scene))
    assert d, rt.last_error()
    try:
        for mode in (0, 1, 2):                      # short reciprocal + LDS nodes | IEEE division | nodes through L1 / L2
            for exit_lanes in (48, 1, 64):
                _same(want, _gpu_trace(rt, d, rays, None, exit_lanes, mode))
    finally:
        rt.diag.rt_scene_release(d)


def test_sah_scene_and_inverted_boxes(rt, oracle, diag):
    """The opt-in SAH builder's tree, and a tree whose boxes are not min <= max (the LDS node blocks must not be used)."""
    from raytracing_c_amd.configs import load_config
    from tests.test_gpu_random_scenes import make_scene
    rng = np.random.default_rng(5)
    hs, _ = load_config("helmet", builder="sah")
    rays = _rays(hs, 20000, rng)
    d = rt.diag.rt_scene_upload(C.byref(hs.scene))
    try:
        _same(_oracle_trace(oracle, hs, rays), _gpu_trace(rt, d, rays))
    finally:
        rt.diag.rt_scene_release(d)
    hs = make_scene(3, 700)
    nodes = np.ctypeslib.as_array(C.cast(hs.scene.bvh.nodes.data, C.POINTER(C.c_float)), (int(hs.scene.bvh.nodes.len) * 48,))
    nb = nodes.reshape(-1, 2, 24)
    nb[1::3] = nb[1::3, ::-1].copy()               # swap mins and maxs of every third node: inverted boxes
    rays = _rays(hs, 20000, rng)
    d = rt.diag.rt_scene_upload(C.byref(hs.scene))
    try:
        _same(_oracle_trace(oracle, hs, rays), _gpu_trace(rt, d, rays))
    finally:
        rt.diag.rt_scene_release(d)


def _pyramid(origin, dirs, margin=0.02):
    """Four outward side planes of a pyramid around a bundle of directions from `origin`: an orthonormal frame around the mean
    direction, the extreme tangents of the bundle widened by `margin` (relative), planes through the origin."""
    origin = np.asarray(origin, np.float64)
    d = dirs.astype(np.float64)
    axis = d.mean(axis=0)
    axis /= np.linalg.norm(axis)
    up = np.array([0.0, 1.0, 0.0]) if abs(axis[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
    u = np.cross(axis, up)
    u /= np.linalg.norm(u)
    v = np.cross(axis, u)
    along = d @ axis
    assert (along > 0).all()
    tu, tv = (d @ u) / along, (d @ v) / along
    spread = max(tu.max() - tu.min(), tv.max() - tv.min(), 1e-4)
    u0, u1, v0, v1 = tu.min() - margin * spread, tu.max() + margin * spread, tv.min() - margin * spread, tv.max() + margin * spread
    corners = [axis + a * u + b * v for a, b in ((u0, v0), (u1, v0), (u1, v1), (u0, v1))]
    pyr = np.zeros(19, np.float32)
    for q in range(4):
        nrm = np.cross(corners[q], corners[(q + 1) % 4])
        if nrm @ corners[(q + 2) % 4] > 0:
            nrm = -nrm                              # outward: the opposite corner is inside
        pyr[4 * q:4 * q + 3] = nrm
        assert (d @ nrm <= 1e-9 * np.linalg.norm(nrm)).all()
    pyr[16:19] = origin
    return pyr


@pytest.mark.parametrize("asset", ["helmet.glb", "spheres.glb", "tower.obj"])
def test_pyramid_culled_node_blocks(rt, oracle, asset, diag):
    """Bundles of rays from one origin inside a narrow pyramid -- what the camera rays of an 8x8 tile are -- take the culled
    node block (pyramid_cull_mask + node_enter_few, 1 to 4 surviving children) and must still visit exactly the oracle's nodes and
    leaves.  Bundles of several widths from far, near and inside the model, so that every survivor count occurs."""
    from raytracing_c_amd.loaders import load_model
    hs = load_model(os.path.join(ASSETS, asset))
    lo, hi = _bbox(hs)
    c, e = (lo + hi) / 2, np.maximum(hi - lo, 1e-3)
    rng = np.random.default_rng(23)
    d = rt.diag.rt_scene_upload(C.byref(hs.scene))
    assert d, rt.last_error()
    try:
        hits = 0
        for case in range(24):
            dist = (3.0, 1.2, 0.6, 0.2)[case % 4]
            origin = (c + rng.normal(size=3) / np.sqrt(3) * e * dist * 2).astype(np.float32)
            target = c + rng.uniform(-0.4, 0.4, 3) * e
            axis = target - origin.astype(np.float64)
            axis /= np.linalg.norm(axis)
            width = (0.002, 0.01, 0.04, 0.15)[(case // 4) % 4]
            n = 4096
            dirs = axis + rng.uniform(-width, width, (n, 3))
            dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
            rays = np.ascontiguousarray(np.concatenate([np.tile(origin, (n, 1)), dirs], axis=1), np.float32)
            pyr = _pyramid(origin, dirs)
            want = _oracle_trace(oracle, hs, rays)
            hits += int((want[1] >= 0).sum())
            _same(want, _gpu_trace(rt, d, rays, pyr, 48, 0))
            _same(want, _gpu_trace(rt, d, rays, pyr, 1, 1))
        assert hits > 10000
    finally:
        rt.diag.rt_scene_release(d)

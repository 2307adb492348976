"""Randomised scenes: rare data-dependent branches need their own inputs.

Procedural triangle soups with duplicated triangles (exactly equal hit distances -> the reference's first-found
tie rule, raytracer.c:159,464), degenerate and axis-aligned geometry, UVs outside [0,1], and materials that
switch on every optional term of driver.c:350-409 (sheen, anisotropy, emission, metalness beyond the 0.9 clamp,
tiny roughness, normal / albedo / metal-roughness / emission textures).  GPU == oracle, bit for bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_scene(seed, n_tris, depth_hint=None, builder="reference", scale=1.0):
    from raytracing_c_amd.background import procedural_background
    from raytracing_c_amd.loaders import camera_from_trs
    from raytracing_c_amd.scene import Material, build_scene
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n_tris, 1, 3))
    P = (c + rng.normal(size=(n_tris, 3, 3)) * rng.choice([0.05, 0.3, 0.8], (n_tris, 1, 1))).astype(np.float32)
    # exact duplicates (ties), a degenerate triangle, axis-aligned quads on the planes x=0 / y=0
    k = max(2, n_tris // 10)
    P[-k:] = P[:k]
    P[k] = P[k][[0, 0, 0]]
    if n_tris > 12:
        P[k + 1] = [[0, -1, -1], [0, 1, -1], [0, 1, 1]]
        P[k + 2] = [[-1, 0, -1], [1, 0, -1], [1, 0, 1]]
    e1, e2 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]
    fn = np.cross(e1, e2)
    fn = fn / np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    N = (fn[:, None, :] + rng.normal(size=(n_tris, 3, 3)) * 0.2).astype(np.float32)     # perturbed vertex normals
    UV = rng.uniform(-1.5, 2.5, (n_tris, 3, 2)).astype(np.float32)
    images = [rng.integers(0, 256, (h, w, comp), dtype=np.uint8) for (h, w, comp) in
              ((16, 16, 3), (8, 32, 4), (5, 7, 3), (32, 32, 3))]
    mats = []
    for m in range(6):
        mt = Material(base_color=tuple(rng.uniform(0, 1, 3)), emission=tuple(rng.choice([0.0, 0.0, 2.0], 3)),
                      roughness=float(rng.choice([0.0, 0.001, 0.2, 0.7, 1.5])), metalness=float(rng.choice([0, 0.5, 0.95, 1.0])),
                      normal_map_strength=float(rng.choice([0.0, 0.5, 1.0])), sheen=float(rng.choice([0.0, 0.6])),
                      sheen_tint=float(rng.uniform(0, 1)), anisotropic_strength=float(rng.choice([0.0, 0.7])))
        if m % 2 == 0:
            mt.texture_albedo = int(rng.integers(0, 4))
            mt.texture_normal = int(rng.integers(0, 4))
        if m % 3 == 0:
            mt.texture_metal_roughness = int(rng.integers(0, 4))
            mt.texture_emission = int(rng.integers(0, 4))
        mats.append(mt)
    ids = rng.integers(0, len(mats), n_tris)
    P = (P * np.float32(scale)).astype(np.float32)
    cam = camera_from_trs((0.1 * scale, 0.2 * scale, 3.5 * scale))
    bg = procedural_background(64, 32)
    return build_scene(P, N, UV, ids, mats, images, cam, 0.9, bg, builder=builder)


@pytest.mark.parametrize("seed,n_tris", [(1, 5), (2, 8), (3, 9), (4, 64), (5, 65), (6, 400), (7, 513), (8, 3000),
                                         (9, 40000)])      # 40 000 triangles: depth 5, 4 681 nodes -- more than the LDS copy holds
def test_random_scene_bit_exact(oracle, seed, n_tris):
    import raytracing_c_amd as rt
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    hs = make_scene(seed, n_tris)
    w, h, s, b = 72, 56, 6, 7
    want = _oracle.render(hs, w, h, s, b, seed=seed)
    got = rt.render_frame(hs, w, h, s, b, seed=seed, want_linear=True, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"])
    assert np.array_equal(want["image"], got["image"])
    c = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        assert want["counters"][k] == getattr(c, k), k
    if n_tris == 40000:
        assert hs.depth == 5 and hs.n_nodes == 4681
    if n_tris >= 64:
        assert c.shades > 200 and c.textured > 50, "the scene must exercise shading and textures"


def test_duplicate_triangles_resolve_ties_like_the_oracle(oracle, diag):
    """Two coincident triangles with different materials: the hit must go to the one the reference's
    traversal order finds first (strict <), so the image differs if the two are swapped."""
    import ctypes as C
    import raytracing_c_amd as rt
    from tests import _oracle
    from tests.test_gpu_parity import _rays_for
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    hs = make_scene(11, 40)
    rng = np.random.default_rng(0)
    n = 20000
    rays = _rays_for(hs, n, rng)
    wt, wtri, wuv = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
    oracle.oracle_trace_rays(C.byref(hs.scene), n, rays.ctypes.data, wt.ctypes.data, wtri.ctypes.data, wuv.ctypes.data)
    d = rt.diag.rt_scene_upload(C.byref(hs.scene))
    assert d, rt.last_error()
    try:
        gt, gtri, guv = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
        assert rt.diag.rt_test_trace(d, n, rays.ctypes.data, gt.ctypes.data, gtri.ctypes.data, guv.ctypes.data) == 0
    finally:
        rt.diag.rt_scene_release(d)
    assert np.array_equal(wtri, gtri) and np.array_equal(wt.view(np.uint32), gt.view(np.uint32))
    # ties really occur: some hit triangle has an exact duplicate elsewhere in the triangle block
    soa = hs.soa_array().T                           # (slots, 9)
    hit_ids = np.unique(wtri[wtri >= 0])
    dup = 0
    for t in hit_ids:
        same = np.nonzero((soa == soa[t]).all(axis=1))[0]
        dup += len(same) > 1
    assert dup >= 2


def _look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """view_matrix of driver.c:765 convention: columns = camera x, y, z axes in world space (camera looks down -z), translation."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up)
    if np.linalg.norm(r) < 1e-6:
        r = np.cross(f, (1.0, 0.0, 0.0))
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    m = np.eye(4, dtype=np.float32)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, u, -f, eye
    return m


@pytest.mark.parametrize("name", ["spheres", "tower", "helmet"])
def test_tile_frustum_root_culling_never_changes_a_ray(oracle, name):
    """The tile-stream kernel lets camera rays skip the root block when the tile's pixel pyramid misses every child
    box of the root (rt_kernels.hip, tile_root_miss) and counts them as the one node visit they would have cost; node
    blocks whose lanes are camera rays on one node test only the child boxes that pyramid can touch (pyramid_cull_mask).
    Both are arguments by margin, so it gets its own fuzz: cameras far away, close up, INSIDE the scene bounds, looking
    away from the scene, grazing the bounds so that tile pyramids pass within a hair of the root's boxes, narrow and wide
    fields of view, frames whose last tiles are ragged.  Radiance sums and all seven counters must equal the oracle's
    (node_visits is the sensitive one: a wrongly skipped ray would still count 1 but lose its later visits; a wrongly
    kept one costs nothing)."""
    import raytracing_c_amd as rt
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    hs, _ = load_config(name)
    nodes = hs.nodes_array()[0]                                   # root: (6, 8) child boxes
    used = np.any(nodes != 0, axis=0)
    lo, hi = nodes[0:3][:, used].min(axis=1), nodes[3:6][:, used].max(axis=1)
    centre, ext = (lo + hi) / 2, (hi - lo) / 2
    rng = np.random.default_rng({"spheres": 21, "tower": 22, "helmet": 23}[name])
    views = []
    for _ in range(5):                                            # anywhere around, looking roughly at the scene
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        eye = centre + d * ext.max() * rng.uniform(1.5, 6.0)
        views.append((eye, centre + rng.normal(size=3) * ext * 0.8, rng.uniform(0.3, 1.6)))
    for _ in range(3):                                            # grazing: aim just past a corner of the bounds
        corner = centre + ext * rng.choice([-1.0, 1.0], 3)
        eye = centre + (corner - centre) * rng.uniform(2.0, 4.0) + rng.normal(size=3) * 0.3 * ext
        views.append((eye, corner + (corner - centre) * rng.uniform(0.01, 0.3), rng.uniform(0.2, 0.8)))
    views.append((centre + rng.uniform(-0.3, 0.3, 3) * ext, centre + rng.normal(size=3), 1.2))          # inside the bounds
    views.append((centre + np.array([0, 0, 1.0]) * ext.max() * 3, centre + np.array([0, 0, 1.0]) * ext.max() * 9, 1.0))   # looking away
    skipped_somewhere = False
    for i, (eye, target, fov) in enumerate(views):
        hs.set_camera(_look_at(eye, target), float(fov))
        # few samples: a wave mixes many pixels of the tile; 32 samples: a wave sits on two pixels, whole node blocks of
        # camera rays on one node -- the blocks that test only the children the tile's pyramid can touch
        shapes = [(88, 56, 2, 3) if i % 2 else (61, 43, 3, 2)]
        if i in (0, 3, 5, 8):
            shapes.append((48, 40, 32, 2))
        for w, h, s, b in shapes:
            want = _oracle.render(hs, w, h, s, b, seed=100 + i)
            got = rt.render_frame(hs, w, h, s, b, seed=100 + i, want_accum=True)
            assert np.array_equal(want["accum"], got["accum"]), (name, i, s)
            c = got["counters"]
            for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
                assert want["counters"][k] == getattr(c, k), (name, i, s, k)
            skipped_somewhere |= c.backgrounds > 0
    assert skipped_somewhere


@pytest.mark.parametrize("scale", [2.0 ** 36, 2.0 ** 39, 2.0 ** 45, 2.0 ** -30])
def test_scene_scale_selects_the_division(oracle, scale):
    """Leaf blocks use the short reciprocal only while the host can bound every triangle determinant below 2^102
    (largest edge component <= 2^38, rt_api.cpp); a scene beyond that is rendered by the kernel that divides.  Either
    way the frame equals the oracle's: scenes of 1e11, 1e12 and 1e13.5 units across the switch, and a tiny one."""
    import raytracing_c_amd as rt
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    hs = make_scene(31, 200, scale=scale)
    w, h, s, b = 72, 40, 8, 4
    want = _oracle.render(hs, w, h, s, b, seed=9)
    got = rt.render_frame(hs, w, h, s, b, seed=9, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"])
    c = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        assert want["counters"][k] == getattr(c, k), k
    if scale >= 1.0:
        assert c.shades > 0               # the scene is hit (a 1e-9 scene lies inside the EPSILON of every ray)


def test_long_paths_and_parked_hits(oracle):
    """Hits wait in memory (18 dwords of path state, the bounce count in 26 bits) until 48 lanes can shade together: paths
    of up to 300 bounces inside a dense triangle soup, few samples (most S blocks are sparse), ragged frame."""
    import raytracing_c_amd as rt
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    hs = make_scene(41, 3000)
    hs.set_camera(_look_at((0.05, 0.02, 0.1), (0.4, 0.3, -1.0)), 1.3)          # inside the soup
    for w, h, s, b in ((45, 37, 5, 300), (64, 40, 40, 24)):
        want = _oracle.render(hs, w, h, s, b, seed=77)
        got = rt.render_frame(hs, w, h, s, b, seed=77, want_accum=True)
        assert np.array_equal(want["accum"], got["accum"]), (w, h, s, b)
        c = got["counters"]
        for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
            assert want["counters"][k] == getattr(c, k), (k, b)
        assert c.rays > 3 * c.paths                                            # the paths really are long


def test_inverted_boxes_take_the_min_max_path(oracle):
    """The LDS node blocks of the tile-stream kernel pick a slab's near / far plane by the sign of the ray direction, which
    presumes min <= max in every child box (what scene_init builds).  A Scene from elsewhere may not keep that: the upload
    checks, and such a scene is traversed through the min / max form of raytracer.c:209-228 -- same image as the oracle,
    which always uses that form."""
    import raytracing_c_amd as rt
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    hs = make_scene(51, 900)
    nodes = hs.nodes_array()                        # (n, 6, 8) view of the host scene: rows min x y z, max x y z
    swapped = 0
    for nd in (0, 1, 5, 9):
        for k in range(8):
            if nodes[nd, 0, k] < nodes[nd, 3, k] and swapped < 6:
                nodes[nd, 0, k], nodes[nd, 3, k] = nodes[nd, 3, k].copy(), nodes[nd, 0, k].copy()     # min.x <-> max.x
                swapped += 1
    assert swapped == 6
    want = _oracle.render(hs, 96, 64, 24, 6, seed=5)
    got = rt.render_frame(hs, 96, 64, 24, 6, seed=5, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"])
    c = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        assert want["counters"][k] == getattr(c, k), k

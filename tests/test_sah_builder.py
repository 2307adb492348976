"""scene_init_sah(): the opt-in quality builder.  It must emit the SAME implicit 8-ary layout
as scene_init() -- so that the oracle and the GPU kernels traverse it unchanged -- with fewer box and triangle tests per
ray; the image may differ from a scene_init() scene only where two triangles are hit at exactly the same distance."""
import os

import numpy as np
import pytest

F = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")


def _stored_triangles(hs):
    soa = hs.soa_array()
    return np.stack([soa[0], soa[3], soa[6], soa[1], soa[4], soa[7], soa[2], soa[5], soa[8]], 1)   # v0 v1 v2 xyz


@pytest.mark.parametrize("asset", ["helmet.glb", "tower.obj", "spheres.glb", "sheen.glb", "fov_test.obj", "quad.obj"])
def test_sah_layout_is_the_reference_layout(asset):
    """Same depth / node slots / triangle slots as scene_init (pure functions of the triangle count, scene.c:224-242);
    every input triangle stored exactly once; every child box = its subtree's bounds padded by EPSILON; unpopulated
    children are the all-zero box; leaf groups hold at most 8 triangles (the slot arithmetic guarantees it)."""
    from raytracing_c_amd.loaders import load_model, load_model_data
    ref = load_model(os.path.join(ASSETS, asset))
    hs = load_model(os.path.join(ASSETS, asset), builder="sah")
    assert (hs.depth, hs.n_nodes, hs.n_slots) == (ref.depth, ref.n_nodes, ref.n_slots)
    assert int(hs.scene.bvh.last_row_offset) == int(ref.scene.bvh.last_row_offset)
    d = load_model_data(os.path.join(ASSETS, asset))
    stored = _stored_triangles(hs)
    used = np.any(stored != 0, axis=1)
    assert used.sum() == hs.n_input_triangles
    a = np.sort(stored[used].view([("", F)] * 9).ravel())
    b = np.sort(np.ascontiguousarray(d["positions"].reshape(-1, 9)).view([("", F)] * 9).ravel())
    assert np.array_equal(a, b)
    if hs.depth == 0:
        return
    nodes = hs.nodes_array()
    depth, last = hs.depth, int(hs.scene.bvh.last_row_offset)

    def bounds(index, level):
        if level == depth:
            sl = stored[(index - last) * 8:(index - last + 1) * 8]
            m = np.any(sl != 0, axis=1)
            if not m.any():
                return None
            p = sl[m].reshape(-1, 3)
            return p.min(0), p.max(0)
        lo = hi = None
        for j in range(8):
            r = bounds(8 * index + 1 + j, level + 1)
            box_lo, box_hi = nodes[index, 0:3, j], nodes[index, 3:6, j]
            if r is None:
                assert not box_lo.any() and not box_hi.any(), "unpopulated child must be the zero box"
                continue
            assert np.all(box_lo <= r[0] - F(0.9e-4)) and np.all(box_hi >= r[1] + F(0.9e-4))
            assert np.all(box_lo >= r[0] - F(1.1e-4)) and np.all(box_hi <= r[1] + F(1.1e-4)), "box is tight + eps"
            lo = r[0] if lo is None else np.minimum(lo, r[0])
            hi = r[1] if hi is None else np.maximum(hi, r[1])
        return None if lo is None else (lo, hi)

    assert bounds(0, 0) is not None


def test_sah_scene_traces_like_brute_force(oracle):
    """Closest-hit distances over the SAH tree equal those over the reference tree for random rays (same triangles,
    different boxes); the hit triangle is the same triangle wherever the distance is not tied."""
    import ctypes as C
    from raytracing_c_amd.loaders import load_model
    from tests.test_gpu_parity import _rays_for
    ref = load_model(os.path.join(ASSETS, "spheres.glb"))
    hs = load_model(os.path.join(ASSETS, "spheres.glb"), builder="sah")
    n = 4000
    rays = _rays_for(ref, n, np.random.default_rng(5))
    out = {}
    for name, sc in (("ref", ref), ("sah", hs)):
        t, tri, uv = np.zeros(n, F), np.zeros(n, np.int32), np.zeros((n, 2), F)
        oracle.oracle_trace_rays(C.byref(sc.scene), n, rays.ctypes.data, t.ctypes.data, tri.ctypes.data, uv.ctypes.data)
        out[name] = (t, tri, _stored_triangles(sc))
    assert np.array_equal(out["ref"][0], out["sah"][0])                       # same distances, bit for bit
    hit = out["ref"][1] >= 0
    assert hit.sum() > 400
    same = np.all(out["ref"][2][out["ref"][1][hit]] == out["sah"][2][out["sah"][1][hit]], axis=1)
    assert same.mean() > 0.999                                                # exact-distance ties aside


@pytest.mark.parametrize("name,w,h,s,b", [("helmet", 240, 135, 8, 8), ("tower", 240, 135, 8, 12), ("spheres", 128, 128, 8, 4)])
def test_sah_needs_fewer_tests_per_ray_and_gives_the_same_picture(name, w, h, s, b):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    r = {}
    for builder in ("reference", "sah"):
        hs, _ = load_config(name, builder=builder)
        r[builder] = _oracle.render(hs, w, h, s, b, n_threads=8)
    cr, cs = r["reference"]["counters"], r["sah"]["counters"]
    assert cs["paths"] == cr["paths"]
    assert abs(cs["rays"] / cr["rays"] - 1) < 2e-3                            # same paths up to ties
    assert cs["leaf_visits"] / cs["rays"] < 0.85 * cr["leaf_visits"] / cr["rays"]
    assert cs["node_visits"] / cs["rays"] < 0.99 * cr["node_visits"] / cr["rays"]
    differing = int(np.any(r["reference"]["image"] != r["sah"]["image"], axis=2).sum())
    assert differing <= 0.002 * w * h, differing                               # only exact-t ties may change a pixel


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,s,b", [("helmet", 160, 90, 6, 8), ("tower", 96, 54, 4, 12), ("spheres", 64, 64, 8, 4)])
def test_gpu_matches_oracle_on_sah_scenes(name, w, h, s, b):
    """GPU == oracle bit for bit on the quality builder's scenes too: both traverse the given Scene."""
    import raytracing_c_amd as rt
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    hs, _ = load_config(name, builder="sah")
    want = _oracle.render(hs, w, h, s, b)
    got = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(want["accum"], got["accum"])
    assert np.array_equal(want["image"], got["image"])
    c = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        assert want["counters"][k] == getattr(c, k), k


@pytest.mark.parametrize("bad", ["inf", "huge"])
def test_sah_keeps_every_triangle_when_surface_areas_are_not_finite(bad):
    """sah_sweep() accepts only finite costs.  One vertex at +inf, or coordinates around 1e20 whose x * y overflows,
    leave an over-capacity slice without an acceptable cut; the builder must then cut by count (as scene_init does) instead of
    pushing 200 triangles into a 64-slot subtree.  Every triangle is stored exactly once and stays reachable."""
    from raytracing_c_amd.background import procedural_background
    from raytracing_c_amd.loaders import camera_from_trs
    from raytracing_c_amd.scene import Material, build_scene
    rng = np.random.default_rng(4)
    n = 200
    P = (rng.uniform(-1, 1, (n, 1, 3)) + rng.normal(size=(n, 3, 3)) * 0.1).astype(np.float32)
    if bad == "inf":
        P[17, 1, 0] = np.inf
    else:
        P *= np.float32(1e20)
    N = np.tile(np.array([0, 0, 1], np.float32), (n, 3, 1))
    UV = np.zeros((n, 3, 2), np.float32)
    hs = build_scene(P, N, UV, np.zeros(n, np.int32), [Material()], [], camera_from_trs((0, 0, 3)), 1.0,
                     procedural_background(16, 8), builder="sah")
    ref = build_scene(P, N, UV, np.zeros(n, np.int32), [Material()], [], camera_from_trs((0, 0, 3)), 1.0,
                      procedural_background(16, 8))
    assert (hs.depth, hs.n_nodes, hs.n_slots) == (ref.depth, ref.n_nodes, ref.n_slots)
    stored = _stored_triangles(hs)
    used = np.any(stored != 0, axis=1)
    assert used.sum() == n
    a = np.sort(stored[used].view([("", F)] * 9).ravel())
    b = np.sort(np.ascontiguousarray(P.reshape(-1, 9)).view([("", F)] * 9).ravel())
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # reachable: every leaf group that holds a triangle hangs under populated (non-zero) boxes all the way up
    nodes = hs.nodes_array()
    last = int(hs.scene.bvh.last_row_offset)
    for g in np.nonzero(used.reshape(-1, 8).any(axis=1))[0]:
        child = last + int(g)
        while child > 0:
            parent, j = (child - 1) // 8, (child - 1) % 8
            assert nodes[parent, :, j].any(), f"leaf group {g} is cut off at node {parent}"
            child = parent

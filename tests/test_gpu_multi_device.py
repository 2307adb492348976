"""N GPUs behind the reference's own boundary (driver.c:793-818): the frame owner inside render_thread_proc drives
rt_device_count() devices -- scene replica, lattice share of the chunks, one peer copy of compact tiles per device, untile on
device 0.  A one-GPU box runs the same code with all logical devices mapped onto GPU 0 (rt_set_devices(n, rehearse=1) /
RT_DEVICES_REHEARSE=1): the image must equal the one-device frame byte for byte, and the counters must add up."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    import raytracing_c_amd as rt
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    yield rt
    rt.lib.rt_set_devices(1, 0)


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_rehearsed_devices_give_the_single_device_frame(rt, oracle, n_dev):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("helmet")
    w, h, s, b = 200, 120, 6, 8
    want = _oracle.render(hs, w, h, s, b)
    assert rt.lib.rt_set_devices(1, 0) == 0
    one = rt.render_context(hs, w, h, s, b, n_threads=1)
    c1 = rt.render.get_counters()
    assert rt.lib.rt_set_devices(n_dev, 1) == 0
    assert rt.lib.rt_device_count() == n_dev
    many = rt.render_context(hs, w, h, s, b, n_threads=max(2, n_dev))
    cn = rt.render.get_counters()
    rt.lib.rt_set_devices(1, 0)
    assert many["finished"] and many["n_threads"] == 0
    assert np.array_equal(one["image"], want["image"])
    assert np.array_equal(many["image"], one["image"])
    assert (cn.paths, cn.rays, cn.node_visits, cn.leaf_visits, cn.shades, cn.backgrounds) == \
        (c1.paths, c1.rays, c1.node_visits, c1.leaf_visits, c1.shades, c1.backgrounds)
    assert cn.rays == want["counters"]["rays"]


def test_c_driver_with_eight_rehearsed_devices(tmp_path, oracle):
    """examples/driver_min ... -T 8 with RT_DEVICES=8 RT_DEVICES_REHEARSE=1 in ITS environment (read once by rt_init)."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    from tests.test_c_driver import _build, _dump, _read_ppm
    exe = _build()
    scene = _dump(tmp_path, "spheres")
    out = str(tmp_path / "o.ppm")
    w, h, s, b, threads = 160, 96, 4, 4, 8
    env = dict(os.environ, RT_DEVICES="8", RT_DEVICES_REHEARSE="1")
    r = subprocess.run([exe, scene, str(w), str(h), str(s), str(b), str(threads), out], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    hs, _ = load_config("spheres")
    want = _oracle.render(hs, w, h, s, b)
    assert np.array_equal(_read_ppm(out), want["image"])


def test_scene_stamp_notices_material_and_pointer_changes(rt, oracle):
    """The per-frame stamp (dimensions, base pointers, material records, image descriptors) re-uploads a scene whose material
    changed; in-place edits of geometry bytes are the documented case for rt_scene_invalidate() / rt_scene_verify()."""
    import ctypes as C
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    w, h, s, b = 64, 40, 4, 4
    a = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(a["accum"], _oracle.render(hs, w, h, s, b)["accum"])
    assert rt.lib.rt_scene_verify(C.byref(hs.scene)) == 1
    # a material parameter changes in place: the next frame must show it without any call
    T = hs.scene.triangles
    m = C.cast(T.aos[0].shader.data, C.POINTER(abi.PBR_Shader_Data)).contents
    old = m.base_color.x
    m.base_color.x = 0.123
    b2 = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(b2["accum"], _oracle.render(hs, w, h, s, b)["accum"])
    assert not np.array_equal(b2["accum"], a["accum"])
    m.base_color.x = old
    a2 = rt.render_frame(hs, w, h, s, b, want_accum=True)      # (the restored material is a change again: uploaded afresh)
    assert np.array_equal(a2["accum"], a["accum"])
    # ONE vertex in the middle of 4 800 triangles moves in place: outside the stamp's sample of the coordinate arrays,
    # seen by the full check, which drops the copy
    i = int(T.len) // 2 + 37
    x0 = T.x[0][i]
    T.x[0][i] = x0 + 0.25
    stale = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(stale["accum"], a["accum"]), "documented: an in-place edit inside a large block needs rt_scene_invalidate"
    assert rt.lib.rt_scene_verify(C.byref(hs.scene)) == 0
    c = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(c["accum"], _oracle.render(hs, w, h, s, b)["accum"])
    T.x[0][i] = x0
    rt.lib.rt_scene_invalidate(C.byref(hs.scene))
    d = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(d["accum"], a["accum"])


def test_frame_timing_is_reported(rt):
    import ctypes as C
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config("spheres")
    rt.render_frame(hs, 128, 128, 4, 4)
    rt.render_frame(hs, 128, 128, 4, 4)
    t = abi.RT_Frame_Timing()
    assert rt.lib.rt_get_frame_timing(C.byref(t)) == 0
    assert t.total_ms > 0 and t.gpu_path_ms > 0 and t.upload_ms == 0
    assert t.stamp_ms < 0.2, "the per-frame scene check must stay in the microseconds"


@pytest.mark.parametrize("n_tiles", [1, 63, 64, 1000, 16384, 32400, 130560])
def test_preparation_kernel_orders_tiles_by_cost(rt, n_tiles, diag):
    """One launch replaces round 2's three memsets and five small kernels per frame; its counting sort (per-wave counts,
    no contended atomic) must hand out every tile exactly once, most expensive cost bucket first."""
    rng = np.random.default_rng(n_tiles)
    cost = (rng.lognormal(6, 2.5, n_tiles)).astype(np.uint32)
    cost[rng.random(n_tiles) < 0.3] = 0
    cost[: n_tiles // 3] = 4096                                  # a big bucket, as the sky tiles of a frame are
    order = np.zeros(n_tiles, np.uint32)
    assert rt.diag.rt_test_tile_order(n_tiles, cost.ctypes.data, order.ctypes.data) == 0, rt.last_error(rt.diag)
    assert np.array_equal(np.sort(order), np.arange(n_tiles, dtype=np.uint32))

    def bucket(c):
        c = c.astype(np.int64)
        e = np.floor(np.log2(np.maximum(c, 1))).astype(np.int64)
        return np.where(c < 4, c, 4 * (e - 1) + ((c >> np.maximum(e - 2, 0)) & 3))
    b = bucket(cost[order])
    assert (np.diff(b) <= 0).all()

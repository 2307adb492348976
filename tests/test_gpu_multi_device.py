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
    changed; rt_scene_verify() is the full comparison on demand.  (In-place edits inside large blocks:
    tests/test_gpu_edge_cases.py.)"""
    import ctypes as C
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    w, h, s, b = 64, 40, 4, 4
    a = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(a["accum"], _oracle.render(hs, w, h, s, b)["accum"])
    assert rt.lib.rt_scene_verify(C.byref(hs.scene)) == 1
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
    i = int(T.len) // 2 + 37
    x0 = T.x[0][i]
    T.x[0][i] = x0 + 0.25
    assert rt.lib.rt_scene_verify(C.byref(hs.scene)) == 0       # the full comparison sees one moved vertex; the copy is dropped
    c = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(c["accum"], _oracle.render(hs, w, h, s, b)["accum"])
    T.x[0][i] = x0
    rt.lib.rt_scene_invalidate(C.byref(hs.scene))
    d = rt.render_frame(hs, w, h, s, b, want_accum=True)
    assert np.array_equal(d["accum"], a["accum"])


def _move_a_vertex(hs, delta):
    """In-place edit inside the (large) coordinate block: the second vertex of every seventh triangle slot moves in y -- enough to
    change 1 500 pixels of the 160 x 128 spheres frame; -delta restores the exact floats (0.3 + y - 0.3 is not relied on: the
    callers re-upload or invalidate afterwards)."""
    T = hs.scene.triangles
    for i in range(0, int(T.len), 7):
        T.y[1][i] = T.y[1][i] + delta


def test_in_place_edit_between_two_multi_device_frames_reaches_every_device(rt, oracle):
    """every device slot keeps the full fingerprint of ITS OWN copy.  One vertex moves in place between two
    frames over 3 devices: the second frame must be the edited scene on every tile, not a mix of old and new copies."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    w, h, s, b = 160, 128, 4, 4
    assert rt.lib.rt_set_devices(3, 1) == 0
    try:
        a = rt.render_context(hs, w, h, s, b, n_threads=3)
        assert np.array_equal(a["image"], _oracle.render(hs, w, h, s, b)["image"])
        _move_a_vertex(hs, 0.3)
        want = _oracle.render(hs, w, h, s, b)["image"]
        assert not np.array_equal(want, a["image"])
        got = rt.render_context(hs, w, h, s, b, n_threads=3)
        assert np.array_equal(got["image"], want)
    finally:
        _move_a_vertex(hs, -0.3)
        rt.lib.rt_set_devices(1, 0)


def test_edit_seen_by_one_device_frame_does_not_leave_stale_copies_on_the_other_slots(rt, oracle):
    """The sequence of ADVICE r04: a frame over N devices; an in-place edit; rt_scene_verify() (or a one-device frame) brings
    slot 0 up to date; the next N-device frame must not accept slot 1 .. N-1's copies of the scene as it was before."""
    import ctypes as C
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    w, h, s, b = 160, 128, 4, 4
    try:
        for via_verify in (True, False):
            assert rt.lib.rt_set_devices(3, 1) == 0
            a = rt.render_context(hs, w, h, s, b, n_threads=3)
            assert np.array_equal(a["image"], _oracle.render(hs, w, h, s, b)["image"])
            _move_a_vertex(hs, 0.3)
            want = _oracle.render(hs, w, h, s, b)["image"]
            if via_verify:
                assert rt.lib.rt_scene_verify(C.byref(hs.scene)) == 0       # drops the copy on EVERY device
            else:
                assert rt.lib.rt_set_devices(1, 0) == 0                      # a one-device frame in between re-uploads slot 0 only
                one = rt.render_context(hs, w, h, s, b, n_threads=1)
                assert np.array_equal(one["image"], want)
                assert rt.lib.rt_set_devices(3, 1) == 0
            got = rt.render_context(hs, w, h, s, b, n_threads=3)
            assert np.array_equal(got["image"], want), via_verify
            _move_a_vertex(hs, -0.3)
            rt.lib.rt_scene_invalidate(C.byref(hs.scene))
    finally:
        rt.lib.rt_set_devices(1, 0)


def _physical_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif("_physical_gpus() < 2", reason="needs two physical GPUs (hipGetDeviceCount() >= 2)")
@pytest.mark.parametrize("force_staged", [False, True])
def test_real_devices_give_the_single_device_frame(oracle, diag, force_staged):
    """Switches itself on the day `pytest -m gpu` runs on a box with more than one GPU: N REAL devices behind
    render_thread_proc (rt_set_devices(n, 0): hipMemcpyPeerAsync over xGMI, or -- forced here -- the staged copy through pinned
    host memory), byte for byte the one-device frame, counters summed.  Matches driver.c:793-818."""
    import raytracing_c_amd as rt
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    n_dev = min(_physical_gpus(), 8)
    hs, _ = load_config("helmet")
    w, h, s, b = 320, 200, 6, 8
    want = _oracle.render(hs, w, h, s, b)
    lib = diag if force_staged else rt.lib                  # peer copies: the PRODUCT library; the staged branch needs the fault switch
    assert lib.rt_init(0) == 0
    try:
        assert lib.rt_set_devices(1, 0) == 0
        one = rt.render_context(hs, w, h, s, b, n_threads=1, lib=lib)
        c1 = rt.render.get_counters(lib)
        assert lib.rt_set_devices(n_dev, 0) == 0
        assert lib.rt_device_count() == n_dev
        diag.rt_diag_multi_fault(1 if force_staged else 0, -1)
        many = rt.render_context(hs, w, h, s, b, n_threads=n_dev, lib=lib)
        cn = rt.render.get_counters(lib)
        assert many["finished"] and many["n_threads"] == 0, rt.last_error(lib)
        assert np.array_equal(one["image"], want["image"])
        assert np.array_equal(many["image"], one["image"])
        assert (cn.paths, cn.rays, cn.node_visits, cn.leaf_visits, cn.shades, cn.backgrounds) == \
            (c1.paths, c1.rays, c1.node_visits, c1.leaf_visits, c1.shades, c1.backgrounds)
    finally:
        diag.rt_diag_multi_fault(0, -1)
        lib.rt_set_devices(1, 0)


def test_multi_device_frame_reports_the_slowest_devices_split(rt):
    """a frame over N devices reports where its time went (round 3 wiped it to total_ms only)."""
    import ctypes as C
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config("helmet")
    w, h, s, b = 512, 512, 8, 8
    assert rt.lib.rt_set_devices(4, 1) == 0
    try:
        for _ in range(3):
            r = rt.render_context(hs, w, h, s, b, n_threads=4)
            assert r["finished"] and r["n_threads"] == 0
        t = abi.RT_Frame_Timing()
        assert rt.lib.rt_get_frame_timing(C.byref(t)) == 0
        assert t.n_devices == 4 and 0 <= t.slowest_device < 4
        assert t.gpu_path_ms > 0 and t.gpu_prep_ms >= 0 and t.gpu_resolve_ms > 0 and t.gpu_copy_ms >= 0 and t.gather_ms > 0
        assert t.upload_ms == 0 and t.total_ms >= t.gpu_path_ms
        assert t.verify_ms > 0, "the full scene check runs while the devices render"
    finally:
        rt.lib.rt_set_devices(1, 0)


def test_staged_tile_copy_when_peer_access_is_refused(oracle, diag):
    """The branch a device takes when hipDeviceCanAccessPeer says no: its tiles go through pinned host memory (diagnostic
    library: rt_diag_multi_fault forces it).  Same image."""
    import raytracing_c_amd as rt
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    w, h, s, b = 160, 100, 4, 4
    want = _oracle.render(hs, w, h, s, b)
    assert diag.rt_set_devices(3, 1) == 0
    diag.rt_diag_multi_fault(1, -1)
    try:
        got = rt.render_context(hs, w, h, s, b, n_threads=3, lib=diag)
        assert got["finished"] and got["n_threads"] == 0
        assert np.array_equal(got["image"], want["image"])
        c = rt.render.get_counters(diag)
        assert c.rays == want["counters"]["rays"]
    finally:
        diag.rt_diag_multi_fault(0, -1)
        diag.rt_set_devices(1, 0)


def test_a_failing_device_leaves_the_image_untouched_and_the_protocol_terminates(oracle, diag):
    """One of N devices fails its part of the frame: render_thread_proc still terminates (n_threads reaches 0: driver.c:810-818
    must not hang), the caller's pixels are untouched, rt_last_error() names the device; the next frame is fine again."""
    import raytracing_c_amd as rt
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    w, h, s, b = 96, 64, 4, 4
    assert diag.rt_set_devices(4, 1) == 0
    diag.rt_diag_multi_fault(0, 2)
    try:
        diag.rt_clear_error()
        got = rt.render_context(hs, w, h, s, b, n_threads=4, lib=diag, fill=7)
        assert got["finished"] and got["n_threads"] == 0
        assert (got["image"] == 7).all(), "a failed frame must not write pixels"
        err = rt.last_error(diag)
        assert "slot 2 of 4" in err and "device" in err, err
        diag.rt_diag_multi_fault(0, -1)
        diag.rt_clear_error()
        ok = rt.render_context(hs, w, h, s, b, n_threads=4, lib=diag)
        assert np.array_equal(ok["image"], _oracle.render(hs, w, h, s, b)["image"])
        assert rt.last_error(diag) == ""
    finally:
        diag.rt_diag_multi_fault(0, -1)
        diag.rt_set_devices(1, 0)


def test_toggling_the_device_configuration_remaps_the_slots(rt, oracle):
    """rt_set_devices() after a slot was used must not leave a stale slot -> GPU mapping behind.  On a one-GPU box
    both mappings are GPU 0; the slots are released and initialised again, and the frames stay right."""
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("quad")
    w, h, s, b = 96, 96, 4, 4
    want = _oracle.render(hs, w, h, s, b)["image"]
    try:
        for (n, rehearse) in ((3, 1), (1, 0), (2, 1), (2, 0), (4, 1)):
            assert rt.lib.rt_set_devices(n, rehearse) == 0
            got = rt.render_context(hs, w, h, s, b, n_threads=2)
            assert got["finished"] and np.array_equal(got["image"], want), (n, rehearse)
    finally:
        rt.lib.rt_set_devices(1, 0)


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
    assert t.n_devices == 1 and t.verify_ms > 0


@pytest.mark.parametrize("n_tiles", [1, 63, 64, 1000, 16384, 32400, 130560])
def test_preparation_kernel_orders_tiles_by_cost(rt, n_tiles, diag):
    """One launch replaces round 2's three memsets and five small kernels per frame; its counting sort (per-wave counts,
    no contended atomic) must hand out every tile exactly once, most expensive cost bucket first."""
    rng = np.random.default_rng(n_tiles)
    cost = (rng.lognormal(6, 2.5, n_tiles)).astype(np.uint32)
    cost[rng.random(n_tiles) < 0.3] = 0
    cost[: n_tiles // 3] = 4096                                  # a big bucket, as the sky tiles of a frame are
    cost[n_tiles // 2] = 0xFFFFFFFF                              # (a cost >= 2^31 must not drop its tile from the order)
    if n_tiles > 5:
        cost[n_tiles - 2] = 0x80000000
    order = np.zeros(n_tiles, np.uint32)
    assert rt.diag.rt_test_tile_order(n_tiles, cost.ctypes.data, order.ctypes.data) == 0, rt.last_error(rt.diag)
    assert np.array_equal(np.sort(order), np.arange(n_tiles, dtype=np.uint32))

    def bucket(c):
        c = np.minimum(c.astype(np.int64), 0x7FFFFFFF)          # the kernel saturates: one top bucket
        e = np.floor(np.log2(np.maximum(c, 1))).astype(np.int64)
        return np.where(c < 4, c, 4 * (e - 1) + ((c >> np.maximum(e - 2, 0)) & 3))
    b = bucket(cost[order])
    assert (np.diff(b) <= 0).all()

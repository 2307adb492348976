"""rt_frame_begin / rt_frame_end (rt_hip.h): two frames on the GPU at once give the pixels and counters of the blocking entry
points bit for bit; the ticket protocol; the per-frame scene check of the blocking path still holds."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import raytracing_c_amd as rt
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    return rt


def _counters_tuple(c):
    return (c.paths, c.rays, c.node_visits, c.leaf_visits, c.shades, c.backgrounds, c.textured)


@pytest.mark.parametrize("name,w,h,s,b", [("spheres", 96, 64, 8, 4), ("helmet", 160, 96, 16, 8), ("quad", 64, 64, 4, 4)])
def test_two_frames_in_flight_equal_the_blocking_frames(rt, oracle, name, w, h, s, b):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config(name)
    seeds = [0x1234ABCD, 7, 0xDEADBEEF, 99, 12345]
    blocking = [rt.render_frame(hs, w, h, s, b, seed=sd) for sd in seeds]
    want0 = _oracle.render(hs, w, h, s, b)                       # (the oracle's default seed is seeds[0])
    assert np.array_equal(blocking[0]["image"], want0["image"])
    # begin(0), begin(1), end(0), begin(2), end(1), ...: two frames on the GPU all the time
    frames = []
    pending = []
    for sd in seeds:
        if len(pending) == 2:
            t, out, keep = pending.pop(0)
            frames.append((out, rt.frame_end(t)))
        pending.append(rt.frame_begin(hs, w, h, s, b, seed=sd))
    while pending:
        t, out, keep = pending.pop(0)
        frames.append((out, rt.frame_end(t)))
    for (out, cnt), ref in zip(frames, blocking):
        assert np.array_equal(out, ref["image"])
        assert _counters_tuple(cnt) == _counters_tuple(ref["counters"])
    assert len({f[0].tobytes() for f in frames}) == len(seeds), "different seeds give different frames"


def test_ticket_protocol(rt):
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config("spheres")
    a = rt.frame_begin(hs, 32, 32, 2, 2)
    b = rt.frame_begin(hs, 32, 32, 2, 2)
    assert {a[0], b[0]} == {0, 1}
    with pytest.raises(RuntimeError, match="in flight"):
        rt.frame_begin(hs, 32, 32, 2, 2)
    rt.frame_end(b[0])                                           # frames may be ended in any order
    with pytest.raises(RuntimeError, match="no frame in flight"):
        rt.frame_end(b[0])
    c = rt.frame_begin(hs, 32, 32, 2, 2)                         # the lane is free again
    assert c[0] == b[0]
    rt.frame_end(a[0])
    rt.frame_end(c[0])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[1], c[1])
    for bad in (-1, 2, 17):
        assert rt.lib.rt_frame_end(bad) != 0
    rt.lib.rt_clear_error()
    # a blocking frame between begin and end uses launch state 0: neither disturbs the other
    t, out, keep = rt.frame_begin(hs, 48, 40, 4, 4, seed=5)
    mid = rt.render_frame(hs, 48, 40, 4, 4, seed=6)
    rt.frame_end(t)
    assert np.array_equal(out, rt.render_frame(hs, 48, 40, 4, 4, seed=5)["image"])
    assert np.array_equal(mid["image"], rt.render_frame(hs, 48, 40, 4, 4, seed=6)["image"])


def test_frames_of_different_shapes_and_scenes_in_flight(rt, oracle):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    sp, _ = load_config("spheres")
    tw, _ = load_config("tower")
    a = rt.frame_begin(sp, 80, 48, 4, 4)
    b = rt.frame_begin(tw, 33, 65, 3, 12)
    ca = rt.frame_end(a[0])
    cb = rt.frame_end(b[0])
    wa = _oracle.render(sp, 80, 48, 4, 4)
    wb = _oracle.render(tw, 33, 65, 3, 12)
    assert np.array_equal(a[1], wa["image"]) and ca.rays == wa["counters"]["rays"]
    assert np.array_equal(b[1], wb["image"]) and cb.rays == wb["counters"]["rays"]


def test_in_place_edit_before_end_is_rendered_again(rt, oracle):
    """The full content check of the blocking path runs inside rt_frame_end(): a vertex that moved in place after the copy was
    made -- outside the bytes the sampled stamp reads -- means the frame is rendered again from the scene as it is."""
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config("spheres")
    w, h, s, b = 64, 40, 4, 4
    rt.render_frame(hs, w, h, s, b)                              # the copy is resident
    T = hs.scene.triangles
    i = int(T.len) // 2 + 37
    x0 = T.x[0][i]
    T.x[0][i] = x0 + 0.25
    fov0 = float(hs.scene.camera.fov)
    try:
        want = _oracle.render(hs, w, h, s, b)
        want2 = _oracle.render(hs, w, h, s, b, seed=22)
        t1, out1, k1 = rt.frame_begin(hs, w, h, s, b)            # both lanes render from the stale copy
        t2, out2, k2 = rt.frame_begin(hs, w, h, s, b, seed=22)
        # seed and camera belong to the frame as it was BEGUN: what the host sets up for its next frame does not leak into a
        # frame that rt_frame_end() has to render again
        rt.lib.rt_set_seed(999)
        hs.scene.camera.fov = fov0 * 0.5
        rt.frame_end(t1)
        tm = abi.RT_Frame_Timing()
        assert rt.lib.rt_get_frame_timing(C.byref(tm)) == 0 and tm.upload_ms > 0.0
        rt.frame_end(t2)
        hs.scene.camera.fov = fov0
        assert np.array_equal(out1, want["image"]) and np.array_equal(out2, want2["image"])
        t3, out3, k3 = rt.frame_begin(hs, w, h, s, b)            # unchanged now: the fresh copy serves
        rt.frame_end(t3)
        assert rt.lib.rt_get_frame_timing(C.byref(tm)) == 0 and tm.upload_ms == 0.0 and tm.verify_ms > 0.0
        assert np.array_equal(out3, want["image"])
    finally:
        T.x[0][i] = x0
        hs.scene.camera.fov = fov0
        rt.lib.rt_scene_invalidate(C.byref(hs.scene))


def test_invalidate_while_a_frame_is_in_flight(rt):
    """rt_scene_invalidate() between begin and end waits for the frame (its pixels are in the lane's buffer) before the copy goes."""
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config("helmet")
    ref = rt.render_frame(hs, 128, 128, 16, 8)
    t, out, keep = rt.frame_begin(hs, 128, 128, 16, 8)
    rt.lib.rt_scene_invalidate(C.byref(hs.scene))
    rt.frame_end(t)
    assert np.array_equal(out, ref["image"])


def test_begin_over_rehearsed_devices_renders_at_once(rt):
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config("spheres")
    ref = rt.render_frame(hs, 96, 64, 4, 4)
    assert rt.lib.rt_set_devices(3, 1) == 0
    try:
        t, out, keep = rt.frame_begin(hs, 96, 64, 4, 4)
        assert np.array_equal(out, ref["image"])                 # rendered inside begin
        cnt = rt.frame_end(t)
        assert cnt.rays == ref["counters"].rays
    finally:
        assert rt.lib.rt_set_devices(1, 0) == 0


def test_two_host_threads_one_lane_each(rt):
    """rt_frame_end() waits without the device's mutex: two host threads, each with its own frame in flight, overlap on the GPU
    and neither sees the other's pixels, counters aside (those describe the frame that ended last)."""
    import threading
    from raytracing_c_amd.configs import load_config
    sp, _ = load_config("spheres")
    he, _ = load_config("helmet")
    jobs = [(sp, 96, 64, 8, 4), (he, 80, 48, 8, 8)]
    refs = [rt.render_frame(hs, w, h, s, b)["image"] for hs, w, h, s, b in jobs]
    outs = [[], []]
    errs = []

    def worker(k):
        hs, w, h, s, b = jobs[k]
        try:
            for _ in range(12):
                t, out, keep = rt.frame_begin(hs, w, h, s, b)
                rt.frame_end(t)
                outs[k].append(out)
        except Exception as e:                                   # noqa: BLE001
            errs.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert len(outs[k]) == 12 and all(np.array_equal(o, refs[k]) for o in outs[k])

"""The wavefront pipeline (csrc/rt_wavefront.hip: camera / shade / trace kernels joined by record queues in HBM), selected
with rt_set_pipeline(1) of the DIAGNOSTIC library (the product library has one pipeline): same radiance sums and the same seven counters as the oracle, bit for bit -- also when the hit queue is
so small that a frame takes many passes, and with bounce limits beyond what any path reaches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rt():
    import raytracing_c_amd as rt
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    assert rt.diag.rt_init(0) == 0, rt.last_error(rt.diag)
    assert rt.diag.rt_set_pipeline(1) == 0
    yield rt
    rt.diag.rt_set_pipeline(0)
    rt.diag.rt_set_wavefront_capacity(96 << 20)


def _check(rt, name, w, h, s, b, **kw):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config(name, **kw)
    want = _oracle.render(hs, w, h, s, b)
    got = rt.render_frame(hs, w, h, s, b, want_accum=True, lib=rt.diag)
    assert np.array_equal(got["accum"], want["accum"])
    assert np.array_equal(got["image"], want["image"])
    c = got["counters"]
    for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
        assert want["counters"][k] == getattr(c, k), k


@pytest.mark.parametrize("name,w,h,s,b", [("quad", 64, 64, 8, 4), ("spheres", 96, 64, 8, 4), ("helmet", 160, 90, 16, 8),
                                          ("tower", 120, 80, 8, 12), ("helmet", 33, 17, 3, 1), ("spheres", 40, 40, 4, 0),
                                          ("helmet", 64, 36, 4, 40)])
def test_wavefront_frames_are_bit_exact(rt, oracle, name, w, h, s, b):
    _check(rt, name, w, h, s, b)


def test_many_passes_through_a_tiny_queue(rt, oracle):
    rt.diag.rt_set_wavefront_capacity(1024)         # the camera kernel fills its hit queue again and again
    _check(rt, "helmet", 192, 108, 8, 8)
    _check(rt, "spheres", 128, 128, 16, 4, builder="sah")

"""lightmap_bake (reference raytracer.c:722-784): GPU vs oracle, bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracing_c_amd.scene import make_image

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def _emissive_spheres():
    from raytracing_c_amd.background import procedural_background
    from raytracing_c_amd.loaders import load_model_data
    from raytracing_c_amd.scene import build_scene
    d = load_model_data(os.path.join(ASSETS, "spheres.glb"))
    for k, m in enumerate(d["materials"]):
        m.emission = (30.0 + 10 * k, 60.0, 90.0 - 10 * k)
    cam = d["camera"]
    return build_scene(d["positions"], d["normals"], d["uvs"], d["material_ids"], d["materials"], d["images"],
                       cam[0], cam[1], procedural_background())


def _oracle_bake(oracle, hs, shape, samples, fill=7):
    from tests import _oracle
    lm = np.full(shape, fill, np.uint8)
    img, keep = make_image(lm)
    cfg = _oracle.config_for(hs, n_threads=1)
    oracle.oracle_lightmap_bake(C.byref(img), C.byref(hs.scene), samples, C.byref(cfg))
    return keep


def test_oracle_lightmap_is_deterministic_and_covers_the_uv_charts(oracle):
    hs = _emissive_spheres()
    a = _oracle_bake(oracle, hs, (48, 48, 3), 2)
    b = _oracle_bake(oracle, hs, (48, 48, 3), 2)
    assert np.array_equal(a, b)
    assert (a != 7).any(axis=-1).mean() > 0.3            # the spheres' UV charts cover a good part of the map
    assert a.max() > 20                                    # emissive neighbours are seen (values are radiance, not *255)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["spheres_emissive", "tower"])
def test_gpu_lightmap_bit_exact(oracle, case):
    import raytracing_c_amd as rt
    from raytracing_c_amd.configs import load_config
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    if case == "tower":            # (helmet.glb keeps its V coordinates in [1, 2): every texel falls outside the map)
        hs, _ = load_config("tower")
        shape, samples = (96, 96, 3), 2
    else:
        hs = _emissive_spheres()
        shape, samples = (56, 64, 4), 3
    want = _oracle_bake(oracle, hs, shape, samples)
    lm = np.full(shape, 7, np.uint8)
    img, keep = make_image(lm)
    rt.lib.rt_clear_error()
    rt.lib.rt_set_seed(0x1234ABCD)          # the frame seed is process state (rt_hip.h); earlier tests change it
    rt.lib.lightmap_bake(C.byref(img), C.byref(hs.scene), samples)
    assert rt.last_error() == ""
    assert np.array_equal(keep, want)
    assert (keep[..., :3] != 7).any()

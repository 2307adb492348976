"""The helmet's silhouette at the reference's own 1024x1024 against the edges of its sample render output.png
(tests/_silhouette.py, fixture tests/golden/reference_output_png_1024_luma.npz): CPU = the oracle, GPU = the product path.
Narrows what a loader / node-transform / camera / builder error could hide behind "GPU == oracle": both sides render the
Scene the product's loader and scene_init build, and until round 4 only five landmarks at 128x128 tied that Scene to anything
the reference holds.  (Does not lift "parity unpinned": one picture is not a test vector.)"""
import numpy as np
import pytest

from tests import _silhouette as S


def test_oracle_silhouette_sits_on_the_reference_pictures_edges():
    from tests import _oracle
    hs = S.white_environment_scene()
    r = _oracle.render(hs, S.SIZE, S.SIZE, 16, 1, n_threads=8)
    salient, aligned = S.check(S.mask_from_frame(r["image"]))
    assert aligned > 0.87


@pytest.mark.gpu
def test_product_path_silhouette_sits_on_the_reference_pictures_edges():
    import raytracing_c_amd as rt
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    hs = S.white_environment_scene()
    got = rt.render_frame(hs, S.SIZE, S.SIZE, 16, 1)
    S.check(S.mask_from_frame(got["image"]))
    want = _oracle.render(hs, S.SIZE, S.SIZE, 16, 1, n_threads=8)
    assert np.array_equal(got["image"], want["image"])

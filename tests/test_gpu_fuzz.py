"""Seeded fuzz of the whole hot path (the one-off script of round 2, tests/fuzz_parity.py, as part of the suite): 200 random cases
-- triangle soups of 3 ... 6 000 triangles, both builders, scene scales 1e-3 ... 2^20, cameras far / inside / grazing / looking away,
frames 9x9 ... 140x90, 1 ... 64 spp, 0 ... 40 bounces -- must give the oracle's radiance sums and all seven counters bit for bit,
through the tile-stream kernel and (every fourth case) through the wavefront pipeline."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(seed):
    from tests.test_gpu_random_scenes import make_scene, _look_at
    rng = np.random.default_rng(seed)
    n_tris = int(rng.choice([3, 9, 40, 200, 700, 2500, 6000]))
    scale = float(rng.choice([1.0, 1.0, 1.0, 1e-3, 37.0, 2.0 ** 20]))
    builder = str(rng.choice(["reference", "reference", "sah"]))
    hs = make_scene(seed, n_tris, builder=builder, scale=scale)
    mode = int(rng.integers(0, 4))
    c = rng.uniform(-1, 1, 3) * scale
    if mode == 0:      # far, looking at the soup
        eye = c + rng.normal(size=3) * scale * rng.uniform(2, 8)
        tgt = rng.uniform(-0.5, 0.5, 3) * scale
    elif mode == 1:    # inside
        eye = rng.uniform(-0.6, 0.6, 3) * scale
        tgt = eye + rng.normal(size=3) * scale
    elif mode == 2:    # grazing past the soup
        eye = c + rng.normal(size=3) * scale * 3
        tgt = eye + np.cross(eye, rng.normal(size=3))
    else:              # looking away
        eye = c + rng.normal(size=3) * scale * 4
        tgt = eye * 2.0
    hs.set_camera(_look_at(eye, tgt), float(rng.uniform(0.2, 2.2)))
    w, h = int(rng.integers(9, 140)), int(rng.integers(9, 90))
    s = int(rng.choice([1, 2, 5, 16, 33, 64]))
    b = int(rng.choice([0, 1, 3, 8, 40]))
    if w * h * s > 200000:
        s = max(1, 200000 // (w * h))
    return hs, w, h, s, b, (n_tris, scale, builder, mode)


@pytest.mark.parametrize("block", range(8))
def test_fuzz_parity(oracle, block):
    import raytracing_c_amd as rt
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    assert rt.diag.rt_init(0) == 0, rt.last_error(rt.diag)
    assert rt.diag.rt_set_pipeline(1) == 0      # every fourth case: the wavefront pipeline of the diagnostic library
    bad = []
    try:
        for seed in range(7000 + 25 * block, 7000 + 25 * (block + 1)):
            hs, w, h, s, b, what = _case(seed)
            want = _oracle.render(hs, w, h, s, b, seed=seed)
            got = rt.render_frame(hs, w, h, s, b, seed=seed, want_accum=True, lib=rt.diag if seed % 4 == 3 else None)
            ok = np.array_equal(want["accum"], got["accum"])
            for k in ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured"):
                ok = ok and want["counters"][k] == getattr(got["counters"], k)
            if not ok:
                bad.append((seed, what, w, h, s, b))
    finally:
        rt.diag.rt_set_pipeline(0)
    assert not bad, bad

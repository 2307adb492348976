"""A plain C host (examples/driver_min.c) drives librt_hip.so through the reference's own protocol:
scene_init -> Rendering_Context -> N threads on render_thread_proc -> poll -> denoise_image."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")
ASSETS = os.path.join(ROOT, "assets")


def _build():
    subprocess.check_call(["make", "-C", EX], stdout=subprocess.DEVNULL)
    return os.path.join(EX, "driver_min")


def _dump(tmp_path, name):
    from raytracing_c_amd.background import procedural_background
    from raytracing_c_amd.configs import CONFIGS
    from raytracing_c_amd.loaders import default_camera, load_model_data
    from raytracing_c_amd.scene_dump import write_scene_dump
    asset, _, _, _, _, cam = CONFIGS[name]
    data = load_model_data(os.path.join(ASSETS, asset))
    cam = cam or data["camera"] or default_camera()
    path = str(tmp_path / (name + ".rtscene"))
    write_scene_dump(path, data, cam[0], cam[1], procedural_background())
    return path


def _read_ppm(path):
    raw = open(path, "rb").read()
    magic, dims, maxv, rest = raw.split(b"\n", 3)
    w, h = [int(v) for v in dims.split()]
    assert magic == b"P6" and maxv == b"255"
    return np.frombuffer(rest, np.uint8).reshape(h, w, 3)


def test_c_driver_builds_as_c11_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    scene = _dump(tmp_path, "quad")
    r = subprocess.run([exe, scene, "32", "32", "2", "2", "2", str(tmp_path / "o.ppm")], capture_output=True, text=True)
    assert r.returncode == 2
    assert "render failed" in r.stderr and "HIP" in r.stderr
    assert not os.path.exists(tmp_path / "o.ppm")


@pytest.mark.gpu
@pytest.mark.parametrize("name,threads,denoise", [("quad", 1, False), ("helmet", 4, False), ("spheres", 3, True)])
def test_c_driver_frame_matches_oracle(tmp_path, oracle, name, threads, denoise):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    from tests.test_denoiser import _oracle_denoise
    exe = _build()
    scene = _dump(tmp_path, name)
    out = str(tmp_path / "o.ppm")
    w, h, s, b = 96, 54, 4, 6
    cmd = [exe, scene, str(w), str(h), str(s), str(b), str(threads), out] + (["-D"] if denoise else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = _read_ppm(out)
    hs, _ = load_config(name)
    want = _oracle.render(hs, w, h, s, b)["image"]
    if denoise:
        want = _oracle_denoise(oracle, want)
    assert np.array_equal(got, want)


def _decode_qoi(raw):
    """reference decoder of the QOI format (qoiformat.org), RGB"""
    assert raw[:4] == b"qoif"
    w, h, ch = int.from_bytes(raw[4:8], "big"), int.from_bytes(raw[8:12], "big"), raw[12]
    assert ch == 3 and raw[-8:] == bytes([0, 0, 0, 0, 0, 0, 0, 1])
    out = np.zeros((w * h, 3), np.uint8)
    index = [(0, 0, 0, 0)] * 64
    px = (0, 0, 0, 255)
    p, i = 14, 0
    while i < w * h:
        b1 = raw[p]
        p += 1
        run = 1
        if b1 == 0xFE:
            px = (raw[p], raw[p + 1], raw[p + 2], px[3])
            p += 3
        elif b1 >> 6 == 0:
            px = index[b1]
        elif b1 >> 6 == 1:
            px = ((px[0] + ((b1 >> 4) & 3) - 2) & 255, (px[1] + ((b1 >> 2) & 3) - 2) & 255, (px[2] + (b1 & 3) - 2) & 255, px[3])
        elif b1 >> 6 == 2:
            b2 = raw[p]
            p += 1
            dg = (b1 & 63) - 32
            px = ((px[0] + dg - 8 + (b2 >> 4)) & 255, (px[1] + dg) & 255, (px[2] + dg - 8 + (b2 & 15)) & 255, px[3])
        else:
            run = (b1 & 63) + 1
        index[(px[0] * 3 + px[1] * 5 + px[2] * 7 + px[3] * 11) % 64] = px
        out[i:i + run] = px[:3]
        i += run
    return out.reshape(h, w, 3)


@pytest.mark.gpu
def test_c_driver_writes_png_and_qoi_like_the_reference_driver(tmp_path, oracle):
    """driver.c:839-877 picks png / qoi / ppm by the suffix of -O; the C host does the same with self-contained writers."""
    from PIL import Image
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    exe = _build()
    scene = _dump(tmp_path, "spheres")
    hs, _ = load_config("spheres")
    want = _oracle.render(hs, 96, 64, 4, 4)["image"]
    for suffix in ("png", "qoi"):
        out = str(tmp_path / f"o.{suffix}")
        r = subprocess.run([exe, scene, "96", "64", "4", "4", "2", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.asarray(Image.open(out).convert("RGB")) if suffix == "png" else _decode_qoi(open(out, "rb").read())
        assert np.array_equal(got, want), suffix

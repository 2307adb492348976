"""examples/rt_model.c: model loading in C for a host of librt_hip.so (driver.c:510-728 does it with
codin's obj.h / gltf.h / stb_image, which are not in the reference tree).  Textures come from RT8I side files
(tools/extract_textures.py) when they exist, else from the streams the model embeds or names: baseline JPEG (examples/rt_jpeg.c,
which restates libjpeg's default arithmetic: the same bytes PIL gives the Python loader) or PNG (examples/rt_png.c).

The C loader is checked against the Python loader the benchmark configs use (same triangles, normals, uvs, materials,
texture assignment, camera), and -- on the GPU -- through examples/driver_min: a C host that takes a MODEL PATH like
driver.c:685-728 renders the same bytes as the oracle on the C-loaded scene."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")
ASSETS = os.path.join(ROOT, "assets")


def _lib():
    from raytracing_c_amd import ctypes_abi as abi
    import raytracing_c_amd as rt
    rt.lib.rt_last_error()                                   # librt_hip.so first (librt_model.so links against it)
    subprocess.check_call(["make", "-C", EX], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(EX, "librt_model.so"))

    class RT_Model(C.Structure):
        _fields_ = [("triangles", C.c_void_p), ("n_triangles", abi.isize), ("materials", C.POINTER(abi.PBR_Shader_Data)),
                    ("n_materials", abi.isize), ("images", C.POINTER(abi.Image)), ("n_images", abi.isize),
                    ("has_camera", C.c_bool), ("camera", abi.Camera)]

    lib.rt_model_load.argtypes = [C.c_char_p, C.POINTER(RT_Model), C.c_char_p, C.c_size_t]
    lib.rt_model_load.restype = C.c_bool
    lib.rt_model_free.argtypes = [C.POINTER(RT_Model)]
    lib.rt_model_default_camera.restype = abi.Camera
    lib.rt_model_camera.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float]
    lib.rt_model_camera.restype = abi.Camera
    return lib, RT_Model


def _load_c(prefix):
    """C loader output as the numpy arrays / Material list raytracing_c_amd.scene.build_scene takes."""
    from raytracing_c_amd import ctypes_abi as abi
    from raytracing_c_amd.scene import Material
    lib, RT_Model = _lib()
    m = RT_Model()
    err = C.create_string_buffer(512)
    assert lib.rt_model_load(prefix.encode(), C.byref(m), err, 512), err.value.decode()
    n = int(m.n_triangles)
    tri = np.frombuffer(C.string_at(m.triangles, n * abi.TRIANGLE_DTYPE.itemsize), abi.TRIANGLE_DTYPE).copy()
    mat_base = C.addressof(m.materials.contents)
    img_base = C.addressof(m.images.contents) if m.n_images else 0

    def tex(ptr):
        return None if not ptr else (C.addressof(ptr.contents) - img_base) // C.sizeof(abi.Image)

    mats = []
    for k in range(int(m.n_materials)):
        d = m.materials[k]
        mats.append(Material(base_color=(d.base_color.x, d.base_color.y, d.base_color.z),
                             emission=(d.emission.x, d.emission.y, d.emission.z), roughness=d.roughness, metalness=d.metalness,
                             normal_map_strength=d.normal_map_strength, sheen=d.sheen, sheen_tint=d.sheen_tint,
                             anisotropic_strength=d.anisotropic_strength, texture_albedo=tex(d.texture_albedo),
                             texture_normal=tex(d.texture_normal), texture_metal_roughness=tex(d.texture_metal_roughness),
                             texture_emission=tex(d.texture_emission)))
    images = []
    for k in range(int(m.n_images)):
        im = m.images[k]
        if im.pixels.data:
            images.append(np.frombuffer(C.string_at(im.pixels.data, im.pixels.len), np.uint8).reshape(im.height, im.width, im.components).copy())
        else:
            images.append(np.zeros((1, 1, 3), np.uint8))            # an image no material uses: not loaded
    out = dict(positions=tri["positions"].copy(), normals=tri["normals"].copy(), uvs=tri["tex_coords"].copy(),
               material_ids=((tri["shader_data"].astype(np.int64) - mat_base) // C.sizeof(abi.PBR_Shader_Data)),
               materials=mats, images=images, camera=None)
    if m.has_camera:
        cam = np.ctypeslib.as_array(m.camera.view_matrix.rows).reshape(4, 4).copy()
        out["camera"] = (cam, float(m.camera.fov), float(m.camera.focal_length))
    lib.rt_model_free(C.byref(m))
    return out


@pytest.mark.parametrize("asset", ["quad.obj", "tower.obj", "fov_test.obj", "spheres.glb", "sheen.glb", "helmet.glb"])
def test_c_loader_matches_the_python_loader(tmp_path, asset):
    from raytracing_c_amd.loaders import load_model_data
    from tools.extract_textures import extract
    prefix = extract(os.path.join(ASSETS, asset), str(tmp_path))
    c = _load_c(prefix)
    p = load_model_data(os.path.join(ASSETS, asset))
    assert c["positions"].shape == p["positions"].shape
    assert np.array_equal(c["material_ids"], p["material_ids"])
    assert np.array_equal(c["uvs"], p["uvs"])
    exact = asset.endswith(".obj")                            # text -> strtod -> f32 on both sides: identical
    for key in ("positions", "normals"):
        a, b = c[key], p[key].astype(np.float32)
        if exact:
            assert np.array_equal(a, b), key
        else:                                                 # node transforms: float64 sums in a different order (BLAS vs loops)
            assert np.allclose(a, b, rtol=0, atol=2e-7), key
            assert (a == b).mean() > 0.999, (key, (a == b).mean())
    assert len(c["materials"]) == len(p["materials"])
    for mc, mp in zip(c["materials"], p["materials"]):
        for f in ("roughness", "metalness", "normal_map_strength", "sheen", "sheen_tint", "anisotropic_strength"):
            assert np.float32(getattr(mc, f)) == np.float32(getattr(mp, f)), f
        assert np.array_equal(np.float32(mc.base_color), np.float32(mp.base_color))
        assert np.array_equal(np.float32(mc.emission), np.float32(mp.emission))
        for f in ("texture_albedo", "texture_normal", "texture_metal_roughness", "texture_emission"):
            assert getattr(mc, f) == getattr(mp, f), f
    for k, im in enumerate(p["images"]):
        if any(k in (m.texture_albedo, m.texture_normal, m.texture_metal_roughness, m.texture_emission) for m in p["materials"]):
            assert np.array_equal(c["images"][k], im)
    if p["camera"] is None:
        assert c["camera"] is None
    else:
        assert np.array_equal(c["camera"][0], p["camera"][0].astype(np.float32))
        assert np.float32(c["camera"][1]) == np.float32(p["camera"][1])
        focal = np.float32(1.0) / np.tan(np.float32(p["camera"][1]) * np.float32(0.5), dtype=np.float32)
        assert abs(np.float32(c["camera"][2]) - focal) <= 2 * np.spacing(focal)      # tanf vs numpy's float32 tan: <= 1 ulp apart


def test_c_loader_fails_loudly(tmp_path):
    lib, RT_Model = _lib()
    m = RT_Model()
    err = C.create_string_buffer(512)
    assert not lib.rt_model_load(str(tmp_path / "nothing.obj").encode(), C.byref(m), err, 512) and b"cannot read" in err.value
    assert not lib.rt_model_load(b"model.fbx", C.byref(m), err, 512) and b"Unrecognized file type" in err.value     # driver.c:724-727
    blob = bytearray(open(os.path.join(ASSETS, "helmet.glb"), "rb").read())   # a codec rt_jpeg.c does not read, no side files
    at = blob.index(b"\xff\xc0\x00\x11\x08")                 # SOF0 of the first embedded image -> SOF9 (arithmetic coding)
    blob[at + 1] = 0xC9
    bad = tmp_path / "helmet_arithmetic.glb"
    bad.write_bytes(bytes(blob))
    assert not lib.rt_model_load(str(bad).encode(), C.byref(m), err, 512)
    assert b"arithmetic" in err.value and b"extract_textures" in err.value


def _jpeg_decode(lib, data):
    from raytracing_c_amd import ctypes_abi as abi
    lib.rt_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(abi.Image), C.c_char_p, C.c_size_t]
    lib.rt_jpeg_decode.restype = C.c_bool
    img = abi.Image()
    err = C.create_string_buffer(256)
    if not lib.rt_jpeg_decode(data, len(data), C.byref(img), err, 256):
        return None, err.value.decode()
    a = np.frombuffer(C.string_at(img.pixels.data, img.pixels.len), np.uint8).reshape(img.height, img.width, 3).copy()
    C.CDLL(None).free(C.c_void_p(img.pixels.data))
    return a, ""


@pytest.mark.parametrize("size", [(64, 64), (33, 17), (1, 1), (2, 3), (4, 5), (3, 8), (5, 2), (6, 6), (100, 75), (17, 33), (250, 129)])
def test_rt_jpeg_is_libjpeg_bit_for_bit(size):
    """examples/rt_jpeg.c against PIL (libjpeg-turbo, default islow IDCT + fancy upsampling): every subsampling PIL writes,
    odd sizes (edge replication of the chroma planes), restart intervals, optimised Huffman tables, grayscale."""
    import io
    from PIL import Image as PI
    lib, _ = _lib()
    w, h = size
    rng = np.random.default_rng(w * 1000 + h)
    im = PI.fromarray(rng.integers(0, 256, (h // 4 + 1, w // 4 + 1, 3), dtype=np.uint8)).resize((w, h), PI.BILINEAR)
    noisy = PI.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    cases = [(src, dict(quality=q, subsampling=sub)) for src in (im, noisy) for sub in (0, 1, 2) for q in (30, 95)]
    cases += [(im.convert("L"), dict(quality=80)), (im, dict(quality=80, restart_marker_blocks=1)),
              (noisy, dict(quality=60, restart_marker_rows=1, subsampling=2)), (im, dict(quality=85, optimize=True))]
    for src, kw in cases:
        bio = io.BytesIO()
        src.save(bio, "JPEG", **kw)
        data = bio.getvalue()
        if "restart_marker_blocks" in kw or "restart_marker_rows" in kw:
            assert b"\xff\xdd" in data
        got, msg = _jpeg_decode(lib, data)
        assert got is not None, (kw, msg)
        assert np.array_equal(got, np.asarray(PI.open(io.BytesIO(data)).convert("RGB"))), kw
    for src, kw in [(im, dict(progressive=True)), (noisy, dict(progressive=True, quality=92, subsampling=0)),
                    (im, dict(progressive=True, quality=40, subsampling=1)), (im.convert("L"), dict(progressive=True)),
                    (noisy, dict(progressive=True, quality=75, restart_marker_blocks=2)), (im, dict(progressive=True, optimize=True, quality=97))]:
        bio = io.BytesIO()
        src.save(bio, "JPEG", **kw)
        assert b"\xff\xc2" in bio.getvalue()                                  # SOF2: DC / AC bands, successive approximation, EOB runs
        got, msg = _jpeg_decode(lib, bio.getvalue())
        assert got is not None, (kw, msg)
        assert np.array_equal(got, np.asarray(PI.open(io.BytesIO(bio.getvalue())).convert("RGB"))), kw
    got, msg = _jpeg_decode(lib, b"\x89PNG\r\n\x1a\n" + bytes(64))
    assert got is None and "not a JPEG" in msg
    got, msg = _jpeg_decode(lib, data[: len(data) // 3])              # truncated entropy-coded data: zeros are fed, no crash
    assert got is None or got.shape == (h, w, 3)


def _image_decode(lib, data, which):
    from raytracing_c_amd import ctypes_abi as abi
    fn = getattr(lib, which)
    fn.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(abi.Image), C.c_char_p, C.c_size_t]
    fn.restype = C.c_bool
    img = abi.Image()
    err = C.create_string_buffer(256)
    if not fn(data, len(data), C.byref(img), err, 256):
        return None, err.value.decode()
    a = np.frombuffer(C.string_at(img.pixels.data, img.pixels.len), np.uint8).reshape(img.height, img.width, 3).copy()
    C.CDLL(None).free(C.c_void_p(img.pixels.data))
    return a, ""


def _raw_png(w, h, depth, ctype, rows, plte=None, interlace=0):
    """A PNG written by hand (filter 0 rows, zlib level 6) for the formats PIL does not write."""
    import struct
    import zlib

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if plte is not None:
        out += chunk(b"PLTE", plte)
    raw = b"".join(b"\0" + r for r in rows)
    if interlace == 1 and depth >= 8:                            # Adam7: the seven sub-images, each with its own filter bytes
        bpp = len(rows[0]) // w
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            for y in range(y0, h, dy):
                px = b"".join(rows[y][x * bpp:(x + 1) * bpp] for x in range(x0, w, dx))
                if px:
                    raw += b"\0" + px
    cut = len(raw) // 2                                          # two IDAT chunks: the stream may be split anywhere
    z = zlib.compress(raw)
    return out + chunk(b"IDAT", z[:cut]) + chunk(b"IDAT", z[cut:]) + chunk(b"IEND", b"")


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (64, 64), (33, 17), (257, 100)])
def test_rt_png_gives_what_pil_gives(size):
    """examples/rt_png.c against PIL's .convert("RGB"): every mode PIL writes (1, L, LA, P at 1 / 2 / 4 / 8 bits, RGB, RGBA), stored /
    fixed / dynamic deflate blocks, all five filters (PIL picks per row), plus hand-written streams of the formats PIL only reads."""
    import io
    from PIL import Image as PI
    lib, _ = _lib()
    w, h = size
    rng = np.random.default_rng(w * 977 + h)
    smooth = np.asarray(PI.fromarray(rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 4), dtype=np.uint8)).resize((w, h), PI.BICUBIC))
    noisy = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    for arr in (smooth, noisy):
        rgb = PI.fromarray(arr[..., :3].copy(), "RGB")
        srcs = [PI.fromarray(arr, "RGBA"), rgb, PI.fromarray(arr[..., 0].copy(), "L"), PI.fromarray(arr[..., :2].copy(), "LA"),
                rgb.quantize(200), rgb.quantize(13), rgb.quantize(3), rgb.quantize(2), PI.fromarray(arr[..., 0].copy(), "L").convert("1")]
        for im in srcs:
            for kw in (dict(compress_level=0), dict(compress_level=1), dict(compress_level=9), dict(optimize=True)):
                bio = io.BytesIO()
                im.save(bio, "PNG", **kw)
                got, msg = _image_decode(lib, bio.getvalue(), "rt_png_decode")
                assert got is not None, (im.mode, kw, msg)
                assert np.array_equal(got, np.asarray(PI.open(io.BytesIO(bio.getvalue())).convert("RGB"))), (im.mode, kw)
    for depth, ctype, ch in [(2, 0, 1), (4, 0, 1), (1, 0, 1), (16, 2, 3), (16, 6, 4), (16, 4, 2), (8, 4, 2), (4, 3, 1), (2, 3, 1)]:
        rows = [bytes(rng.integers(0, 256, (w * ch * depth + 7) // 8, dtype=np.uint8)) for _ in range(h)]
        plte = bytes(rng.integers(0, 256, 3 * 3, dtype=np.uint8)) if ctype == 3 else None        # a SHORT palette: black beyond it
        data = _raw_png(w, h, depth, ctype, rows, plte)
        got, msg = _image_decode(lib, data, "rt_png_decode")
        assert got is not None, (depth, ctype, msg)
        assert np.array_equal(got, np.asarray(PI.open(io.BytesIO(data)).convert("RGB"))), (depth, ctype)
    for depth, ctype, ch in [(8, 2, 3), (8, 6, 4), (8, 0, 1), (16, 2, 3), (8, 3, 1)]:                          # Adam7
        rows = [bytes(rng.integers(0, 256, w * ch * depth // 8, dtype=np.uint8)) for _ in range(h)]
        data = _raw_png(w, h, depth, ctype, rows, bytes(rng.integers(0, 256, 768, dtype=np.uint8)) if ctype == 3 else None, interlace=1)
        got, msg = _image_decode(lib, data, "rt_png_decode")
        assert got is not None, (depth, ctype, msg)
        assert np.array_equal(got, np.asarray(PI.open(io.BytesIO(data)).convert("RGB"))), ("adam7", depth, ctype)
    rows = [bytes(2 * w) for _ in range(h)]
    for data, word in ((_raw_png(w, h, 16, 0, rows), "16-bit grayscale"), (_raw_png(w, h, 8, 0, rows, interlace=2), "interlace"),
                       (b"\xff\xd8\xff\xe0" + bytes(32), "not a PNG"), (_raw_png(w, h + 1, 8, 0, [bytes(w)] * h), "zlib stream")):
        got, msg = _image_decode(lib, data, "rt_png_decode")
        assert got is None and word in msg, (word, msg)
    good = _raw_png(w, h, 8, 2, [bytes(rng.integers(0, 256, 3 * w, dtype=np.uint8)) for _ in range(h)])
    for cut in (20, 40, len(good) // 2, len(good) - 13):                  # truncated anywhere: a message, not a crash
        got, msg = _image_decode(lib, good[:cut], "rt_png_decode")
        assert got is None and msg


def test_c_loader_reads_the_texture_maps_of_an_mtl(tmp_path):
    """map_Kd / map_Ke / norm / map_Pm of a .mtl (driver.c:549-568 loads them through stb_image): PNG and JPEG files next to the .mtl,
    decoded in C, in the order and with the assignment of the Python loader; a map that is not there is no map."""
    from PIL import Image as PI
    from raytracing_c_amd.loaders import load_model_data
    rng = np.random.default_rng(5)
    obj = open(os.path.join(ASSETS, "quad.obj")).read()
    (tmp_path / "q.obj").write_text(obj.replace("quad.mtl", "q.mtl").replace("usemtl Material", "usemtl A", 1)
                                    .replace("f 1/1/1 3/4/1 4/2/1", "usemtl B\nf 1/1/1 3/4/1 4/2/1"))
    (tmp_path / "q.mtl").write_text("newmtl A\nKd 0.5 0.6 0.7\nmap_Kd -s 1 1 1 kd.png\nmap_Ke ke.jpg\nnorm missing.png\n"
                                    "newmtl B\nKd 1 1 1\nPr 0.3\nPm 0.9\nmap_Kd kd.png\nnorm n.png\nmap_Pm pm.jpg\nmap_Ke nothing.png\n")
    PI.fromarray(rng.integers(0, 256, (16, 24, 4), dtype=np.uint8), "RGBA").save(tmp_path / "kd.png")
    PI.fromarray(rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)).resize((40, 24), PI.BILINEAR).save(tmp_path / "ke.jpg", quality=90)
    PI.fromarray(rng.integers(0, 256, (9, 5), dtype=np.uint8), "L").save(tmp_path / "n.png")
    PI.fromarray(rng.integers(0, 256, (4, 4, 3), dtype=np.uint8)).resize((17, 33), PI.BICUBIC).save(tmp_path / "pm.jpg", subsampling=1)
    c = _load_c(str(tmp_path / "q.obj"))
    p = load_model_data(str(tmp_path / "q.obj"))
    assert len(p["images"]) == 5 and len(c["images"]) == 5          # A: kd, ke; B: kd, n, pm (norm of the non-PBR material A does not count)
    for a, b in zip(c["images"], p["images"]):
        assert np.array_equal(a, b)
    for mc, mp in zip(c["materials"], p["materials"]):
        for f in ("texture_albedo", "texture_normal", "texture_metal_roughness", "texture_emission"):
            assert getattr(mc, f) == getattr(mp, f), f
    assert np.array_equal(c["material_ids"], p["material_ids"])
    (tmp_path / "kd.png").write_bytes(b"\x89PNG\r\n\x1a\n" + bytes(40))             # a map that is there but broken: loud
    lib, RT_Model = _lib()
    m = RT_Model()
    err = C.create_string_buffer(512)
    assert not lib.rt_model_load(str(tmp_path / "q.obj").encode(), C.byref(m), err, 512) and b"kd.png" in err.value


def test_c_loader_decodes_png_images_of_a_gltf(tmp_path):
    """A .gltf whose material names a PNG by uri and another embedded in the buffer: the C loader's texels == the Python loader's."""
    import io
    import json
    from PIL import Image as PI
    from raytracing_c_amd.loaders import load_model_data
    rng = np.random.default_rng(9)
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    uv = np.array([[0, 0], [1, 0], [0, 1]], np.float32)
    bio = io.BytesIO()
    PI.fromarray(rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)).quantize(40).save(bio, "PNG")
    emb = bio.getvalue()
    blob = pos.tobytes() + uv.tobytes() + emb
    (tmp_path / "t.bin").write_bytes(blob)
    PI.fromarray(rng.integers(0, 256, (31, 15, 4), dtype=np.uint8), "RGBA").save(tmp_path / "base.png")
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "material": 0}]}],
           "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}, "emissiveTexture": {"index": 1}, "emissiveFactor": [1, 1, 1]}],
           "textures": [{"source": 0}, {"source": 1}],
           "images": [{"uri": "base.png"}, {"bufferView": 2, "mimeType": "image/png"}],
           "buffers": [{"uri": "t.bin", "byteLength": len(blob)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24},
                           {"buffer": 0, "byteOffset": 60, "byteLength": len(emb)}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
                         {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"}]}
    (tmp_path / "t.gltf").write_text(json.dumps(doc))
    c = _load_c(str(tmp_path / "t.gltf"))
    p = load_model_data(str(tmp_path / "t.gltf"))
    assert len(p["images"]) == 2
    for k in range(2):
        assert np.array_equal(c["images"][k], p["images"][k]), k
    assert c["materials"][0].texture_albedo == p["materials"][0].texture_albedo == 0
    assert c["materials"][0].texture_emission == p["materials"][0].texture_emission == 1


def test_c_readers_survive_damaged_files_under_sanitizers(tmp_path):
    """examples/fuzz_images.c: the JPEG and PNG decoders and the OBJ / MTL / glTF / JSON readers of the C host, built with
    -fsanitize=address,undefined, fed thousands of mutated files (bit flips, random bytes, truncations, spliced blocks): every one
    either loads or is refused WITH a message; no out-of-bounds access, overflow or leak (the sanitizers abort the run)."""
    import io
    import json
    import shutil
    from PIL import Image as PI
    subprocess.check_call(["make", "-C", EX, "fuzz_images"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(1)
    a = np.asarray(PI.fromarray(rng.integers(0, 256, (9, 9, 3), dtype=np.uint8)).resize((67, 45), PI.BICUBIC))
    PI.fromarray(a).save(tmp_path / "f1.jpg", quality=85)
    PI.fromarray(a).save(tmp_path / "f2.jpg", quality=60, subsampling=0, restart_marker_blocks=3)
    PI.fromarray(a).convert("L").save(tmp_path / "f3.jpg")
    PI.fromarray(a).save(tmp_path / "f9.jpg", quality=80, progressive=True)
    PI.fromarray(a).save(tmp_path / "f4.png")
    PI.fromarray(a).quantize(16).save(tmp_path / "f5.png")
    PI.fromarray(a[..., 0].copy()).save(tmp_path / "f6.png", compress_level=0)
    (tmp_path / "f7.png").write_bytes(_raw_png(37, 21, 8, 2, [bytes(rng.integers(0, 256, 37 * 3, dtype=np.uint8)) for _ in range(21)], interlace=1))
    (tmp_path / "f8.png").write_bytes(_raw_png(19, 9, 2, 3, [bytes(rng.integers(0, 256, 5, dtype=np.uint8)) for _ in range(9)],
                                               bytes(rng.integers(0, 256, 12, dtype=np.uint8))))
    images = [str(tmp_path / f) for f in ("f1.jpg", "f2.jpg", "f3.jpg", "f4.png", "f5.png", "f6.png", "f7.png", "f8.png", "f9.jpg")]
    r = subprocess.run([os.path.join(EX, "fuzz_images"), "1500"] + images, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ok, bad = (int(w) for w in r.stdout.split() if w.isdigit())
    assert ok > 1000 and bad > 1000, r.stdout
    # models: an OBJ with its MTL and maps, and a .gltf whose JSON names an external buffer, a PNG by uri and one in the buffer
    for f in ("quad.obj", "quad.mtl"):
        shutil.copy(os.path.join(ASSETS, f), tmp_path / f)
    with open(tmp_path / "quad.mtl", "a") as f:
        f.write("map_Kd f4.png\nmap_Ke f1.jpg\nPr 0.5\nnorm f5.png\n")
    bio = io.BytesIO()
    PI.fromarray(a).quantize(40).save(bio, "PNG")
    emb = bio.getvalue()
    blob = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32).tobytes() + np.array([[0, 0], [1, 0], [0, 1]], np.float32).tobytes() +
            np.array([0, 1, 2, 0], np.uint16).tobytes() + emb)
    (tmp_path / "t.bin").write_bytes(blob)
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 1]}],
           "nodes": [{"mesh": 0, "translation": [1, 2, 3], "rotation": [0, 0.7071, 0, 0.7071], "scale": [1, 2, 1], "children": [2]},
                     {"camera": 0, "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 5, 1]}, {"mesh": 0}],
           "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.1}}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 2, "material": 0}]}],
           "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "baseColorFactor": [1, 0.5, 0.25, 1], "metallicFactor": 0.5},
                          "emissiveTexture": {"index": 1}, "emissiveFactor": [1, 1, 1], "name": 'm\u00e9tal "q" \\ back',
                          "extensions": {"KHR_materials_sheen": {"sheenColorFactor": [0.1, 0.2, 0.3]}}}],
           "textures": [{"source": 0}, {"source": 1}], "images": [{"uri": "f4.png"}, {"bufferView": 3, "mimeType": "image/png"}],
           "buffers": [{"uri": "t.bin", "byteLength": len(blob)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24},
                           {"buffer": 0, "byteOffset": 60, "byteLength": 6}, {"buffer": 0, "byteOffset": 68, "byteLength": len(emb)}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"},
                         {"bufferView": 2, "componentType": 5123, "count": 3, "type": "SCALAR"}]}
    (tmp_path / "t.gltf").write_text(json.dumps(doc))
    work = tmp_path / "mutants"
    work.mkdir()
    for f in ("quad.mtl", "f1.jpg", "f4.png", "f5.png", "t.bin"):           # what the mutants refer to by name
        shutil.copy(tmp_path / f, work / f)
    r = subprocess.run([os.path.join(EX, "fuzz_images"), "-m", str(work), "4000", str(tmp_path / "quad.obj"), str(tmp_path / "t.gltf"),
                        os.path.join(ASSETS, "sheen.glb")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ok, bad = (int(w) for w in r.stdout.split() if w.isdigit())
    assert ok > 1000 and bad > 1000, r.stdout


def test_c_loader_decodes_the_embedded_jpegs_of_the_helmet(tmp_path):
    """No preparation step: helmet.glb by itself (four 2048 x 2048 baseline 4:2:0 JPEGs) -> the texels the Python loader gets."""
    from raytracing_c_amd.loaders import load_model_data
    link = tmp_path / "helmet.glb"
    os.symlink(os.path.join(ASSETS, "helmet.glb"), link)
    c = _load_c(str(link))
    p = load_model_data(os.path.join(ASSETS, "helmet.glb"))
    used = {t for m in p["materials"] for t in (m.texture_albedo, m.texture_normal, m.texture_metal_roughness, m.texture_emission)
            if t is not None}
    assert len(used) == 4
    for k in used:
        assert np.array_equal(c["images"][k], p["images"][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("asset,camera,side_files", [("helmet.glb", None, True), ("helmet.glb", None, False), ("spheres.glb", None, True),
                                                     ("tower.obj", "0 12.5 32 0 0 0 1 1.2217305", True)])
def test_c_driver_takes_a_model_path(tmp_path, oracle, asset, camera, side_files):
    """driver_min MODEL ... : load with rt_model.c, scene_init, render_thread_proc threads, PPM -- byte-equal to the oracle's
    render of the same C-loaded scene (camera of the file, or the override the tower config needs); once from the
    helmet's .glb alone, its JPEGs decoded in C."""
    from raytracing_c_amd.background import procedural_background
    from raytracing_c_amd.scene import build_scene
    from tests import _oracle
    from tests.test_c_driver import _read_ppm
    from tools.extract_textures import extract
    prefix = extract(os.path.join(ASSETS, asset), str(tmp_path), background=True)
    if not side_files:                                        # the .glb alone: rt_jpeg.c decodes what it embeds
        import glob
        for f in glob.glob(prefix + ".image*.rgb8"):
            os.remove(f)
    out = str(tmp_path / "o.ppm")
    w, h, s, b = 96, 54, 4, 6
    cmd = [os.path.join(EX, "driver_min"), prefix, str(w), str(h), str(s), str(b), "3", out, "--background", prefix + ".background.rgb8"]
    if camera:
        cmd += ["--camera", camera]
    _lib()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = _read_ppm(out)
    c = _load_c(prefix)
    if camera:                                                # the same override, built by the same C function the driver calls
        lib, _ = _lib()
        v = (C.c_float * 8)(*[float(x) for x in camera.split()])
        cc = lib.rt_model_camera(C.cast(v, C.POINTER(C.c_float)), C.cast(C.byref(v, 12), C.POINTER(C.c_float)), v[7])
        cam, fov, focal = np.ctypeslib.as_array(cc.view_matrix.rows).reshape(4, 4).copy(), float(cc.fov), float(cc.focal_length)
    else:
        cam, fov, focal = c["camera"]
    hs = build_scene(c["positions"], c["normals"], c["uvs"], c["material_ids"], c["materials"], c["images"], cam, fov,
                     procedural_background())
    hs.scene.camera.focal_length = focal                      # the C loader's tanf, not numpy's float32 tan
    want = _oracle.render(hs, w, h, s, b)["image"]
    assert np.array_equal(got, want)

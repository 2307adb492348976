"""The C-ABI library loads and exports every symbol include/*.h declares; no compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import raytracing_c_amd as rt
    from raytracing_c_amd import ctypes_abi as abi
    for name in abi.EXPORTED_SYMBOLS:
        assert getattr(rt.lib, name) is not None, name


def test_product_library_has_one_pipeline_and_no_test_entry_points():
    """librt_hip.so exports no rt_wf_*, no rt_set_pipeline, no rt_test_*; they live in librt_hip_diag.so
    (include/rt_hip_diag.h), which exports the product's symbols as well."""
    import subprocess
    import raytracing_c_amd as rt
    from raytracing_c_amd import ctypes_abi as abi
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {l.split()[-1] for l in out.splitlines() if l.strip()}
    prod = exported(rt.native.LIB_PATH)
    bad = sorted(n for n in prod if n.startswith(("rt_wf_", "rt_test_", "rt_launch_test_")) or "wavefront" in n or n in abi.DIAG_ONLY_SYMBOLS)
    assert bad == [], bad
    assert not any("rt_wf_" in n or "rt_test_" in n for n in prod)          # (mangled kernel names included)
    diag = exported(rt.native.DIAG_PATH)
    for name in abi.EXPORTED_SYMBOLS + abi.DIAG_ONLY_SYMBOLS:
        assert name in diag, name
    for name in abi.DIAG_ONLY_SYMBOLS:
        assert getattr(rt.diag, name) is not None, name


def test_headers_and_symbol_list_agree():
    """Every `extern` function of include/*.h is in EXPORTED_SYMBOLS and vice versa."""
    from raytracing_c_amd import ctypes_abi as abi
    declared = set()
    for h in ("rt_scene.h", "rt_raytracer.h", "rt_materials.h", "rt_hip.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"extern\s+[^;(]*?\b(\w+)\s*\(", text):
            declared.add(m.group(1))
    assert declared == set(abi.EXPORTED_SYMBOLS), declared ^ set(abi.EXPORTED_SYMBOLS)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rt_hip_diag.h")).read(), flags=re.S)
    diag_declared = {m.group(1) for m in re.finditer(r"extern\s+[^;(]*?\b(\w+)\s*\(", text)}
    assert diag_declared == set(abi.DIAG_ONLY_SYMBOLS), diag_declared ^ set(abi.DIAG_ONLY_SYMBOLS)


def test_struct_sizes_match_reference_contract():
    """Sizes SURVEY.md section 8 quotes for the reference's structs (scene.h, raytracer.h, driver.c)."""
    from raytracing_c_amd import ctypes_abi as abi
    assert C.sizeof(abi.BVH_Node) == 192
    assert C.sizeof(abi.Triangle) == 112
    assert C.sizeof(abi.Triangle_AOS) == 112
    assert C.sizeof(abi.Hit) == 88
    assert C.sizeof(abi.Ray) == 24
    assert C.sizeof(abi.PBR_Shader_Data) == 80
    assert C.sizeof(abi.Shader) == 16
    assert C.sizeof(abi.Camera) == 72


def test_chunk_arithmetic():
    import raytracing_c_amd as rt
    from raytracing_c_amd.multi_gpu import FramePartition
    assert rt.lib.rt_chunk_count(1920, 1080) == 60 * 34 == 2040
    assert rt.lib.rt_chunk_count(3840, 2160) == 120 * 68 == 8160
    for (w, h, world) in [(1920, 1080, 8), (100, 70, 3), (33, 1, 2), (64, 64, 5)]:
        part = FramePartition(w, h, world)
        total = 0
        for r in range(world):
            n = rt.lib.rt_local_chunk_count(w, h, r, world)
            assert n == part.n_local(r)
            total += n
        assert total == part.n_chunks == rt.lib.rt_chunk_count(w, h)


def test_render_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without a HIP device the render entry points report an error."""
    import torch
    import raytracing_c_amd as rt
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    from raytracing_c_amd.configs import load_config
    hs, _ = load_config("quad")
    rt.lib.rt_clear_error()
    img = np.full((16, 16, 3), 7, np.uint8)
    image, _k = rt.scene.make_image(img)
    image.pixels.data = img.ctypes.data
    rc = rt.lib.rt_render_frame(C.byref(hs.scene), C.byref(image), 4, 2, None, None)
    assert rc != 0
    assert "no HIP device" in rt.last_error() or "failed" in rt.last_error()
    assert (img == 7).all(), "image must stay untouched"
    rt.lib.rt_clear_error()
    assert rt.lib.rt_frame_begin(C.byref(hs.scene), C.byref(image), 4, 2) < 0        # frames in flight: no device, no ticket
    assert "no HIP device" in rt.last_error() or "failed" in rt.last_error()
    assert rt.lib.rt_frame_end(0) != 0 and "no frame in flight" in rt.last_error()
    assert (img == 7).all()
    # the reference protocol still completes (n_threads reaches 0) so a driver does not hang
    r = rt.render_context(hs, 16, 16, 2, 2, n_threads=2)
    assert r["finished"] and r["n_threads"] == 0


def test_material_tokens_are_not_callable_shaders():
    import raytracing_c_amd as rt
    from raytracing_c_amd import ctypes_abi as abi
    rt.lib.rt_clear_error()
    out = abi.Shader_Output()
    inp = abi.Shader_Input()
    fn = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(abi.Shader_Input), C.POINTER(abi.Shader_Output))(
        rt.native.symbol_address("disney_shader_proc"))
    fn(None, C.byref(inp), C.byref(out))
    assert out.terminate
    assert "token" in rt.last_error()

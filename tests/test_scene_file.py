"""The `.scene` cache file (reference scene.c:13-76): byte layout, round trip, rejection of
damaged files, and (GPU) that a loaded scene renders the same image as the scene it was saved from."""
import ctypes as C
import struct

import numpy as np
import pytest

import raytracing_c_amd as rt
from raytracing_c_amd import ctypes_abi as abi
from raytracing_c_amd.configs import load_config

HEADER = 96          # {i32 x4, Camera 72 B} = 88, aligned(32) -> 96


def _save(scene):
    n = rt.lib.scene_file_size(C.byref(scene))
    raw = np.zeros(n + 32, dtype=np.uint8)
    off = (-raw.ctypes.data) % 32
    buf = raw[off:off + n]
    assert buf.ctypes.data % 32 == 0
    assert rt.lib.scene_save_bytes(C.byref(scene), buf.ctypes.data, n) == n
    return buf


def _load(buf):
    sc = abi.Scene()
    ok = rt.lib.scene_load_bytes(abi.Byte_Slice(buf.ctypes.data, buf.size), C.byref(sc))
    return ok, sc


@pytest.mark.parametrize("name", ["quad", "spheres"])
def test_layout_and_round_trip(name):
    hs, _ = load_config(name)
    sc = hs.scene
    buf = _save(sc)
    n_nodes, n_tris = sc.bvh.nodes.len, sc.triangles.len
    assert buf.size == HEADER + 192 * n_nodes + n_tris * (9 * 4 + 112)        # scene.c:45
    version, hn, ht, hd = struct.unpack_from("<4i", buf, 0)
    assert (version, hn, ht, hd) == (0, n_nodes, n_tris, sc.bvh.depth)
    cam = np.frombuffer(buf, dtype="<f4", count=18, offset=16)                 # Matrix_4x4 + fov + focal_length
    assert np.array_equal(cam[:16].reshape(4, 4), np.ctypeslib.as_array(sc.camera.view_matrix.rows).reshape(4, 4))
    assert cam[16] == sc.camera.fov and cam[17] == sc.camera.focal_length
    if n_nodes:
        nodes = np.ctypeslib.as_array(C.cast(sc.bvh.nodes.data, C.POINTER(C.c_float)), shape=(n_nodes * 48,))
        assert np.array_equal(np.frombuffer(buf, "<f4", n_nodes * 48, HEADER).view(np.uint32), nodes.view(np.uint32))
    x0 = np.ctypeslib.as_array(sc.triangles.x[0], shape=(n_tris,))
    assert np.array_equal(np.frombuffer(buf, "<f4", n_tris, HEADER + 192 * n_nodes), x0)

    ok, ld = _load(buf)
    assert ok
    assert (ld.bvh.depth, ld.bvh.nodes.len, ld.bvh.last_row_offset, ld.triangles.len) == \
           (sc.bvh.depth, n_nodes, sc.bvh.last_row_offset, n_tris)
    # aliases the buffer (no copy), pointers laid out as scene.c:60-73
    base = buf.ctypes.data + HEADER + 192 * n_nodes
    assert C.addressof(ld.triangles.x[0].contents) == base
    assert C.addressof(ld.triangles.y[0].contents) == base + 4 * n_tris * 3
    assert C.addressof(ld.triangles.z[2].contents) == base + 4 * n_tris * 8
    assert C.addressof(ld.triangles.aos.contents) == base + 4 * n_tris * 9
    assert np.array_equal(_save(ld), buf)                                        # save(load(save(s))) == save(s)


def test_rejects_damaged_files():
    hs, _ = load_config("spheres")
    buf = _save(hs.scene)
    assert not _load(buf[:64])[0]                                  # shorter than the header
    assert not _load(buf[:buf.size - 32])[0]                       # truncated
    bad = buf.copy()
    struct.pack_into("<i", bad, 4, 7)                              # node count that does not match the size
    assert not _load(bad)[0]
    bad = buf.copy()
    struct.pack_into("<i", bad, 0, 1)                              # unknown version
    assert not _load(bad)[0]
    raw = np.zeros(buf.size + 64, dtype=np.uint8)
    off = (-raw.ctypes.data) % 32 + 4                              # misaligned
    mis = raw[off:off + buf.size]
    mis[:] = buf
    assert not _load(mis)[0]
    small = np.zeros(16, dtype=np.uint8)
    assert rt.lib.scene_save_bytes(C.byref(hs.scene), small.ctypes.data, small.size) == -1


@pytest.mark.gpu
def test_loaded_scene_renders_like_the_original():
    from raytracing_c_amd.render import render_frame
    hs, cfg = load_config("spheres")
    ref = render_frame(hs, 96, 64, 4, 4, seed=0x1234ABCD)
    buf = _save(hs.scene)
    ok, ld = _load(buf)
    assert ok
    ld.background = hs.scene.background                              # not part of the file (scene.c:18-34)
    loaded = type(hs).__new__(type(hs))                             # a view: owns nothing, never frees
    loaded.__dict__.update(hs.__dict__)
    loaded.scene = ld
    loaded._freed = True
    try:
        out = render_frame(loaded, 96, 64, 4, 4, seed=0x1234ABCD)
    finally:
        rt.lib.rt_scene_invalidate(C.byref(ld))                      # drop the device copy cached for this Scene*
    assert np.array_equal(out["image"], ref["image"])
    assert out["counters"].rays == ref["counters"].rays

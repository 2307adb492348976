"""Host-side scene assembly: numpy triangle soup + materials + images -> the reference's
`Scene` (include/rt_scene.h) through the native scene_init (reference scene.c:416-426).

Mirrors what driver.c:510-683 does after parsing a model: one PBR_Shader_Data per material,
one Triangle per face with shader = {&material, disney_shader_proc}, background = an Image
looked up by sample_background.
"""
import ctypes as C
import time
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import ctypes_abi as abi
from .native import lib, last_error, symbol_address


@dataclass
class Material:
    """Fields of PBR_Shader_Data (driver.c:191-198); textures are indices into `images`."""
    base_color: tuple = (0.8, 0.8, 0.8)
    emission: tuple = (0.0, 0.0, 0.0)
    roughness: float = 0.5
    metalness: float = 0.0
    normal_map_strength: float = 0.0
    sheen: float = 0.0
    sheen_tint: float = 0.0
    anisotropic_strength: float = 0.0
    texture_albedo: Optional[int] = None
    texture_normal: Optional[int] = None
    texture_metal_roughness: Optional[int] = None
    texture_emission: Optional[int] = None


def make_image(pixels: np.ndarray):
    """(H, W, C) uint8 -> (abi.Image, backing array)."""
    arr = np.ascontiguousarray(pixels, dtype=np.uint8)
    if arr.ndim != 3 or arr.shape[2] < 3:
        raise ValueError("image must be (H, W, C>=3) uint8")
    img = abi.Image()
    img.components = arr.shape[2]
    img.pixel_type = 0
    img.width = arr.shape[1]
    img.stride = arr.shape[1]
    img.height = arr.shape[0]
    img.pixels.data = arr.ctypes.data
    img.pixels.len = arr.size
    return img, arr


def set_camera(cam: abi.Camera, matrix: np.ndarray, yfov: float):
    m = np.asarray(matrix, dtype=np.float32).reshape(4, 4)
    for i in range(4):
        for j in range(4):
            cam.view_matrix.rows[i][j] = float(m[i, j])
    cam.fov = float(np.float32(yfov))
    # driver.c:607,767  focal_length = 1 / tan(fov / 2), evaluated in fp32
    cam.focal_length = float(np.float32(1.0) / np.tan(np.float32(yfov) * np.float32(0.5), dtype=np.float32))


class HostScene:
    """Owns a native Scene plus everything it points to."""

    def __init__(self):
        self.scene = abi.Scene()
        self.materials = None          # ctypes array of PBR_Shader_Data
        self.images: List[abi.Image] = []
        self._image_arrays = []
        self.background_image = None
        self._bg_array = None
        self.n_input_triangles = 0
        self.shader_kind = "disney"
        self._freed = False
        self.scene_init_seconds = 0.0

    # --- convenience views -------------------------------------------------------------
    @property
    def depth(self):
        return int(self.scene.bvh.depth)

    @property
    def n_nodes(self):
        return int(self.scene.bvh.nodes.len)

    @property
    def n_slots(self):
        return int(self.scene.triangles.len)

    def nodes_array(self):
        n = self.n_nodes
        if n == 0:
            return np.zeros((0, 6, 8), np.float32)
        return np.ctypeslib.as_array(C.cast(self.scene.bvh.nodes.data, C.POINTER(C.c_float)), (n, 6, 8))

    def soa_array(self):
        """(9, len) view: x0 x1 x2 y0 y1 y2 z0 z1 z2."""
        n = self.n_slots
        return np.ctypeslib.as_array(self.scene.triangles.x[0], (9, n))

    def populated_nodes(self):
        nodes = self.nodes_array()
        if len(nodes) == 0:
            return 0
        return int(np.count_nonzero(np.any(nodes.reshape(len(nodes), 48) != 0, axis=1)))

    def populated_leaves(self):
        soa = self.soa_array()
        n = self.n_slots
        used = np.any(soa.reshape(9, n // 8, 8) != 0, axis=(0, 2))
        return int(np.count_nonzero(used))

    def set_camera(self, matrix, yfov):
        set_camera(self.scene.camera, matrix, yfov)

    def free(self):
        if not self._freed:
            try:
                lib.rt_scene_invalidate(C.byref(self.scene))
                from . import native
                if native.diag._dll is not None:            # a test process that rendered this scene through the diagnostic library too
                    native.diag.rt_scene_invalidate(C.byref(self.scene))
            except Exception:
                pass
            lib.rt_scene_free(C.byref(self.scene))
            self._freed = True

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def build_scene(positions, normals, uvs, material_ids, materials: List[Material], images, camera_matrix, yfov,
                background, shader="disney", builder="reference") -> HostScene:
    """positions/normals: (N,3,3) f32, uvs: (N,3,2) f32, material_ids: (N,) int,
    images: list of (H,W,C) uint8, background: (H,W,C) uint8 equirect sRGB."""
    positions = np.ascontiguousarray(positions, np.float32)
    n = positions.shape[0]
    normals = np.ascontiguousarray(normals, np.float32).reshape(n, 3, 3)
    uvs = np.ascontiguousarray(uvs, np.float32).reshape(n, 3, 2)
    material_ids = np.asarray(material_ids, np.int64).reshape(n)
    if not materials:
        raise ValueError("need at least one material")

    hs = HostScene()
    hs.n_input_triangles = n
    hs.shader_kind = shader
    for im in images:
        ci, arr = make_image(im)
        hs.images.append(ci)
        hs._image_arrays.append(arr)

    def tex(i):
        return C.pointer(hs.images[i]) if i is not None else None

    hs.materials = (abi.PBR_Shader_Data * len(materials))()
    for k, m in enumerate(materials):
        d = hs.materials[k]
        d.base_color = abi.Vec3(*[float(np.float32(v)) for v in m.base_color])
        d.emission = abi.Vec3(*[float(np.float32(v)) for v in m.emission])
        d.roughness = m.roughness
        d.metalness = m.metalness
        d.normal_map_strength = m.normal_map_strength
        d.sheen = m.sheen
        d.sheen_tint = m.sheen_tint
        d.anisotropic_strength = m.anisotropic_strength
        if m.texture_albedo is not None:
            d.texture_albedo = tex(m.texture_albedo)
        if m.texture_normal is not None:
            d.texture_normal = tex(m.texture_normal)
        if m.texture_metal_roughness is not None:
            d.texture_metal_roughness = tex(m.texture_metal_roughness)
        if m.texture_emission is not None:
            d.texture_emission = tex(m.texture_emission)

    proc = symbol_address("disney_shader_proc" if shader == "disney" else "debug_shader_proc")
    mat_base = C.addressof(hs.materials)
    tri = np.zeros(n, abi.TRIANGLE_DTYPE)
    tri["positions"] = positions
    tri["normals"] = normals
    tri["tex_coords"] = uvs
    tri["shader_data"] = mat_base + material_ids.astype(np.uint64) * C.sizeof(abi.PBR_Shader_Data)
    tri["shader_proc"] = proc

    bg_img, bg_arr = make_image(background)
    hs.background_image = bg_img
    hs._bg_array = bg_arr
    hs.scene.background.proc = symbol_address("sample_background")
    hs.scene.background.data = C.addressof(hs.background_image)

    set_camera(hs.scene.camera, camera_matrix, yfov)

    sl = abi.Triangle_Slice(tri.ctypes.data, n)
    t0 = time.perf_counter()
    if builder == "reference":
        lib.scene_init(C.byref(hs.scene), sl, abi.Allocator(None, None))          # scene.c:416-426
    elif builder == "sah":
        lib.scene_init_sah(C.byref(hs.scene), sl, abi.Allocator(None, None))      # opt-in quality builder, same layout
    elif builder == "gpu":
        if lib.scene_init_gpu(C.byref(hs.scene), sl, abi.Allocator(None, None)) != 0:   # scene_init by GPU kernels, same bytes
            raise RuntimeError("scene_init_gpu: " + last_error())
    else:
        raise ValueError(f"unknown BVH builder {builder!r}")
    hs.builder = builder
    hs.scene_init_seconds = time.perf_counter() - t0
    if not hs.scene.triangles.x[0]:
        raise MemoryError("scene_init failed")
    return hs

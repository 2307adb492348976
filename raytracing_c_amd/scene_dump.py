"""Flat binary dump of loader output (triangles, materials, images, camera) for C hosts
(examples/driver_min.c).  Not the reference's `.scene` cache file (scene.c:13-76, out of scope): that one
stores raw function pointers; this one stores what driver.c's loaders hand to scene_init."""
import struct

import numpy as np


def write_scene_dump(path, data, camera_matrix, yfov, background):
    """data: dict from loaders.load_model_data(); camera_matrix 4x4, yfov radians; background (H,W,3) uint8."""
    images = [np.ascontiguousarray(im, np.uint8) for im in data["images"]] + [np.ascontiguousarray(background, np.uint8)]
    n = len(data["positions"])
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", 0x43535452, 1, n, len(data["materials"]), len(images), len(images) - 1))
        m = np.asarray(camera_matrix, np.float32).reshape(16)
        fov = np.float32(yfov)
        focal = np.float32(1.0) / np.tan(fov * np.float32(0.5), dtype=np.float32)
        f.write(m.tobytes() + struct.pack("<2f", float(fov), float(focal)))
        tri = np.zeros(n, np.dtype([("pos", "<f4", 9), ("nrm", "<f4", 9), ("uv", "<f4", 6), ("mat", "<i4")]))
        tri["pos"] = np.asarray(data["positions"], np.float32).reshape(n, 9)
        tri["nrm"] = np.asarray(data["normals"], np.float32).reshape(n, 9)
        tri["uv"] = np.asarray(data["uvs"], np.float32).reshape(n, 6)
        tri["mat"] = np.asarray(data["material_ids"], np.int32)
        f.write(tri.tobytes())
        for mt in data["materials"]:
            t = [(-1 if v is None else int(v)) for v in (mt.texture_albedo, mt.texture_normal,
                                                          mt.texture_metal_roughness, mt.texture_emission)]
            f.write(struct.pack("<12f4i", *[float(np.float32(v)) for v in mt.base_color],
                                *[float(np.float32(v)) for v in mt.emission], mt.roughness, mt.metalness,
                                mt.normal_map_strength, mt.sheen, mt.sheen_tint, mt.anisotropic_strength, *t))
        for im in images:
            f.write(struct.pack("<3i", im.shape[1], im.shape[0], im.shape[2]))
            f.write(im.tobytes())

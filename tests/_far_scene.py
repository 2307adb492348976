"""spheres.glb with every vertex AND the camera translated by (k, k, k): the same picture, far from the origin.
Used by the tests of deviation D9's domain (include/rt_math.h, RT_SLAB_FUSED_MAX_ORIGIN)."""
import os

import numpy as np

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def translated_spheres(k, builder="reference"):
    from raytracing_c_amd.background import procedural_background
    from raytracing_c_amd.loaders import load_model_data
    from raytracing_c_amd.scene import build_scene
    d = load_model_data(os.path.join(ASSETS, "spheres.glb"))
    pos = d["positions"] + np.float32(k)
    cam = np.array(d["camera"][0], np.float32).copy()
    cam[:3, 3] += np.float32(k)
    return build_scene(pos, d["normals"], d["uvs"], d["material_ids"], d["materials"], d["images"], cam, d["camera"][1],
                       procedural_background(), builder=builder)

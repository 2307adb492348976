"""The five configurations BASELINE.json names, with the concrete cameras of SURVEY.md 8d."""
import math
import os

import numpy as np

from .loaders import camera_from_trs, load_model

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")

FOV70 = float(np.float32(np.float32(70.0) / np.float32(360.0)) * np.float32(np.pi) * np.float32(2.0))

CONFIGS = {
    # name: (asset, width, height, samples, bounces, camera override or None)
    "spheres": ("spheres.glb", 256, 256, 16, 4, None),
    "quad": ("quad.obj", 512, 512, 64, 4,
             (camera_from_trs((3, 0, 0), (0, math.sin(math.pi / 4), 0, math.cos(math.pi / 4))), FOV70)),
    "helmet": ("helmet.glb", 1920, 1080, 256, 8, None),
    "tower": ("tower.obj", 1920, 1080, 512, 12, (camera_from_trs((0, 12.5, 32)), FOV70)),
    "helmet4k": ("helmet.glb", 3840, 2160, 1024, 16, None),
}


def load_config(name, shader="disney", builder="reference"):
    """builder: "reference" = scene_init (the reference's split, scene.c:311-414; every headline number) or "sah" =
    scene_init_sah (opt-in quality builder emitting the same layout)."""
    asset, w, h, s, b, cam = CONFIGS[name]
    hs = load_model(os.path.join(ASSETS, asset), camera=cam, shader=shader, builder=builder)
    return hs, dict(width=w, height=h, samples=s, max_bounces=b, asset=asset)

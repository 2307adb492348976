"""Procedural environment map.

The reference driver loads `background.png` from the working directory (driver.c:758-763);
that file is a missing blob of the reference, so the configs use this closed-form
2048x1024 equirectangular sRGB image instead: a sky gradient, a warm horizon band, a dark
ground and a bright sun disc (values are integers computed with exact integer arithmetic, so
the image is identical on every machine).
"""
import numpy as np


def procedural_background(width=2048, height=1024):
    y = np.arange(height, dtype=np.int64)[:, None]
    x = np.arange(width, dtype=np.int64)[None, :]
    h2 = height // 2
    img = np.zeros((height, width, 3), np.int64)
    # sky: top (y=0) deep blue -> horizon pale
    t = np.clip(y * 256 // max(h2, 1), 0, 256)          # 0 at zenith, 256 at horizon
    sky = np.stack([(60 + (150 * t) // 256), (110 + (110 * t) // 256), (200 + (40 * t) // 256)], -1)
    # ground: horizon brownish -> nadir dark
    g = np.clip((y - h2) * 256 // max(h2, 1), 0, 256)
    ground = np.stack([(110 - (80 * g) // 256), (95 - (70 * g) // 256), (80 - (60 * g) // 256)], -1)
    img[:] = np.where((y < h2)[..., None], np.broadcast_to(sky, img.shape), np.broadcast_to(ground, img.shape))
    # gentle azimuthal variation so that left/right are distinguishable
    az = (np.abs((x * 512 // width) % 512 - 256) * 24) // 256      # 0..24
    img[..., 0] += az
    img[..., 1] += az // 2
    # sun disc (integer distance test in pixel space, aspect corrected)
    sx, sy, r = (width * 5) // 8, height // 4, height // 24
    d2 = (x - sx) ** 2 + ((y - sy) * 1) ** 2
    sun = d2 <= r * r
    halo = (d2 <= (3 * r) ** 2) & ~sun
    img[sun] = (255, 250, 235)
    img[halo] = np.minimum(img[halo] + 40, 255)
    return np.clip(img, 0, 255).astype(np.uint8)

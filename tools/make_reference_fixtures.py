#!/usr/bin/env python3
"""Fixtures derived from the reference's ONE full-frame artifact, its sample render `output.png` (helmet.gltf, 1024x1024, the
reference's own environment map -- a missing blob here).  Data, not source: images reduced from that picture.

    tests/golden/reference_output_png_128.npz        128x128 box-filtered RGB (landmarks: tests/test_oracle_kat.py)
    tests/golden/reference_output_png_1024_luma.npz  1024x1024 mean of R, G, B, u8 (silhouette: tests/_silhouette.py)

    python tools/make_reference_fixtures.py [/root/reference/output.png]       (this container only; the GPU box has no reference)"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main(path):
    img = Image.open(path).convert("RGB")
    assert img.size == (1024, 1024), img.size
    small = np.asarray(img.resize((128, 128), Image.BOX))
    old = os.path.join(GOLDEN, "reference_output_png_128.npz")
    if os.path.exists(old):
        assert np.array_equal(np.load(old)["image"], small), "the committed 128x128 fixture is not this reduction of output.png"
    else:
        np.savez_compressed(old, image=small, note=np.array("output.png of the reference, 1024x1024 -> 128x128 box filter"))
    ref = np.asarray(img).astype(np.float32)
    luma = np.clip(np.rint(ref.mean(-1)), 0, 255).astype(np.uint8)
    np.savez_compressed(os.path.join(GOLDEN, "reference_output_png_1024_luma.npz"), luma=luma,
                        note=np.array("mean of R, G, B of the reference's sample render output.png (helmet.gltf, 1024x1024, its own "
                                      "environment map), u8; tools/make_reference_fixtures.py"))
    print("fixtures written:", small.shape, luma.shape)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/output.png")

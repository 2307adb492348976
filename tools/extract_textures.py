#!/usr/bin/env python3
"""Writes the RT8I side files the C model loader (examples/rt_model.c) reads instead of decoding JPEG / PNG:
    <model>.image<k>.rgb8   for every image k of the model, in the order of raytracing_c_amd/loaders.py
    <model>.background.rgb8 the procedural environment map of the benchmark configs (--background)
Header: b"RT8I", i32 width, height, components (little endian), then the rows.
    python tools/extract_textures.py assets/helmet.glb [--out-dir DIR] [--background]"""
import argparse
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_rgb8(path, array):
    import numpy as np
    a = np.ascontiguousarray(array, np.uint8)
    with open(path, "wb") as f:
        f.write(b"RT8I" + struct.pack("<3i", a.shape[1], a.shape[0], a.shape[2]))
        f.write(a.tobytes())


def extract(model, out_dir=None, background=False):
    """Returns the path prefix the C loader must be given (a copy / link of the model next to its side files)."""
    from raytracing_c_amd.background import procedural_background
    from raytracing_c_amd.loaders import load_model_data
    data = load_model_data(model)
    prefix = model
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        prefix = os.path.join(out_dir, os.path.basename(model))
        if not os.path.exists(prefix):
            os.symlink(os.path.abspath(model), prefix)
        base, ext = os.path.splitext(model)
        for side in ((base + ".mtl",) if ext.lower() == ".obj" else ()):        # the .mtl travels with an .obj
            dst = os.path.join(out_dir, os.path.basename(side))
            if os.path.exists(side) and not os.path.exists(dst):
                os.symlink(os.path.abspath(side), dst)
    for k, im in enumerate(data["images"]):
        write_rgb8(f"{prefix}.image{k}.rgb8", im)
    if background:
        write_rgb8(f"{prefix}.background.rgb8", procedural_background())
    return prefix


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("--out-dir", default=None)
    ap.add_argument("--background", action="store_true")
    a = ap.parse_args()
    print(extract(a.model, a.out_dir, a.background))

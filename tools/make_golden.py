#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- committed input/output vectors of the render path.

The reference has no tests and cannot be built or imported here (C sources that need the absent
codin library), so these vectors are produced by the CPU oracle (oracle/oracle.c) AFTER it has been
pinned by tests/test_oracle_kat.py; they freeze its behaviour so that both the oracle and the HIP
path are checked against committed data, not only against each other.

    python tools/make_golden.py          # rewrites tests/golden/
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracing_c_amd.configs import load_config          # noqa: E402
from tests import _oracle                                 # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

FRAMES = [  # name, config, shader, width, height, samples, bounces, seed
    ("quad_64", "quad", "disney", 64, 64, 16, 4, 0x1234ABCD),
    ("spheres_64", "spheres", "disney", 64, 64, 8, 4, 0x1234ABCD),
    ("spheres_debug_48", "spheres", "debug", 48, 48, 4, 2, 0x1234ABCD),
    ("helmet_64x36", "helmet", "disney", 64, 36, 4, 8, 0x1234ABCD),
    ("tower_64x36", "tower", "disney", 64, 36, 4, 12, 7),
    ("helmet_ragged_45x31", "helmet", "disney", 45, 31, 3, 5, 99),
]


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    lib = _oracle.load()
    for name, cfgname, shader, w, h, s, b, seed in FRAMES:
        hs, _ = load_config(cfgname, shader=shader)
        r = _oracle.render(hs, w, h, s, b, seed=seed, n_threads=8)
        c = r["counters"]
        np.savez_compressed(
            os.path.join(GOLDEN, name + ".npz"),
            config=np.array([cfgname, shader]), params=np.array([w, h, s, b, seed], np.int64),
            image=r["image"], linear=r["linear"],
            accum_sha256=np.array(hashlib.sha256(r["accum"].tobytes()).hexdigest()),
            accum_sum=r["accum"].sum(axis=(0, 1), dtype=np.uint64),
            counters=np.array([c[k] for k in ("paths", "rays", "node_visits", "leaf_visits", "shades",
                                               "backgrounds", "textured")], np.int64))
        print(name, c)

    # unit-level vectors
    rng = np.random.default_rng(2024)
    n = 256
    base = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    v = rng.normal(size=(n, 3))
    v[:, 2] = np.abs(v[:, 2]) + 0.05
    in_dir = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    par = np.stack([rng.uniform(0.001, 1, n), rng.uniform(0, 1, n), rng.choice([0, 0.5], n), rng.uniform(0, 1, n),
                    rng.choice([0, 0.25], n)], 1).astype(np.float32)     # roughness metal sheen tint aniso2
    states = rng.integers(1, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    out_dir, brdf, st_out = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.uint32)
    for i in range(n):
        st = C.c_uint32(int(states[i]))
        lib.oracle_sample_disney_brdf(*[C.c_float(float(x)) for x in par[i]], base[i].ctypes.data, in_dir[i].ctypes.data,
                                      C.byref(st), out_dir[i].ctypes.data, brdf[i].ctypes.data)
        st_out[i] = st.value
    rand = np.zeros(16, np.uint32)
    lib.oracle_rand_u32_seq(1, 16, rand.ctypes.data)
    mx = rng.uniform(0, 1, 512).astype(np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "unit_vectors.npz"),
                        brdf_base=base, brdf_in_dir=in_dir, brdf_params=par, brdf_state_in=states,
                        brdf_out_dir=out_dir, brdf_value=brdf, brdf_state_out=st_out,
                        rand_u32_from_1=rand,
                        math_x=mx, srgb_to_linear=_oracle.math(7, mx), linear_to_srgb=_oracle.math(8, mx),
                        sin=_oracle.math(3, mx * np.float32(6.28)), cos=_oracle.math(4, mx * np.float32(6.28)))
    print("unit_vectors written")


if __name__ == "__main__":
    main()

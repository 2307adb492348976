"""Numeric contract v1 on the GPU: librt_hip_v1.so -- the product's sources compiled with -DRT_MATH_NO_FMA (make v1) -- still
renders round 3's committed fixtures (tests/golden/v1/) bit for bit: radiance sums and counters.  Together with
tests/test_oracle_contracts.py (liboracle_v1.so reproduces the same fixtures) this keeps the OLD contract alive on both sides,
as the A/B partner of the explicit-FMA contract the product ships (include/rt_math.h).  Own process (RT_LIB_PATH)."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V1 = os.path.join(ROOT, "raytracing_c_amd", "librt_hip_v1.so")
FRAMES = sorted(f for f in glob.glob(os.path.join(ROOT, "tests", "golden", "v1", "*.npz")) if not os.path.basename(f).startswith("unit_vectors"))


def test_v1_library_reproduces_round3_fixtures():
    if not os.path.exists(V1):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracing_c_amd", "csrc"), "v1"], stdout=subprocess.DEVNULL)
    jobs, want = [], []
    for f in FRAMES:
        g = np.load(f)
        cfgname, shader = [str(x) for x in g["config"]]
        w, h, s, b, seed = [int(x) for x in g["params"]]
        jobs.append(dict(config=cfgname, shader=shader, w=w, h=h, s=s, b=b, seed=seed, env={}, slabs=[0]))
        c = g["counters"].tolist()                        # paths rays nodes leaves shades backgrounds textured
        want.append((str(g["accum_sha256"]), [c[1], c[2], c[3], c[4]]))
    assert len(jobs) == 6
    env = dict(os.environ, RT_LIB_PATH=V1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_diag_worker.py")], input=json.dumps(jobs), text=True,
                       capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == len(jobs), (r.stdout[-1000:], r.stderr[-1000:])
    for job, got, (digest, counters) in zip(jobs, lines, want):
        assert got["error"] is None, (job, got["error"])
        assert got["contract"] == 1
        assert got["digests"] == [digest], job
        assert got["counters"] == counters, job


def test_product_library_is_contract_v2():
    import raytracing_c_amd as rt
    assert rt.lib.rt_math_contract() == 2

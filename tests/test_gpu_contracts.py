"""The earlier numeric contracts on the GPU: the product's sources compiled with -DRT_MATH_NO_FMA (`make v1`, round 3's
arithmetic) and with -DRT_MATH_V2 (`make v2`, round 4's: explicit FMA, sRGB decode through exp(2.4 log x)) still render the
fixtures committed under THEIR contract (tests/golden/v1/, tests/golden/v2/) bit for bit: radiance sums and counters.  Together
with tests/test_oracle_contracts.py (liboracle_v1.so / liboracle_v2.so reproduce the same fixtures) this keeps the old contracts
alive on both sides, as the A/B partners of the contract the product ships (include/rt_math.h: v3).  Own process (RT_LIB_PATH)."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("version", [1, 2])
def test_old_contract_library_reproduces_its_fixtures(version):
    lib = os.path.join(ROOT, "raytracing_c_amd", f"librt_hip_v{version}.so")
    frames = sorted(f for f in glob.glob(os.path.join(ROOT, "tests", "golden", f"v{version}", "*.npz"))
                    if not os.path.basename(f).startswith("unit_vectors"))
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracing_c_amd", "csrc"), f"v{version}"], stdout=subprocess.DEVNULL)
    jobs, want = [], []
    for f in frames:
        g = np.load(f)
        cfgname, shader = [str(x) for x in g["config"]]
        w, h, s, b, seed = [int(x) for x in g["params"]]
        jobs.append(dict(config=cfgname, shader=shader, w=w, h=h, s=s, b=b, seed=seed, env={}, slabs=[0]))
        c = g["counters"].tolist()                        # paths rays nodes leaves shades backgrounds textured
        want.append((str(g["accum_sha256"]), [c[1], c[2], c[3], c[4]]))
    assert len(jobs) == 6
    env = dict(os.environ, RT_LIB_PATH=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_diag_worker.py")], input=json.dumps(jobs), text=True,
                       capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == len(jobs), (r.stdout[-1000:], r.stderr[-1000:])
    for job, got, (digest, counters) in zip(jobs, lines, want):
        assert got["error"] is None, (job, got["error"])
        assert got["contract"] == version
        assert got["digests"] == [digest], job
        assert got["counters"] == counters, job


def test_product_library_is_contract_v3():
    import raytracing_c_amd as rt
    assert rt.lib.rt_math_contract() == 3

"""The diagnostic build (librt_hip_diag.so, `make -C raytracing_c_amd/csrc diag`): the superseded kernel generations
(RT_KERNEL=1..4) and the RT_* scheduling knobs exist ONLY there -- the product library librt_hip.so has one path kernel and
reads no such variable -- and every one of them must still give the oracle's radiance sums and counters bit for bit.
The diagnostic library is loaded in a process of its own (tests/_diag_worker.py, RT_LIB_PATH)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "raytracing_c_amd", "librt_hip_diag.so")

VARIANT_JOBS = [dict(config=name, w=w, h=h, s=s, b=b, env={"RT_KERNEL": str(v)}, slabs=[0])
                for v in (1, 2, 3, 5) for (name, w, h, s, b) in (("helmet", 96, 54, 6, 8), ("quad", 40, 40, 4, 3))]
KNOBS = [{"RT_SCHED_THRESH": "1"}, {"RT_SCHED_THRESH": "64"}, {"RT_LDS_NODES": "9"}, {"RT_LDS_NODES": "0"}, {"RT_WAVES_PER_CU": "1"},
         {"RT_ORDER": "identity"}, {"RT_GRAB": "1"}, {"RT_GRAB": "4"}, {"RT_DRAIN_THRESH": "1"}, {"RT_PYRAMID": "0"},
         {"RT_SHORT_DIV": "0"}, {"RT_PARK": "0"}, {"RT_KERNEL": "3", "RT_SCHED_THRESH": "16"},
         {"RT_WG_WAVES": "8"}, {"RT_WG_WAVES": "12"}, {"RT_WG_WAVES": "16"},      # the three workgroup sizes the product picks from by launch size
         {"RT_PIPELINE": "wf"}, {"RT_PIPELINE": "wf", "RT_WF_GEOMETRY": "1"}, {"RT_PIPELINE": "wf", "RT_WF_GEOMETRY": "2", "RT_LDS_NODES": "40"}]
KNOB_JOBS = [dict(config="helmet", w=80, h=45, s=5, b=8, env=k, slabs=[0, 1, 4, 64]) for k in KNOBS]


@pytest.fixture(scope="module")
def diag_results():
    if not os.path.exists(DIAG):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracing_c_amd", "csrc"), "diag"], stdout=subprocess.DEVNULL)
    jobs = VARIANT_JOBS + KNOB_JOBS
    env = dict(os.environ, RT_LIB_PATH=DIAG)
    for k in list(env):
        if k.startswith("RT_") and k != "RT_LIB_PATH":
            del env[k]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_diag_worker.py")], input=json.dumps(jobs), text=True,
                       capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == len(jobs), (r.stdout[-1000:], r.stderr[-1000:])
    return dict(zip([json.dumps(j, sort_keys=True) for j in jobs], lines))


def _want(job):
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    hs, _ = load_config(job["config"])
    want = _oracle.render(hs, job["w"], job["h"], job["s"], job["b"])
    return hashlib.sha256(want["accum"].tobytes()).hexdigest(), [want["counters"][k] for k in ("rays", "node_visits", "leaf_visits", "shades")]


@pytest.mark.parametrize("job", VARIANT_JOBS, ids=[f"k{j['env']['RT_KERNEL']}-{j['config']}" for j in VARIANT_JOBS])
def test_every_kernel_generation_is_bit_exact(oracle, diag_results, job):
    """RT_KERNEL=1 plain while-while kernel, 2 phase-scheduled, 3 phase-scheduled + BVH top in LDS, 5 tile streams (the product's)."""
    got = diag_results[json.dumps(job, sort_keys=True)]
    assert got["error"] is None, got["error"]
    digest, counters = _want(job)
    assert got["digests"] == [digest]
    assert got["counters"] == counters


@pytest.mark.parametrize("job", KNOB_JOBS, ids=["-".join(f"{k[3:]}={v}" for k, v in j["env"].items()) for j in KNOB_JOBS])
def test_scheduling_knobs_do_not_change_the_image(oracle, diag_results, job):
    """Scheduling is free to change; results are not (order-free fixed-point accumulation)."""
    got = diag_results[json.dumps(job, sort_keys=True)]
    assert got["error"] is None, got["error"]
    digest, _ = _want(job)
    assert got["digests"] == [digest] * len(job["slabs"])


def test_product_library_ignores_the_experiment_knobs(oracle, monkeypatch):
    """librt_hip.so has ONE path kernel and no environment switch: RT_KERNEL / RT_PIPELINE / RT_PARK in the host application's
    environment change neither the kernel that runs (rays are counted by the tile-stream kernel's own counters) nor a pixel."""
    import ctypes as C
    import numpy as np
    import raytracing_c_amd as rt
    from raytracing_c_amd.configs import load_config
    from tests import _oracle
    assert rt.lib.rt_init(0) == 0, rt.last_error()
    raw = open(rt.native.LIB_PATH, "rb").read()
    for name in (b"RT_KERNEL", b"RT_SCHED_THRESH", b"RT_PARK", b"RT_PIPELINE", b"RT_WAVES_PER_CU", b"rt_path_kernel_sched"):
        assert name not in raw, name
    hs, _ = load_config("helmet")
    want = _oracle.render(hs, 64, 36, 4, 8)
    for k, v in (("RT_KERNEL", "1"), ("RT_PARK", "0"), ("RT_PIPELINE", "wf"), ("RT_SCHED_THRESH", "1")):
        monkeypatch.setenv(k, v)
    got = rt.render_frame(hs, 64, 36, 4, 8, want_accum=True)
    assert np.array_equal(got["accum"], want["accum"])

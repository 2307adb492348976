"""bench.py started WITHOUT a launcher must start its own ranks (`python bench.py --gpus N`, the form the driver uses):
the parent spawns `python -m torch.distributed.run ...` as a child before it imports torch or touches a GPU, relays
rank 0's single JSON line and exits with the children's status.  `--dry-run` keeps the children off the GPU (gloo
rendezvous, partition tables, the tile gather), so the whole launcher branch runs on a CPU-only machine."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=300, env=e, cwd=ROOT)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_bench_starts_its_own_ranks(n):
    r = _run("--gpus", str(n), "--dry-run", "--width", "200", "--height", "120")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly ONE line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["dry_run"] is True
    assert sum(out["chunks_per_rank"]) == 7 * 4 and len(out["chunks_per_rank"]) == n


def test_bench_launcher_relays_failure_status():
    # the children fail (an option the ranks reject): the parent must not print a result line and must not exit 0
    r = _run("--gpus", "2", "--dry-run", "--config-does-not-exist")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_stopping_the_launcher_stops_every_rank():
    """the parent is killed (a `timeout`, the driver's limit, Ctrl-C) exactly when a rank hangs.  The ranks run in
    a process group of their own; SIGTERM to the parent must take torchrun and every rank down with it."""
    import re
    import signal
    import time
    e = dict(os.environ, RT_BENCH_KILL_GRACE="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--dry-run-sleep", "120"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, cwd=ROOT)
    pids, t0 = [], time.time()
    os.set_blocking(p.stderr.fileno(), False)
    buf = ""
    while len(pids) < 2 and time.time() - t0 < 120:
        try:
            buf += os.read(p.stderr.fileno(), 65536).decode("utf-8", "replace")
        except BlockingIOError:
            pass
        pids = [int(m) for m in re.findall(r"rank \d+ pid (\d+) sleeping", buf)]
        if p.poll() is not None:
            break
        time.sleep(0.1)
    assert len(pids) == 2, buf[-2000:]
    p.send_signal(signal.SIGTERM)
    try:
        rc = p.wait(timeout=30)
    finally:
        if p.poll() is None:
            p.kill()
    assert rc != 0

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except PermissionError:
            return True
        try:                                           # a zombie waiting for its (dead) parent's reaper is not a survivor
            return open(f"/proc/{pid}/stat").read().split(")")[1].split()[0] != "Z"
        except OSError:
            return False

    t0 = time.time()
    while any(alive(x) for x in pids) and time.time() - t0 < 10:
        time.sleep(0.1)
    assert not [x for x in pids if alive(x)], "ranks survived the launcher"


@pytest.mark.gpu
def test_bench_distributed_pipeline_on_one_gpu(tmp_path):
    """The N > 1 frame pipeline of bench.py -- path kernels of consecutive frames on two streams with a device scene each,
    RCCL gather of the tiles on a third, untile, D2H -- rehearsed with ONE rank (RT_BENCH_FORCE_DIST=1): the frame it
    delivers is the frame of the plain single-GPU path, byte for byte, with and without the overlap."""
    common = ("--steps", "3", "--warmup", "2", "--samples", "24", "--no-cpu-baseline", "--no-bvh-compare")
    imgs = []
    for name, env in (("plain", {}), ("dist", {"RT_BENCH_FORCE_DIST": "1"}),
                      ("dist_one_stream", {"RT_BENCH_FORCE_DIST": "1", "RT_BENCH_OVERLAP": "0"})):
        png = str(tmp_path / f"{name}.png")
        r = _run(*common, "--save", png, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["n_gpus"] == 1 and out["value"] > 0
        if env:
            assert out["distributed"]["world_size"] == 1 and out["distributed"]["backend"] == "nccl" and out["distributed"]["equal"]
        imgs.append(open(png, "rb").read())
    assert imgs[0] == imgs[1] == imgs[2]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_deliver_the_same_frame(tmp_path):
    """`python bench.py --gpus 2` (the driver's form: the parent starts the ranks) with both ranks on GPU 0 and the tiles
    exchanged over gloo -- what a one-GPU box can run of the N > 1 path: lattice partition, two frames in flight, gather
    to rank 0, untile.  Rank 0's frame equals the single-GPU frame byte for byte."""
    common = ("--steps", "3", "--warmup", "1", "--samples", "24", "--no-cpu-baseline", "--no-bvh-compare")
    one, two = str(tmp_path / "one.png"), str(tmp_path / "two.png")
    r = _run(*common, "--save", one)
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run("--gpus", "2", *common, "--save", two, env={"RT_BENCH_BACKEND": "gloo", "RT_BENCH_DEVICE": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "two streams" in out["config"]["partition"]
    assert open(one, "rb").read() == open(two, "rb").read()
    d = out["distributed"]                               # bench.py's own check of the gathered frame against a one-rank render
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["equal"] and d["image_sha256"] == d["one_rank_image_sha256"]


def _physical_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
@pytest.mark.skipif("_physical_gpus() < 2", reason="needs two physical GPUs")
def test_bench_two_ranks_on_two_gpus_over_rccl(tmp_path):
    """Switches itself on when the box has two GPUs: `python bench.py --gpus 2` as the driver runs it -- one
    rank per GPU, the tile gather over RCCL / xGMI.  The line must say world 2 / nccl and carry the hash check; rank 0's frame is
    the single-GPU frame byte for byte."""
    common = ("--steps", "3", "--warmup", "1", "--samples", "24", "--no-cpu-baseline", "--no-bvh-compare")
    one, two = str(tmp_path / "one.png"), str(tmp_path / "two.png")
    r = _run(*common, "--save", one)
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run("--gpus", "2", *common, "--save", two)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    d = out["distributed"]
    assert out["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "nccl" and d["equal"]
    assert open(one, "rb").read() == open(two, "rb").read()

"""AddressSanitizer + UndefinedBehaviorSanitizer and ThreadSanitizer over the host-side C code (CPU build only; the GPU pool has no
sanitizer support): scene_init with its builder threads, the .scene file functions, and the oracle's loops."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
@pytest.mark.parametrize("sanitizers", [["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                                        ["-fsanitize=thread"]], ids=["asan+ubsan", "tsan"])
def test_host_c_code_is_clean_under_sanitizers(tmp_path, sanitizers):
    exe = str(tmp_path / "sanitize_host")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fno-omit-frame-pointer", *sanitizers, "-ffp-contract=off", "-march=x86-64-v3",
           os.path.join(ROOT, "tests", "c", "sanitize_host.c"),
           os.path.join(ROOT, "raytracing_c_amd", "csrc", "rt_scene_build.c"),
           os.path.join(ROOT, "oracle", "oracle.c"), "-o", exe, "-lpthread", "-lm"]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") == 8, r.stdout
    assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, r.stderr

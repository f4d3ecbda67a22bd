"""The host-only C++ of the library (scene flattening, SAH BVH build + 8-wide collapse, light grid, meshletizer, view
constants) under AddressSanitizer + UndefinedBehaviorSanitizer.  GPU sanitizers are not available on the MI355X pool;
this is the CPU build the task statement asks sanitizers to run on."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_ubsan():
    san = os.path.join(ROOT, "tools", "sanitize")
    subprocess.run(["make", "-C", san, "-s"], check=True)
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found next to g++")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(san, "run_host_san.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "host sanitizer run clean" in r.stdout and "host sanitizer fuzz clean" in r.stdout

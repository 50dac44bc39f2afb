"""The host-side planning arithmetic of libgsx (csrc/gsx_plan.h: workspace carving, pair capacity, tile grid and
window, the rectangles a frame zeroes) compiled with g++ under AddressSanitizer + UndefinedBehaviorSanitizer and
swept to the 2^31 limits (tests/host/plan_sanitize.cpp).  GPU sanitizers are not available on the MI355X pool;
this code is pure host integer arithmetic, so the CPU build covers it completely."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_plan_arithmetic_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "plan_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "intro_to_gaussian_splatting_amd", "csrc"),
           os.path.join(ROOT, "tests", "host", "plan_sanitize.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert run.stdout.startswith("ok:")

"""Host-side C++ (pair dictionary training/encoding, observation file IO) under AddressSanitizer + UBSan.
GPU sanitizers are not available on this pool, so the CPU build of the host logic is what gets sanitized:
tests/host_sanitizer.cpp trains dictionaries on random streams, encodes every level, decodes them back and
round-trips the packed cache format."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_code_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = tmp_path / "host_sanitizer"
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                            "-o", str(exe), os.path.join(HERE, "host_sanitizer.cpp")], capture_output=True, text=True, cwd=HERE)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert run.returncode == 0 and "sanitizer run ok" in run.stdout, (run.stdout + run.stderr)[-2000:]

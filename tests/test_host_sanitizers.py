"""AddressSanitizer + UBSan over the host-side builders (GPU sanitizers are not available on the pool;
the device code is covered by the parity tests).  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_builders_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "edipack_amd", "csrc"), "-o", exe,
           os.path.join(ROOT, "tests", "host_sanitize_main.cpp"),
           os.path.join(ROOT, "edipack_amd", "csrc", "host_build.cpp"),
           os.path.join(ROOT, "edipack_amd", "csrc", "host_ib.cpp")]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "0 failures" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr

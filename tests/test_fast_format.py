"""metamaps_amd/csrc/host/fast_format.hpp (the printf-free %g / %f of the host program's output files) against snprintf: tests/test_fast_format.cpp.  CPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fast_format_equals_printf(tmp_path):
    exe = str(tmp_path / "tff")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "test_fast_format.cpp")], check=True, timeout=300)
    p = subprocess.run([exe, "400000"], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.decode().startswith("ok "), p.stdout.decode()[-500:]

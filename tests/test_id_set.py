"""metamaps_amd/csrc/host/id_set.hpp (the read IDs `mapDirectly` has handled) against std::set<std::string>: tests/test_id_set.cpp.  CPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_id_set_equals_std_set(tmp_path):
    exe = str(tmp_path / "tis")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "test_id_set.cpp")], check=True, timeout=300)
    p = subprocess.run([exe, "300000"], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.decode().startswith("ok "), p.stdout.decode()[-500:]

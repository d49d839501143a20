"""metamaps_amd/csrc/host/huge_new.hpp (the host program's operator new: blocks from 4 MiB on on transparent huge pages): tests/test_huge_new.cpp.  CPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_huge_new_blocks(tmp_path):
    exe = str(tmp_path / "thn")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "test_huge_new.cpp"), "-lpthread"], check=True, timeout=300)
    p = subprocess.run([exe], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.decode().startswith("ok"), p.stdout.decode()[-500:]

"""metamaps_amd/csrc/mm_slab.hpp (pieces of pooled device blocks: the bookkeeping, no device) under a random alloc / free load: tests/test_slab.cpp.  CPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_slab_pieces_never_overlap_and_merge_back(tmp_path):
    exe = str(tmp_path / "tslab")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "test_slab.cpp"), "-lpthread"], check=True, timeout=300)
    p = subprocess.run([exe, "200000"], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.decode().startswith("ok "), p.stdout.decode()[-500:]

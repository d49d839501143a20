"""metamaps_amd/csrc/host_pack.cpp (AVX2 packing of bases for mm_seqset_upload): tests/test_host_pack.cpp.  CPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_simd_pack_equals_the_bytewise_definition(tmp_path):
    exe = str(tmp_path / "thp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "test_host_pack.cpp")], check=True, timeout=300)
    p = subprocess.run([exe], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.decode().startswith("ok"), p.stdout.decode()[-500:]

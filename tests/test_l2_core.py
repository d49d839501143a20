"""Builds and runs tests/test_l2_core.cpp: the K5 integer state machine vs the oracle's ordered-map window."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_l2_state_machine_matches_ordered_map(tmp_path):
    exe = str(tmp_path / "t")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "test_l2_core.cpp"), "-lz"], check=True, timeout=300)
    p = subprocess.run([exe, "300"], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stdout.decode()
    assert b"ok" in p.stdout

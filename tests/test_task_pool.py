"""metamaps_amd/csrc/task_pool.hpp (the helper threads that format a batch's text): tests/test_task_pool.cpp.  CPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_task_pool_runs_every_task_once(tmp_path):
    exe = str(tmp_path / "ttp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "test_task_pool.cpp"), "-lpthread"], check=True, timeout=300)
    p = subprocess.run([exe], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.decode().startswith("ok"), p.stdout.decode()[-500:]

"""The host program stays readable: no function of metamaps_amd/csrc/host/ above 200 lines (round-4 review: map_mode was one 797-line function, classify_one 280).  CPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_function_above_200_lines_in_the_host_program():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "func_lengths.py")], capture_output=True, timeout=120)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    longest = int(out.split()[0])
    assert 40 < longest <= 200, out                                # (40 <: the script did find the functions)

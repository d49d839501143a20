"""The MM_* environment switches are listed in ONE table (metamaps_amd/csrc/mm_env.hpp) that is complete — every getenv("MM_...") of the library and the
CLI is in it, nothing in it is dead — and INTEGRATION.md carries that table.  CPU."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import env_table  # noqa: E402


def _read_switches():
    names = set()
    for pat in ("metamaps_amd/csrc/*.hip", "metamaps_amd/csrc/*.hpp", "metamaps_amd/csrc/host/*.cpp", "metamaps_amd/csrc/host/*.hpp"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            names |= set(re.findall(r'getenv\("(MM_[A-Z0-9_]+)"\)', open(f).read()))
    return names


def test_every_switch_the_sources_read_is_in_the_table_and_back():
    table = [r[0] for r in env_table.rows()]
    assert len(table) == len(set(table)) and len(table) > 60
    read = _read_switches()
    assert read - set(table) == set(), sorted(read - set(table))
    assert set(table) - read == set(), sorted(set(table) - read)   # a row nothing reads any more is stale documentation
    kinds = {r[2] for r in env_table.rows()}
    assert kinds == {"user", "tuning", "test", "debug"}


def test_integration_md_carries_the_table():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for line in env_table.markdown().splitlines():
        assert line in doc, line

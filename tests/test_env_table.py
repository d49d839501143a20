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


def test_switches_of_the_python_and_shell_side_are_known_or_exempt():
    """Everything bench.py, the tests and tools/ set or read (os.environ / getenv / VAR=... on a command line) is either in the table or
    carries one of the prefixes env_unknown() exempts — so MM_STRICT_ENV=1 can be combined with a fuzz campaign or an A/B run."""
    src = open(os.path.join(ROOT, "metamaps_amd", "csrc", "mm_env.hpp")).read()
    exempt = re.findall(r'"(MM_[A-Z_]+)"', src[src.index("static const char* const outside[]"):src.index("bool ours = false;")])
    assert "MM_LIB_PATH" in exempt and "MM_FUZZ_" in exempt
    table = {r[0] for r in env_table.rows()}
    seen = set()
    for pat in ("*.py", "tests/*.py", "tools/*.py", "tools/*.sh", "metamaps_amd/*.py"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            seen |= set(re.findall(r'\b(MM_[A-Z][A-Z0-9_]+)\b', open(f).read()))
    seen = {n for n in seen if not n.startswith(("MM_ERR_", "MM_OK", "MM_HIP", "MM_REQUIRE", "MM_KERNEL", "MM_HD"))}   # constants of the C ABI quoted in Python, macros
    unknown = {n for n in seen if n not in table and not any(n.startswith(e) for e in exempt)}
    # names that only occur as deliberately misspelt / historical examples in tests and docs
    unknown -= {"MM_L2_FUL", "MM_L2_FUSE"}
    assert unknown == set(), sorted(unknown)

"""Prints the MM_* switch table of metamaps_amd/csrc/mm_env.hpp as the markdown table INTEGRATION.md carries (tests/test_env_table.py holds the two together)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows():
    src = open(os.path.join(ROOT, "metamaps_amd", "csrc", "mm_env.hpp")).read()
    return re.findall(r'\{"(MM_[A-Z0-9_]+)",\s*"((?:[^"\\]|\\.)*)",\s*"(user|tuning|test|debug)",\s*"((?:[^"\\]|\\.)*)"\}', src)


def markdown():
    out = ["| switch | default | kind | effect |", "|---|---|---|---|"]
    for name, dflt, kind, what in rows():
        out.append(f"| `{name}` | {dflt} | {kind} | {what.replace(chr(92) + chr(34), chr(34))} |")
    return "\n".join(out)


if __name__ == "__main__":
    print(markdown())

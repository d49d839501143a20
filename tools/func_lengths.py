"""Longest functions of the host program (round-4 review item 8: no function above 200 lines in metamaps_amd/csrc/host/).  A function = a brace block whose
opening line ends in ') {' or ') const {' (or carries a trailing comment behind that) at nesting depth <= 1 (top level, or directly inside a struct / namespace)."""
import glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def functions(path):
    lines = open(path).read().split("\n")
    depth, stack, out = 0, [], []
    for i, ln in enumerate(lines):
        code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', ln)
        code = re.sub(r"'(?:[^'\\]|\\.)'", "''", code).split("//")[0]
        opens, closes = code.count("{"), code.count("}")
        if opens > closes and re.search(r"\)\s*(const\s*)?(noexcept\s*)?(->\s*[\w:<>]+\s*)?\{\s*$", code.rstrip()) and not re.match(r"\s*(if|for|while|switch|else|do)\b", code) and "[&" not in code and "[=" not in code and "[this" not in code:
            stack.append((depth, i, ln.strip()[:90]))
        depth += opens - closes
        while stack and depth <= stack[-1][0]:
            d0, i0, name = stack.pop()
            out.append((i - i0 + 1, os.path.relpath(path, ROOT), i0 + 1, name))
    return out


if __name__ == "__main__":
    allf = []
    for f in glob.glob(os.path.join(ROOT, "metamaps_amd", "csrc", "host", "*")):
        allf += functions(f)
    allf.sort(reverse=True)
    for n, f, l, name in allf[:12]:
        print(f"{n:5d}  {f}:{l}  {name}")
    sys.exit(1 if allf and allf[0][0] > 200 else 0)

"""The index build of the bench reference in N processes one after the other (MM_HOST_TIMING=1: the build's sections on stderr), to see what a
process pays for device memory that an earlier process has used: the first child of a fresh box gets memory nobody has written, the later ones get
what the child before gave back.  `--pause S` sleeps between children, `--env KEY=VAL` (repeatable) goes to every child, `--inproc R` builds R
times inside each child (destroying the index in between: the pool's regime).

    python tools/index_build_repeat.py --children 3 [--pause 0] [--scale 1.0] [--inproc 1]
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, time
sys.path.insert(0, %(root)r)
import numpy as np
from metamaps_amd import capi
scale, inproc = %(scale)r, %(inproc)d
t0 = time.time()
ctx = capi.Context(0)
ng, sp, ge = max(4, int(12000 * scale)), max(2, int(3000 * scale)), max(1, int(600 * scale))
human = max(1, int(round(24 * min(scale, 1.0)))) if scale >= 0.04 else 0
ref, genome = ctx.synth_community(seed=20260928, n_genomes=ng, n_species=sp, n_genera=ge, median_len=2.0e6, sigma_len=0.6, min_len=5_000, max_len=12_000_000,
                                  strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
                                  human_contigs=human, human_bases=int(3.1e9 * min(scale, 1.0)), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000,
                                  total_bases_target=int(26_762_276_280 * scale))
t1 = time.time()
for r in range(inproc):
    t2 = time.time()
    idx = ctx.index(ref, 16, 8)
    print(f"CHILD build {r}: {time.time() - t2:.3f} s (context + reference {t1 - t0:.3f} s)", file=sys.stderr, flush=True)
    if r + 1 < inproc:
        idx.close()
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--children", type=int, default=3)
    ap.add_argument("--pause", type=float, default=0.0)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--inproc", type=int, default=1)
    ap.add_argument("--env", action="append", default=[])
    a = ap.parse_args()
    env = dict(os.environ, MM_HOST_TIMING="1")
    for kv in a.env:
        k, _, v = kv.partition("=")
        env[k] = v
    code = CHILD % {"root": ROOT, "scale": a.scale, "inproc": a.inproc}
    for c in range(a.children):
        t = time.time()
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        print(f"--- child {c} (wall {time.time() - t:.2f} s, rc {p.returncode}) env {a.env}")
        for ln in p.stderr.splitlines():
            if "index build:" in ln or ln.startswith("CHILD") or p.returncode or ("MM_ALLOC_TRACE" in ln and "slab piece" not in ln):
                print("   " + ln)
        sys.stdout.flush()
        if a.pause:
            time.sleep(a.pause)


if __name__ == "__main__":
    main()

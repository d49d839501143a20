"""where the first mm_map_batch of a context spends its time: per process (code objects) or per context (allocations)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi

def t(f):
    t0 = time.perf_counter(); r = f(); return r, (time.perf_counter() - t0) * 1e3

rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
contigs = [rng.choice(acgt, size=1_000_000).tobytes() for _ in range(20)]
nr = int(os.environ.get("NR", "6000"))
reads = []
for r in range(nr):
    g = int(rng.integers(20)); p = int(rng.integers(0, 990_000)); reads.append(contigs[g][p:p + 10_000])
k, w = 16, 8
A, ta = t(lambda: capi.Context(0)); print(f"context A {ta:.1f} ms")
S = A.seqset(contigs); idx, ti = t(lambda: A.index(S, k, w)); print(f"index {ti:.1f} ms")
for name in ("A", "A", "B", "B", "C(small first)", "C", "C"):
    if name == "B" and "B" not in globals(): B, tb = t(lambda: capi.Context(0)); print(f"context B {tb:.1f} ms")
    if name.startswith("C") and "C" not in globals(): C, tc = t(lambda: capi.Context(0)); print(f"context C {tc:.1f} ms")
    ctx = {"A": A, "B": globals().get("B"), "C": globals().get("C")}[name[0]]
    rd = reads[:50] if "small" in name else reads
    R, tu = t(lambda: ctx.seqset(rd))
    M, tm = t(lambda: ctx.map_batch(idx, R, k, w))
    _, tq = t(lambda: M.add_qualities(k))
    _, tf = t(lambda: M.fetch())
    print(f"ctx {name:15s} upload {tu:7.1f}  map {tm:7.1f}  mapq {tq:6.1f}  fetch {tf:6.1f} ms")

"""where the wall time of the whole-reference index build goes: MM_ALLOC_TRACE lines (driver allocations with their cost) summed, against the wall clock"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MM_ALLOC_TRACE"] = "1"
import bench
from metamaps_amd import capi
sys.argv = ["bench.py"]
args = bench.parse_args()
ctx = capi.Context(0)
ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
ctx.synchronize()
print("BUILD BEGIN", file=sys.stderr, flush=True)
t0 = time.time()
idx = ctx.index(ref, 16, 8)
ctx.synchronize()
print(f"BUILD END wall {time.time() - t0:.3f} s", file=sys.stderr, flush=True)

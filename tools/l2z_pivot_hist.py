"""Where the pivot of a candidate's best window lies against the zone kernel's estimate (MM_L2Z_DBG: mm_l2z.hpp), and how many zone passes candidates take:
   python tools/l2z_pivot_hist.py LO HI N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metamaps_amd import capi

lo, hi, n = (int(x) for x in (sys.argv[1:4] + ["10000", "10000", "20000"][len(sys.argv) - 1:]))
sys.argv = ["bench.py"]
args = bench.parse_args()
ctx = capi.Context(0)
ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
idx = ctx.index(ref, 16, 8)
rd, truth = ctx.synth_reads(ref, seed=77, n_reads=n, read_len=hi, read_len_min=lo, frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
os.environ["MM_L2Z_DBG"] = os.environ.get("DBG", "1")   # 2: against the PREDICTED estimate (seed hits of L1) instead of the one from pass A's matched counts
M = ctx.map_batch(idx, rd, 16, 8)
st = M.stats()
a = M.debug_l2(st["n_candidates"])
acc = a[a[:, 5] == 1]
d = acc[:, 1]
print("accepted", len(acc), "of", len(a), "| pivot - estimate: mean %.1f sd %.1f" % (d.mean(), d.std()), "percentiles 1 5 25 50 75 95 99:", np.percentile(d, [1, 5, 25, 50, 75, 95, 99]))
print("second passes over the stream: ", {int(k): int(v) for k, v in zip(*np.unique(a[:, 2], return_counts=True))})

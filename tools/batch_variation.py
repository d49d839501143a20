"""stage times and work counters of the bench's distinct read batches, one line per batch (why do some steps take 60-100 ms?)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metamaps_amd import capi

sys.argv = ["bench.py"]
args = bench.parse_args()
ctx = capi.Context(0)
ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
idx = ctx.index(ref, 16, 8)
cl = ref.lengths()
n = int(os.environ.get("NB", "24"))
for b in range(n):
    rd, truth = ctx.synth_reads(ref, seed=1000 + 97 * b, n_reads=100_000, read_len=10_000, read_len_min=0, frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
    for rep in range(2):
        M = ctx.map_batch(idx, rd, 16, 8)
        st = M.stats(); M.close()
    t = truth[truth >= 0]
    human = int((contig_taxon[t] == n_taxa - 1).sum())
    big = np.bincount(t, minlength=len(cl)); top = np.argsort(-big)[:3]
    print(f"batch {b:2d}: total {st['ms_total']:6.1f}  K1 {st['ms_minimizer']:5.1f} K3 {st['ms_hit_filter']:5.1f} sort {st['ms_sort_hits']:5.1f} L1 {st['ms_l1_scan']:4.1f} K5 {st['ms_l2']:5.1f} | cands {st['n_candidates']:7d} "
          f"stream {st['sum_l2_stream_entries'] / 1e9:5.2f}e9 evals {st['sum_l2_evals'] / 1e6:6.1f}e6 rebuilds {st['n_l2_rebuilds']:7d} hits {st['sum_hits'] / 1e9:5.2f}e9 kept {st['sum_hits_kept'] / 1e6:6.1f}e6 "
          f"| reads from human-like contigs {human}, top contigs {[(int(c), int(big[c]), int(cl[c])) for c in top]}", flush=True)
    rd.close()

"""Stage times and counters of each of the bench's distinct read batches mapped alone (which batch makes a step of the timed region long?): python tools/batch_probe.py [N_BATCHES]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metamaps_amd import capi

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 15
sys.argv = ["bench.py"]
args = bench.parse_args()
ctx = capi.Context(0)
ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
idx = ctx.index(ref, 16, 8)
for b in range(nb):
    rd, _ = ctx.synth_reads(ref, seed=1000 + 97 * b, n_reads=100000, read_len=10000, read_len_min=0, frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
    best = None
    for it in range(2):
        M = ctx.map_batch(idx, rd, 16, 8, pi=80.0, min_read_len=1000)
        st = M.stats(); M.close()
        if best is None or st["ms_total"] < best["ms_total"]: best = st
    print(f"batch {b}: total {best['ms_total']:.1f} K1 {best['ms_minimizer']:.1f} K2 {best['ms_sketch']:.1f} K3 {best['ms_probe_gather']:.1f} sort {best['ms_sort_hits']:.1f} K5 {best['ms_l2']:.1f} | cands {best['n_candidates']} "
          f"stream {best['sum_l2_stream_entries'] / 1e9:.2f}e9 hits {best['sum_hits'] / 1e9:.2f}e9 kept {best['sum_hits_kept'] / 1e6:.0f}e6 wide_redo {best.get('n_l2_wide_redo')}", flush=True)
    rd.close()

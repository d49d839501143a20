"""K5 kernel time when the zone kernel leaves after a phase (MM_L2_STOP=n: 10 nothing, 1 set-up, 11 the waves' preamble, 2 pass A, 3 e_min + first bounds, 4 pass B + second bounds,
8 first window state, 5 sweep, 0 everything), bench batch: python tools/l2z_stops.py [LO HI N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metamaps_amd import capi

lo, hi, n = (int(x) for x in (sys.argv[1:4] + ["10000", "10000", "100000"][len(sys.argv) - 1:]))
sys.argv = ["bench.py"]
args = bench.parse_args()
ctx = capi.Context(0)
ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
idx = ctx.index(ref, 16, 8)
rd, truth = ctx.synth_reads(ref, seed=77, n_reads=n, read_len=hi, read_len_min=lo, frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
prev = 0.0
for stop in os.environ.get("STOPS", "1 2 3 4 8 5 0").split():
    os.environ["MM_L2_STOP"] = stop
    best = 1e9
    for it in range(3):
        M = ctx.map_batch(idx, rd, 16, 8)
        st = M.stats(); best = min(best, st["ms_l2"]); M.close()
    print(f"stop {stop}: ms_l2 {best:.2f}  (+{best - prev:.2f})" + (f"  evals {st['sum_l2_evals'] / 1e6:.1f}e6 rebuilds {st['n_l2_rebuilds']} stream {st['sum_l2_stream_entries'] / 1e9:.3f}e9" if stop == "0" else ""), flush=True)
    prev = best

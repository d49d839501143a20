"""K5 phase clocks (MM_L2_PHASES) and stage times for reads of one length band against the bench's community index:
   python tools/l2_long_phases.py LO HI N   (defaults 32000 50000 20000)
The zone kernel carries its clocks only in a build with -DL2Z_CLOCKS (`tools/ab_build.sh clocks "-DL2Z_CLOCKS"` -> _ab/clocks/, picked up here when it exists;
without it the zone kernel's clocks read 0 and only the stage times and counts below are meaningful)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("MM_LIB_PATH") and os.path.exists(os.path.join(ROOT, "_ab", "clocks", "libmetamaps_hip.so")):
    os.environ["MM_LIB_PATH"] = os.path.join(ROOT, "_ab", "clocks", "libmetamaps_hip.so")
    print("(library: _ab/clocks/libmetamaps_hip.so, the build with the zone kernel's phase clocks)", flush=True)
import bench
from metamaps_amd import capi

lo, hi, n = (int(x) for x in (sys.argv[1:4] + ["32000", "50000", "20000"][len(sys.argv) - 1:]))
sys.argv = ["bench.py"]
args = bench.parse_args()
ctx = capi.Context(0)
ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
idx = ctx.index(ref, 16, 8)
for band in ((lo, hi),) if len(os.environ.get("BANDS", "")) == 0 else [tuple(int(v) for v in b.split("-")) for b in os.environ["BANDS"].split(",")]:
    rd, truth = ctx.synth_reads(ref, seed=77, n_reads=n, read_len=band[1], read_len_min=band[0], frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
    for rep in range(2):
        if rep == 1: os.environ["MM_L2_PHASES"] = "1"
        M = ctx.map_batch(idx, rd, 16, 8)
        st = M.stats(); M.close()
        os.environ.pop("MM_L2_PHASES", None)
        print(f"band {band} rep {rep}: bases {rd.lengths().sum() / 1e9:.3f}e9 total {st['ms_total']:.1f} K1 {st['ms_minimizer']:.1f} K2 {st['ms_sketch']:.1f} K3 {st['ms_probe_gather']:.1f} (filter {st['ms_hit_filter']:.1f}) sort {st['ms_sort_hits']:.1f} L1 {st['ms_l1_scan']:.1f} "
              f"K5 {st['ms_l2']:.1f} | cands {st['n_candidates']} stream {st['sum_l2_stream_entries'] / 1e9:.2f}e9 evals {st['sum_l2_evals'] / 1e6:.1f}e6 rebuilds {st['n_l2_rebuilds']}", flush=True)
    rd.close()

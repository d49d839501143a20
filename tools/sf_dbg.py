"""Timing aid: cost of the fused probe + filter kernel (K3 + K3c) up to each of its phases (MM_SF_DBG = 1 probe, 2 + offsets, 3 + lists/counting, 4 + bit tests; 0 = all)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi
ctx = capi.Context(0)
ref = ctx.synth_reference(seed=20260928, n_species=3000, strains_per_species=4, genome_len=2_200_000, strain_divergence=0.02, genus_divergence=0.2)
idx = ctx.index(ref, 16, 8)
reads, truth = ctx.synth_reads(ref, seed=1000, n_reads=int(os.environ.get("NR", "100000")), read_len=int(os.environ.get("RL", "10000")), sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
for d in sys.argv[1:]:
    os.environ["MM_SF_DBG"] = d
    best = 1e9
    for it in range(3):
        M = ctx.map_batch(idx, reads, 16, 8)
        st = M.stats()
        best = min(best, st["ms_hit_filter"])
        M.close()
    print("dbg", d, "ms_seed_filter", round(best, 2), "hits", st["sum_hits"], "kept", st["sum_hits_kept"], flush=True)

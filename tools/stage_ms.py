"""Timing aid: stage times of one batch of the bench workload (NR reads of RL bases; SHAPE=uniform|community)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi
ctx = capi.Context(0)
if os.environ.get("SHAPE", "uniform") == "community":
    ref, genome = ctx.synth_community(seed=20260928, n_genomes=12000, n_species=3000, n_genera=600, median_len=2.0e6, sigma_len=0.6, min_len=5000, max_len=12_000_000,
                                      strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
                                      human_contigs=24, human_bases=int(3.1e9), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000, total_bases_target=26_762_276_280)
else:
    ref = ctx.synth_reference(seed=20260928, n_species=int(os.environ.get("NS", "3000")), strains_per_species=4, genome_len=2_200_000, strain_divergence=0.02, genus_divergence=0.2)
idx = ctx.index(ref, 16, 8)
reads, truth = ctx.synth_reads(ref, seed=1000, n_reads=int(os.environ.get("NR", "100000")), read_len=int(os.environ.get("RL", "10000")), read_len_min=int(os.environ.get("RLMIN", "0")),
                               sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
best = None
for it in range(int(os.environ.get("ITERS", "3"))):
    M = ctx.map_batch(idx, reads, 16, 8)
    st = M.stats()
    M.close()
    if best is None or st["ms_total"] < best["ms_total"]:
        best = st
print({k: round(v, 2) for k, v in best.items() if k.startswith("ms_")}, flush=True)
print({k: v for k, v in best.items() if not k.startswith("ms_")}, flush=True)

"""K3 (seed_filter_stream_kernel) on the bench batch with fewer resident workgroups (MM_SF_GRID): does its time follow the number of CUs at work
(the CU's own phases bound it) or stay put (the memory side bounds it)?   python tools/sf_grid_sweep.py [grids...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi
ctx = capi.Context(0)
ref, genome = ctx.synth_community(seed=20260928, n_genomes=12000, n_species=3000, n_genera=600, median_len=2.0e6, sigma_len=0.6, min_len=5000, max_len=12_000_000,
                                  strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
                                  human_contigs=24, human_bases=int(3.1e9), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000, total_bases_target=26_762_276_280)
idx = ctx.index(ref, 16, 8)
reads, truth = ctx.synth_reads(ref, seed=1000, n_reads=100000, read_len=10000, read_len_min=0, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
grids = [int(a) for a in sys.argv[1:]] or [256, 224, 192, 160, 128, 96, 64, 256]
for g in grids:
    os.environ["MM_SF_GRID"] = str(g)
    best = None
    for it in range(3):
        M = ctx.map_batch(idx, reads, 16, 8)
        st = M.stats()
        M.close()
        if best is None or st["ms_hit_filter"] < best:
            best = st["ms_hit_filter"]
    print(f"MM_SF_GRID={g:4d}  K3 {best:7.2f} ms   x grid / 256 = {best * g / 256:6.2f}", flush=True)

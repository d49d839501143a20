import sys, time
sys.path.insert(0, '.')
from metamaps_amd import capi
import numpy as np
ctx = capi.Context(0)
ref = ctx.synth_reference(seed=20260928, n_species=int(sys.argv[1]), strains_per_species=4, genome_len=int(sys.argv[2]), strain_divergence=0.02, genus_divergence=0.2)
idx = ctx.index(ref, 16, 8)
reads, truth = ctx.synth_reads(ref, seed=1000, n_reads=100000, read_len=10000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
for it in range(3):
    t=[time.perf_counter()]
    M = ctx.map_batch(idx, reads, 16, 8); t.append(time.perf_counter())
    M.add_qualities(16); ctx.synchronize(); t.append(time.perf_counter())
    off, rec = M.fetch(); t.append(time.perf_counter())
    st = M.stats(); t.append(time.perf_counter())
    M.close(); ctx.synchronize(); t.append(time.perf_counter())
    print([round((b-a)*1e3,2) for a,b in zip(t,t[1:])], 'map, mapq, fetch, stats, close')

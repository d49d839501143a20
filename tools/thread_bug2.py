import sys, time
sys.path.insert(0, '.')
from metamaps_amd import capi
ctx = capi.Context(0)
print(ctx.device_info(), flush=True)
ref = ctx.synth_reference(seed=20260928, n_species=64, strains_per_species=4, genome_len=500000, strain_divergence=0.02, genus_divergence=0.2)
idx = ctx.index(ref, 16, 8)
print(idx.info(), idx.freq_threshold, flush=True)
c, nh = idx.freq_hist(); print(list(zip(c[:5].tolist(), nh[:5].tolist())), flush=True)
h, ct, wp, st = idx.entries(); print(h[:5], ct[:5], wp[:5], flush=True)
r, _ = ctx.synth_reads(ref, seed=1000, n_reads=2000, read_len=10000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
for it in range(3):
    M = ctx.map_batch(idx, r, 16, 8); s = M.stats(); print({k: s[k] for k in ("n_mappings","sum_hits","sum_hits_kept","n_candidates")}, flush=True); M.close()

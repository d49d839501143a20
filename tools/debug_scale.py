import sys, time, json
sys.path.insert(0, '.')
from metamaps_amd import capi
import numpy as np
sp, strains, glen, nreads = [int(x) for x in sys.argv[1:5]]
ctx = capi.Context(0)
t=time.time(); ref = ctx.synth_reference(seed=20260928, n_species=sp, strains_per_species=strains, genome_len=glen, strain_divergence=0.02, genus_divergence=0.2); ctx.synchronize(); print('ref', time.time()-t, flush=True)
t=time.time(); idx = ctx.index(ref, 16, 8); ctx.synchronize(); print('index', time.time()-t, idx.info(), 'thr', idx.freq_threshold, flush=True)
c, nh = idx.freq_hist(); print('hist head', list(zip(c[:8].tolist(), nh[:8].tolist())), 'tail', list(zip(c[-5:].tolist(), nh[-5:].tolist())), flush=True)
reads, truth = ctx.synth_reads(ref, seed=1000, n_reads=nreads, read_len=10000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
t=time.time(); M = ctx.map_batch(idx, reads, 16, 8); print('map', time.time()-t, json.dumps(M.stats()), flush=True)
print(ctx.device_info())

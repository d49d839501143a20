import sys, time, threading
sys.path.insert(0, '.')
from metamaps_amd import capi
import numpy as np
sp, glen, nthreads, nreads = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctxs = [capi.Context(0) for _ in range(nthreads)]
ref = ctxs[0].synth_reference(seed=20260928, n_species=sp, strains_per_species=4, genome_len=glen, strain_divergence=0.02, genus_divergence=0.2)
idx = ctxs[0].index(ref, 16, 8)
sets = []
for t in range(nthreads):
    r, _ = ctxs[t].synth_reads(ref, seed=1000 + t, n_reads=nreads // nthreads, read_len=10000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
    sets.append(r)
def work(t, out):
    M = ctxs[t].map_batch(idx, sets[t], 16, 8); M.add_qualities(16); out[t] = M.stats()["n_mappings"]; M.close()
for it in range(4):
    out = [0] * nthreads
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t, out)) for t in range(nthreads)]
    [x.start() for x in th]; [x.join() for x in th]
    print(nthreads, 'threads', round((time.perf_counter() - t0) * 1e3, 1), 'ms', out, flush=True)

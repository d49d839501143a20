import sys, time, threading
sys.path.insert(0, '.')
from metamaps_amd import capi
ctx = capi.Context(0)
ref = ctx.synth_reference(seed=20260928, n_species=256, strains_per_species=4, genome_len=1000000, strain_divergence=0.02, genus_divergence=0.2)
idx = ctx.index(ref, 16, 8)
r, _ = ctx.synth_reads(ref, seed=1000, n_reads=20000, read_len=10000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
def work(out):
    M = ctx.map_batch(idx, r, 16, 8); out.append(M.stats()); M.close()
keys = ("n_reads_mapped","n_mappings","sum_sketch","sum_hits","sum_hits_kept","n_candidates","sum_l2_evals","n_ambiguous_sketch_reads","n_l2_wide_redo")
for it in range(6):
    out = []
    if it % 2 == 0: work(out)
    else:
        t = threading.Thread(target=work, args=(out,)); t.start(); t.join()
    print('main' if it % 2 == 0 else 'thread', {k: out[0][k] for k in keys}, flush=True)

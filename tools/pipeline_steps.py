"""Experiment: whole steps (map + mapQ + fetch + EM) of alternating batches on W contexts / host threads of one GPU, index shared.
usage: pipeline_steps.py <workers> <steps> [shape]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metamaps_amd import capi
W, K = int(sys.argv[1]), int(sys.argv[2])
shape = sys.argv[3] if len(sys.argv) > 3 else "uniform"
ctxs = [capi.Context(0) for _ in range(W)]
for c in ctxs:
    c.comm_init(capi.Context.comm_unique_id(), 0, 1)
c0 = ctxs[0]
if shape == "uniform":
    ref = c0.synth_reference(seed=20260928, n_species=3000, strains_per_species=4, genome_len=2_200_000, strain_divergence=0.02, genus_divergence=0.2)
    contig_taxon = np.arange(12000, dtype=np.int32); n_taxa = 12000
else:
    ref, genome = c0.synth_community(seed=20260928, n_genomes=12000, n_species=3000, n_genera=600, median_len=2.0e6, sigma_len=0.6, min_len=5000, max_len=12_000_000,
                                     strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
                                     human_contigs=24, human_bases=int(3.1e9), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000, total_bases_target=26_762_276_280)
    contig_taxon = genome.astype(np.int32); n_taxa = 12001
idx = c0.index(ref, 16, 8)
contig_len = ref.lengths().astype(np.int32)
reads = [ctxs[i].synth_reads(ref, seed=1000, n_reads=100000, read_len=10000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)[0] for i in range(W)]
bases = int(reads[0].lengths().sum())
def step(i):
    c = ctxs[i]
    M = c.map_batch(idx, reads[i], 16, 8); M.add_qualities(16)
    off, rec = M.fetch()
    em = c.em_from_mapping(M, contig_taxon, contig_len, n_taxa); M.close()
    seen = (em.taxon_counts() > 0).astype(np.float64); c.comm_allreduce(seen)
    f = np.where(seen > 0, 1.0 / max(int((seen > 0).sum()), 1), 0.0)
    f, lls = em.run(f); post, best = em.posteriors(f); em.close()
def worker(i, n):
    for _ in range(n): step(i)
for i in range(W): step(i)
for rep in range(3):
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i, K // W)) for i in range(W)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(W, "workers", shape, round(dt / (K // W * W) * 1e3, 2), "ms/step", round(bases * (K // W * W) / dt / 1e9, 2), "Gbp/s", flush=True)

"""--maxmemory chunk planner at miniSeq+H scale (26.4 Gbp): boundaries and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi
ctx = capi.Context(0)
ref = ctx.synth_reference(seed=20260928, n_species=3000, strains_per_species=4, genome_len=2_200_000, strain_divergence=0.02, genus_divergence=0.2)
t0 = time.time(); idx = ctx.index(ref, 16, 8); ctx.synchronize(); print("index", round(time.time() - t0, 2), "s", idx.info())
for gib in (262, 128, 64):
    t0 = time.time()
    fc = idx.plan_chunks(gib << 30)
    print(f"--mm {gib}: {len(fc)} chunks, first contigs {fc[:8]}{'...' if len(fc) > 8 else ''}  ({time.time() - t0:.2f} s)", flush=True)

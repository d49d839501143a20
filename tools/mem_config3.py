"""where the device memory of the config-3 bench goes: free memory (hipMemGetInfo via torch) after every setup stage and after a step on each worker context"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from metamaps_amd import capi

def free(tag):
    torch.cuda.synchronize()
    f, t = torch.cuda.mem_get_info(0)
    print(f"{tag:60s} used {(t - f) / 2**30:7.1f} GiB  free {f / 2**30:7.1f} GiB", flush=True)

sys.argv = ["bench.py", "--config", "3"]
args = bench.parse_args()
args.read_len, args.read_len_min, args.pacbio, args.reads = 50_000, 1_000, True, 60_000
W = int(os.environ.get("W", "2"))
ctxs = [capi.Context(0) for _ in range(W)]
ctx = ctxs[0]
free("contexts")
ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
free("reference")
whole = ctx.index(ref, 16, 8); plan = whole.plan_chunks(int(70.0 * (1 << 30))); print("whole index hbm GiB", whole.info()["hbm_bytes"] / 2**30); whole.close()
free("whole index built, planned, closed")
bounds = [(a, (plan[i + 1] if i + 1 < len(plan) else ref.count) - a) for i, a in enumerate(plan)]
chunk_idx = []
for a, n in bounds:
    sl = ref.slice(a, n); ix = ctx.index(sl, 16, 8, auto_threshold=False); sl.close(); ix.set_freq_threshold(500); chunk_idx.append(ix)
    free(f"chunk index {len(chunk_idx)}: {ix.info()['hbm_bytes'] / 2**30:.1f} GiB, {ix.info()['n_entries'] / 1e9:.2f}e9 entries")
err = dict(sub_rate=0.02, ins_rate=0.08, del_rate=0.02)
batches = [ctx.synth_reads(ref, seed=1000 + 97 * b, n_reads=60_000, read_len=50_000, read_len_min=1_000, frac_random=0.05, n_abundant=100, **err)[0] for b in range(4)]
free("4 read batches")
base = [a for a, _ in bounds]
for wi, c in enumerate(ctxs):
    for rep in range(2):
        parts = []
        for ix in chunk_idx:
            parts.append(c.map_batch(ix, batches[(wi + rep) % 4], 16, 8, pi=80.0, min_read_len=1000, sketch_of=parts[0] if parts else None))
            free(f"  ctx {wi} rep {rep}: mapped chunk {len(parts)} (hits {parts[-1].stats()['sum_hits'] / 1e9:.2f}e9 kept {parts[-1].stats()['sum_hits_kept'] / 1e9:.2f}e9 cands {parts[-1].stats()['n_candidates']})")
        U = capi.Mapping.concat(c, parts, base)
        for p_ in parts: p_.close()
        U.add_qualities(16); U.fetch(); U.close()
        free(f"ctx {wi} rep {rep}: step done")

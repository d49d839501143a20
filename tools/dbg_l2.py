"""Compare the skip-ahead L2 kernel with the full slide candidate by candidate (debug aid, GPU only)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi

ctx = capi.Context(0)
ref = ctx.synth_reference(seed=5, n_species=30, strains_per_species=4, genome_len=200_000, strain_divergence=0.02, genus_divergence=0.08)
reads, _ = ctx.synth_reads(ref, seed=11, n_reads=int(os.environ.get("NREADS", "400")), read_len=int(os.environ.get("RLEN", "5000")), sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.1, n_abundant=20)
idx = ctx.index(ref, 16, 8)
res = {}
for mode in ("0", "1"):
    os.environ["MM_L2_FULL"] = mode
    M = ctx.map_batch(idx, reads, 16, 8)
    off, cand = M.debug_candidates()
    res[mode] = (M.debug_l2(len(cand)).copy(), cand.copy(), M.stats())
    M.close()
a, b = res["0"][0], res["1"][0]
bad = np.nonzero((a != b).any(axis=1))[0]
print("candidates", len(a), "mismatching", len(bad), "evals skip/full", res["0"][2]["sum_l2_evals"], res["1"][2]["sum_l2_evals"], "rebuilds", res["0"][2]["n_l2_rebuilds"])
for i in bad[:12]:
    print(i, "cand", res["0"][1][i], "skip", a[i], "full", b[i])

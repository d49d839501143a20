"""How long are the occurrence lists the bench batch asks the bench index for?  (Round-4 review, item 3: before a table layout with short lists inline
is built, measure what share of the PROBED lists is short.)  One JSON line: the distribution of list lengths over the sketch hashes of one
batch of 10^5 x 10 kb reads against the community reference, and over the hashes of the index (mm_index_freq_hist).
    python tools/probed_lists.py [--scale 1.0] [--reads 100000]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--window", type=int, default=8)
    a = ap.parse_args()
    s = a.scale
    ctx = capi.Context(0)
    ng, sp, ge = max(4, int(12000 * s)), max(2, int(3000 * s)), max(1, int(600 * s))
    human = max(1, int(round(24 * min(s, 1.0)))) if s >= 0.04 else 0
    ref, _ = ctx.synth_community(seed=20260928, n_genomes=ng, n_species=sp, n_genera=ge, median_len=2.0e6, sigma_len=0.6, min_len=5_000, max_len=12_000_000,
                                 strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
                                 human_contigs=human, human_bases=int(3.1e9 * min(s, 1.0)), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000,
                                 total_bases_target=int(26_762_276_280 * s))
    idx = ctx.index(ref, 16, a.window)
    reads, _t = ctx.synth_reads(ref, seed=1000, n_reads=a.reads, read_len=10000, read_len_min=0, frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
    M = ctx.map_batch(idx, reads, 16, a.window)
    nb = 130
    h = M.debug_probed_lists(idx, nb)
    st = M.stats()
    M.close()
    tot = int(h.sum())
    cum = np.cumsum(h[1:nb - 2])
    present = int(h[1:].sum())
    counts, nh = idx.freq_hist()
    cum_i = {c: int(nh[counts <= c].sum()) for c in (1, 2, 3, 4, 5, 8, 16, 24, 32, 64, 128)}
    U = int(nh.sum())
    ent_i = {c: int((counts * nh)[counts <= c].sum()) for c in (1, 2, 3, 4, 5, 8, 16, 24, 32, 64, 128)}
    out = {"reads": a.reads, "scale": s, "freq_threshold": idx.freq_threshold, "index": idx.info(),
           "probes": tot, "absent": int(h[0]), "cut_by_threshold": int(h[nb - 1]), "longer_than_%d" % (nb - 3): int(h[nb - 2]),
           "share_of_present_probes_with_at_most": {str(c): round(float(cum[c - 1]) / present, 4) for c in (1, 2, 3, 4, 5, 8, 16, 24, 32, 64, 127)},
           "entries_of_probed_lists_with_at_most": {str(c): int((np.arange(1, c + 1) * h[1:c + 1]).sum()) for c in (1, 2, 3, 4, 5, 8, 16, 24, 32, 64, 127)},
           "sum_hits": st["sum_hits"], "sum_sketch": st["sum_sketch"],
           "share_of_index_hashes_with_at_most": {str(c): round(v / U, 4) for c, v in cum_i.items()},
           "index_entries_in_lists_with_at_most": ent_i,
           "hist_first_40": [int(x) for x in h[:41]]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""How much of K3's time (random requests for table sectors and occurrence lists) is the ORDER of the reads?  The bench batch (10^5 reads from 100
abundant genomes: ~5 x coverage, so every hash is looked up by several reads) is mapped in file order, ordered by where the reads map (the
best case for the caches: reads that share hashes are in flight together) and ordered by their smallest sketch hash (what a device-side
ordering could know before the seed stage).  One JSON line: stage times per order.
    python tools/k3_locality.py [--scale 1.0] [--reads 100000]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi  # noqa: E402


def stage(ctx, idx, reads, reps=3):
    best = None
    for _ in range(reps):
        M = ctx.map_batch(idx, reads, 16, 8)
        st = M.stats()
        M.close()
        if best is None or st["ms_probe_gather"] < best["ms_probe_gather"]:
            best = st
    return {k: round(best[k], 3) for k in ("ms_minimizer", "ms_sketch", "ms_probe_gather", "ms_sort_hits", "ms_l1_scan", "ms_l2", "ms_total")}


def reorder(ctx, reads, order):
    buf, ln = reads.fetch_range(0, reads.count)
    at = np.concatenate(([0], np.cumsum(ln)))
    mv = memoryview(buf)
    return ctx.seqset([bytes(mv[at[i]:at[i + 1]]) for i in order.tolist()])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=100000)
    a = ap.parse_args()
    s = a.scale
    ctx = capi.Context(0)
    ng, sp, ge = max(4, int(12000 * s)), max(2, int(3000 * s)), max(1, int(600 * s))
    human = max(1, int(round(24 * min(s, 1.0)))) if s >= 0.04 else 0
    ref, _ = ctx.synth_community(seed=20260928, n_genomes=ng, n_species=sp, n_genera=ge, median_len=2.0e6, sigma_len=0.6, min_len=5_000, max_len=12_000_000,
                                 strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
                                 human_contigs=human, human_bases=int(3.1e9 * min(s, 1.0)), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000,
                                 total_bases_target=int(26_762_276_280 * s))
    idx = ctx.index(ref, 16, 8)
    reads, _t = ctx.synth_reads(ref, seed=1000, n_reads=a.reads, read_len=10000, read_len_min=0, frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
    out = {"reads": a.reads, "file_order": stage(ctx, idx, reads)}
    M = ctx.map_batch(idx, reads, 16, 8)
    off, rec = M.fetch()
    soff, sh, _ss = M.debug_sketch()
    M.close()
    n = reads.count
    key = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
    has = np.diff(off) > 0
    first = rec[off[:-1][has]]
    key[has] = first["ref_contig"].astype(np.int64) << 32 | first["ref_start"].astype(np.int64)
    by_pos = np.argsort(key, kind="stable")
    r2 = reorder(ctx, reads, by_pos)
    out["by_mapped_position"] = stage(ctx, idx, r2)
    r2.close()
    minh = np.full(n, 2**32 - 1, dtype=np.int64)
    nz = np.diff(soff) > 0
    minh[nz] = sh[soff[:-1][nz]]
    by_hash = np.argsort(minh, kind="stable")
    r3 = reorder(ctx, reads, by_hash)
    out["by_smallest_sketch_hash"] = stage(ctx, idx, r3)
    r3.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

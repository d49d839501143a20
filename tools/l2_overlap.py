"""How much of what K5 streams is streamed more than once?  A candidate streams the index entries of [rangeStart, rangeEnd + readLength) of its
contig (computeMap.hpp:466-477); L1 merges candidates of a contig only when their [start, end] ranges touch (:374-380), so two candidates of one read
on one contig whose ranges lie closer than a read length stream common entries.  Prints the share of the streamed positions that lie in such overlaps
(the round-3 review asked: share pass A if it is above 10 %).  Usage: python tools/l2_overlap.py [n_reads]   (bench reference, full size)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metamaps_amd import capi


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    sys.argv = sys.argv[:1]
    args = bench.parse_args()
    ctx = capi.Context(0)
    ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
    idx = ctx.index(ref, 16, 8)
    reads, _ = ctx.synth_reads(ref, seed=1000, n_reads=n, read_len=10000, read_len_min=0, frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
    M = ctx.map_batch(idx, reads, 16, 8)
    off, tri = M.debug_candidates()
    lens = reads.lengths()
    tot = dup = 0
    pairs = 0
    for r in range(n):
        c = tri[off[r]:off[r + 1]]
        if len(c) == 0:
            continue
        L = int(lens[r])
        lo, hi, cg = c[:, 1].astype(np.int64), c[:, 2].astype(np.int64) + L, c[:, 0]
        tot += int((hi - lo).sum())
        same = cg[1:] == cg[:-1]                                  # candidates are in (contig, position) order
        ov = np.maximum(0, hi[:-1] - lo[1:])[same]
        dup += int(ov.sum()); pairs += int((ov > 0).sum())
    print(f"{n} reads, {len(tri)} candidates: {tot / 1e9:.3f} G streamed positions, {dup / 1e9:.4f} G of them ({100.0 * dup / max(tot, 1):.2f} %) in the overlap of "
          f"two candidates of one read on one contig ({pairs} such pairs)")


if __name__ == "__main__":
    main()

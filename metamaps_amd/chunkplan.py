"""The --maxmemory chunk rule (Sketch::build's flush test, winSketch.hpp:274-329) for a reference whose whole index does not fit the device, and the
per-chunk frequency thresholds of a pass over such chunks — host-side glue over the C ABI, shared by bench.py (--config 5) and
tests/test_gpu_refseq_scale.py.  The drop-in CLI does the same in C++ (metamaps_main.cpp)."""
from __future__ import annotations

import numpy as np

INT_MAX = 2**31 - 1


def plan_chunks_by_ranges(ctx, ref, contig_len, k: int, w: int, max_memory: int, range_bases: int):
    """First contig of every index chunk under `max_memory` (mm_index_plan_chunks evaluates the rule on the index of a WHOLE reference): the rule is
    evaluated on the indexes of contig ranges of about `range_bases` bases — every cut inside a range is final, the range's last, open chunk starts
    the next range; a range the rule does not cut at all is doubled.  Returns (first contigs, {"n_entries", "hbm_bytes" (largest range index)})."""
    C = ref.count
    plan, c0, info = [0], 0, {"n_contigs": C, "n_entries": 0, "n_unique_hashes": 0, "hbm_bytes": 0}
    while c0 < C:
        c1, bases = c0, 0
        while c1 < C and (bases < range_bases or c1 == c0):
            bases += int(contig_len[c1]); c1 += 1
        sl = ref.slice(c0, c1 - c0); ri = ctx.index(sl, k, w, auto_threshold=False); sl.close()
        loc = ri.plan_chunks(max_memory)
        ii = ri.info()
        info["n_entries"] += ii["n_entries"] if c0 == 0 or len(loc) > 1 else 0
        info["hbm_bytes"] = max(info["hbm_bytes"], ii["hbm_bytes"])
        ri.close()
        if len(loc) == 1 and c1 < C:
            range_bases = bases * 2
            continue
        plan += [c0 + x for x in loc[1:]]
        if c1 == C:
            break
        c0 += loc[-1]
    return plan, info


def chunk_bounds(plan, n_contigs: int):
    """[(first contig, number of contigs)] of the chunks of a plan"""
    return [(a, (plan[i + 1] if i + 1 < len(plan) else n_contigs) - a) for i, a in enumerate(plan)]


class AccumulatedThreshold:
    """freqThreshold of chunk after chunk: the occurrence histogram is never cleared between the chunks of a run (winSketch.hpp:452-494), so
    chunk c's cut-off comes from the histograms of chunks 0 .. c.  next(ix) takes the freshly built index of the next chunk, sets and returns its threshold."""

    def __init__(self):
        self.acc: dict[int, int] = {}
        self.thr = INT_MAX

    def next(self, ix) -> int:
        from . import capi
        counts, nh = ix.freq_hist()
        for c_, n_ in zip(counts.tolist(), nh.tolist()):
            self.acc[c_] = self.acc.get(c_, 0) + n_
        cc = np.array(sorted(self.acc), dtype=np.int64)
        hh = np.array([self.acc[c_] for c_ in cc.tolist()], dtype=np.int64)
        self.thr = int(capi.lib().mm_freq_threshold_from_hist(cc.ctypes.data, hh.ctypes.data, len(cc), ix.info()["n_unique_hashes"], self.thr))
        ix.set_freq_threshold(self.thr)
        return self.thr

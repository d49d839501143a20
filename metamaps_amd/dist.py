"""One-process-per-GPU orchestration helpers (mapping shards by read; EM exchanges one small vector per
iteration).  Backend-agnostic so that the logic is testable with gloo on CPUs: the compute callbacks are
injected (libmetamaps_hip on a GPU box; the tests inject the oracle).

TEST-SIDE code: the product (libmetamaps_hip.so, the `metamaps` CLI) never imports this module.  bench.py uses `shard_range` to cut its strong-scaling
batch; `em_distributed` is the loop the world-size-2 gloo test (tests/test_em_host_and_dist.py) and the eight-shard test on one GPU
(tests/test_gpu_fullsize.py) hold the per-rank partial sums against — the product's own exchange is C++ (mm_post.hip: P1-P3' | ncclAllReduce | finalize)."""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous read shard of `rank`; concatenating shards in rank order restores input order
    (the only ordering the reference guarantees, ThreadPool.hpp:13-17)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def em_distributed(local_step, allreduce_sum, seen_local: np.ndarray, max_iter: int = 10_000):
    """meta::doEM's loop (fEM.h:491-661) with the per-thread partial sums replaced by per-rank partial sums.

    local_step(f) -> (partial f_next over this rank's reads [T], partial log-likelihood)
    allreduce_sum(vec) -> element-wise sum over ranks (in place or returned)
    seen_local[T] -> 1.0 for taxa that occur in this rank's mappings (the reference initialises f over the
                     taxa seen in the whole mappings file, fEM.h:471,491-495)."""
    seen = allreduce_sum(np.asarray(seen_local, dtype=np.float64).copy()) > 0
    n_seen = int(seen.sum())
    f = np.where(seen, 1.0 / n_seen, 0.0)
    lls, ll_prev = [], 0.0
    for it in range(max_iter):
        part, ll = local_step(f)
        buf = allreduce_sum(np.concatenate([np.asarray(part, dtype=np.float64), [ll]]))
        tot = buf[:-1]
        ll = float(buf[-1])
        f_next = tot / tot.sum()                                 # fEM.h:606-615
        lls.append(ll)
        stop = it > 0 and (ll - ll_prev) <= 1 and (1 - ll / ll_prev) < 1e-4   # fEM.h:624-639
        f, ll_prev = f_next, ll
        if stop:
            break
    return f, lls

"""Latency of one EM iteration (K9) on an idle GPU, on a problem of the bench's shape (10^5 reads, ~4 mappings per read, 12 001 taxa of
which a few hundred have mappings, lognormal abundances): the resident kernel at several grid sizes, the same phases as separate launches
(MM_EM_SPLIT), and the collective form (kernel A | ncclAllReduce | kernel B) on a one-rank communicator.  Prints one line per variant.
Usage: python tools/em_latency.py [n_reads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi


def problem(n_reads, n_taxa=12001, n_present=300, seed=1):
    rng = np.random.default_rng(seed)
    present = rng.choice(n_taxa, size=n_present, replace=False)
    ab = rng.lognormal(0, 1.5, n_present); ab /= ab.sum()
    true = rng.choice(n_present, size=n_reads, p=ab)
    extra = rng.poisson(3.2, size=n_reads)
    off = np.concatenate([[0], np.cumsum(1 + extra)]).astype(np.int64)
    taxon = present[rng.integers(0, n_present, size=int(off[-1]))].astype(np.int32)
    taxon[off[:-1]] = present[true]
    mapq = rng.uniform(0.01, 1.0, len(taxon))
    inv = 1.0 / rng.integers(1_000_000, 8_000_000, size=len(taxon)).astype(np.float64)
    return off, taxon, mapq, inv, n_taxa


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    off, taxon, mapq, inv, T = problem(n_reads)
    print(f"{n_reads} reads, {len(taxon)} mappings, {T} taxa, largest taxon {np.bincount(taxon).max()} mappings")
    f0 = np.full(T, 1.0 / T)
    variants = [("launch per phase (default)", {}), ("P1 | P2+P3 in one launch", {"MM_EM_SPLIT": "2"}), ("thread-per-read P1", {"MM_EM_DBG": "3"}), ("resident grid 128", {"MM_EM_RESIDENT": "1"}), ("resident grid 64", {"MM_EM_GRID": "64", "MM_EM_RESIDENT": "1"}), ("resident grid 256", {"MM_EM_GRID": "256", "MM_EM_RESIDENT": "1"}), ("launches, grid 256", {"MM_EM_GRID": "256"}), ("launches, grid 1024", {"MM_EM_GRID": "1024"}),
                ("collective, one rank", {"MM_EM_FORCE_COLLECTIVE": "1", "_comm": "1"}),
                ("collective, P2+P3 in one launch", {"MM_EM_FORCE_COLLECTIVE": "1", "MM_EM_SPLIT": "2", "_comm": "1"}),
                ("collective, kernel A resident", {"MM_EM_FORCE_COLLECTIVE": "1", "MM_EM_RESIDENT": "1", "_comm": "1"})]
    for name, env in variants:
        for k in ("MM_EM_GRID", "MM_EM_SPLIT", "MM_EM_FORCE_COLLECTIVE", "MM_EM_ORDER", "MM_EM_DBG", "MM_EM_RESIDENT"):
            os.environ.pop(k, None)
        for k, v in env.items():
            if not k.startswith("_"):
                os.environ[k] = v
        ctx = capi.Context(0)
        if env.get("_comm"):
            ctx.comm_init(capi.Context.comm_unique_id(), 0, 1)
        e = ctx.em(off, taxon, mapq, inv, T)
        e.run(f0, max_iter=3)                                     # first launch, allocation of the loop's buffers
        best = None
        for n_it in (40, 40, 200):
            ctx.synchronize()
            t0 = time.perf_counter()
            f, lls = e.run(f0, max_iter=n_it)
            dt = time.perf_counter() - t0
            per = dt / max(len(lls), 1) * 1e6
            best = per if best is None else min(best, per)
        print(f"{name:<24} {best:8.1f} us per iteration ({len(lls)} iterations in the last run, ll {lls[-1]:.6f})")
        e.close(); ctx.close()


def phases(dbg=None):
    """MM_EM_PROF=1: the library prints workgroup 0's time per phase of the resident kernel (dbg: MM_EM_DBG, parts of P1 left out)"""
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    off, taxon, mapq, inv, T = problem(n_reads)
    for k in ("MM_EM_GRID", "MM_EM_SPLIT", "MM_EM_FORCE_COLLECTIVE", "MM_EM_DBG"):
        os.environ.pop(k, None)
    os.environ["MM_EM_PROF"] = "1"; os.environ["MM_EM_RESIDENT"] = "1"      # (the phase clocks live in the resident kernel)
    if dbg:
        os.environ["MM_EM_DBG"] = dbg
        print(f"MM_EM_DBG={dbg}:", file=sys.stderr, flush=True)
    ctx = capi.Context(0)
    e = ctx.em(off, taxon, mapq, inv, T)
    e.run(np.full(T, 1.0 / T), max_iter=3)
    sys.stderr.flush()
    t0 = time.perf_counter(); f, lls = e.run(np.full(T, 1.0 / T), max_iter=12); dt = time.perf_counter() - t0
    print(f"with MM_EM_PROF: {dt / len(lls) * 1e6:.1f} us per iteration over {len(lls)} iterations (phase split on stderr)")
    e.close(); ctx.close()
    os.environ.pop("MM_EM_PROF", None); os.environ.pop("MM_EM_RESIDENT", None)


if __name__ == "__main__":
    main()
    phases()
    phases("3")                                                   # the thread-per-read form of P1

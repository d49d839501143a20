"""Plain-Python statements of the *parallel* forms the HIP kernels implement (not the reference's
serial loops).  CPU tests check them against the oracle; the GPU tests then check the kernels."""
from __future__ import annotations


def winnow_pure(hash_fwd, hash_bwd, w):
    """K1 (csrc/mm_minimizer.hpp): positions -> emitted (hash, wpos, strand) list.

    hash_fwd/hash_bwd: per-position forward / reverse-complement k-mer hashes."""
    n = len(hash_fwd)
    ns = [hash_fwd[i] != hash_bwd[i] for i in range(n)]
    can = [min(hash_fwd[i], hash_bwd[i]) for i in range(n)]
    fwd = [hash_fwd[i] < hash_bwd[i] for i in range(n)]

    def c(i):
        best, bj = can[i], i
        for q in range(i - 1, max(i - w, -1), -1):
            if ns[q] and can[q] < best:
                best, bj = can[q], q
        return bj

    # jstar: first change point after window 0 whose (hash,strand) differs from the window-0 emission
    jstar = w - 1
    if n >= w and ns[w - 1]:
        c0 = c(w - 1)
        jstar = n
        prev = c0
        for p in range(w, n):
            if not ns[p]:
                continue
            cur = c(p)
            if cur != prev and (can[cur] != can[c0] or fwd[cur] != fwd[c0]):
                jstar = p
                break
            prev = cur
    out = []
    for p in range(w - 1, n):
        if not ns[p]:
            continue
        pj = -1
        for q in range(p - 1, max(p - w, w - 2), -1):
            if ns[q]:
                pj = q
                break
        emit = pj < 0 or c(pj) != c(p)
        if w - 1 < p < jstar:
            emit = False
        if emit:
            cp = c(p)
            out.append((can[cp], p - w + 1, 1 if fwd[cp] else -1))
    return out

"""Independent cross-checks of the oracle's stateful restatements (validation notes H1/H5 and A12).

The reference cannot be built in this image (Boost), so the oracle's A6/A7/A8/A12 logic is pinned here against *definitions*
instead of against the serial data structures it restates:

  A7  sliding MinHash window   brute force per window:  shared(W) = |{h in Q ∩ W : h among the s smallest of Q ∪ W}|
                               over every window of the MIIteratorL2 sequence, stated by positions (a window is the set of index
                               entries with wpos in [wpos of the last entry at or before p, p + cnt - 1]) — no ordered map, no pivot
                               iterator, no insert/erase bookkeeping (slidingMap.hpp:139-316 is what this replaces)
  A8  strand vote              sum over the same hashes of strandQ * strandR(last occurrence in the window)
  A6  L1 candidates            union of the intervals [max(0, wpos[i+m-1] - len + 1), wpos[i]] over every run of m hits closer
                               than the read length (computeMap.hpp:355-385 is a running merge)
  A5  seed hits                every index entry whose hash is in the sketch and occurs fewer than freqThreshold times
  A12 mapping qualities        scipy.stats.binom.pmf (scipy embeds Boost.Math, the library the reference calls, mapWrap.h:340)
                               + float32 exp + "%g" text, against the oracle's add_mapping_qualities on thousands of lines
More than 10^5 windows are checked, on a reference with segmental duplications, tandem repeats and a homopolymer so that
duplicate hashes inside one window (the REV / NOOP cases of slidingMap.hpp:148-157, :186-209) are common."""
import os

import numpy as np
import pytest


def _write_db(path, rng):
    """6 contigs: related pairs, segmental duplications inside a contig, a tandem repeat and a homopolymer run"""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    base = rng.choice(acgt, size=60_000)
    seqs.append(base.copy())
    m = base.copy(); mut = rng.random(len(m)) < 0.03; m[mut] = rng.choice(acgt, size=int(mut.sum())); seqs.append(m)
    d = rng.choice(acgt, size=50_000)
    seg = d[5_000:9_000].copy()
    for at in (15_000, 21_000, 40_000):                           # the same 4 kb three more times: duplicate hashes within windows
        s2 = seg.copy(); mm = rng.random(len(s2)) < 0.01; s2[mm] = rng.choice(acgt, size=int(mm.sum())); d[at:at + 4_000] = s2
    d[30_000:30_600] = np.frombuffer(b"ACGTTGCA" * 75, dtype=np.uint8)
    d[33_000:33_300] = ord("A")
    seqs.append(d)
    seqs.append(rng.choice(acgt, size=45_000))
    e = seqs[3].copy(); mut = rng.random(len(e)) < 0.08; e[mut] = rng.choice(acgt, size=int(mut.sum())); seqs.append(e)
    seqs.append(rng.choice(acgt, size=500))
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(f">C{i}|kraken:taxid|{100 + i}|x\n".encode() + s.tobytes() + b"\n")
    return seqs


def _reads(seqs, rng, n, lens):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    out = []
    for i in range(n):
        g = 2 if i % 2 else int(rng.integers(0, 5)); L = int(lens[i % len(lens)])   # every other read from the contig with the duplications
        p = int(rng.integers(0, len(seqs[g]) - L))
        s = seqs[g][p:p + L].copy()
        mut = rng.random(L) < 0.06; s[mut] = rng.choice(acgt, size=int(mut.sum()))
        if i % 3 == 0:
            s = comp[s[::-1]]
        out.append(s.tobytes())
    return out


def _brute_l2(H, S, WP, ST, qh, qs, cand, L, k, w):
    """(shared, meanPos, optBeg, optEnd, strandVotes, n_windows, n_windows_with_a_repeated_hash) of one candidate, by definition"""
    seq, rs, re = (int(x) for x in cand)
    lo, hi = np.searchsorted(S, seq, "left"), np.searchsorted(S, seq, "right")       # the contig's entries
    wp = WP[lo:hi]
    first = lo + int(np.searchsorted(wp, rs, "left"))
    cnt = L - (w - 1) - (k - 1)
    first_end = lo + int(np.searchsorted(wp, WP[first] + cnt, "left"))
    last_end = lo + int(np.searchsorted(wp, re + L, "left"))
    p0 = int(WP[first])
    ev = np.unique(np.concatenate([WP[first + 1:hi], WP[first_end:last_end] - (cnt - 1)]))
    ev = ev[ev > p0]
    s = len(qh)
    best, beg, last, ob, oe, nwin, ndup = 0, 0, 0, 0, 0, 0, 0
    for p in [p0] + [int(x) for x in ev]:
        b = lo + int(np.searchsorted(wp, p, "right")) - 1
        e = lo + int(np.searchsorted(wp, p + cnt - 1, "right"))
        if e >= last_end:
            break
        nwin += 1
        wd = np.unique(H[b:e])
        ndup += int(len(wd) < e - b)
        union = np.union1d(qh, wd)
        thr = union[s - 1]
        both = np.intersect1d(qh, wd, assume_unique=True)
        sh = int(np.count_nonzero(both <= thr))
        if sh > best:
            best, beg, last, ob, oe = sh, int(WP[b]), int(WP[b]), b, e
        elif sh == best:
            last = int(WP[b])
    votes = 0
    if best > 0:
        wd = np.unique(H[ob:oe]); union = np.union1d(qh, wd); thr = union[s - 1]
        strand_r = {}
        for i in range(ob, oe):
            strand_r[int(H[i])] = int(ST[i])                     # the last occurrence wins (insert_ref overwrites, slidingMap.hpp:155-156)
        qidx = {int(h): i for i, h in enumerate(qh)}
        for h, sr in strand_r.items():
            if h <= thr and h in qidx:
                votes += int(qs[qidx[h]]) * sr
    return best, (beg + last) // 2, ob, oe, votes, nwin, ndup


def _interval_union(hc, hw, m, L):
    """L1 candidates as a union of closed intervals (touching or overlapping intervals of one contig merge)"""
    out = []
    n = len(hc)
    for i in range(0, n - m + 1):
        j = i + m - 1
        if hc[i] != hc[j] or hw[j] - hw[i] >= L:
            continue
        out.append((int(hc[i]), max(0, int(hw[j]) - L + 1), int(hw[i])))
    merged = []
    for c in sorted(out):
        if merged and merged[-1][0] == c[0] and merged[-1][2] >= c[1]:
            merged[-1][2] = max(merged[-1][2], c[2])
        else:
            merged.append(list(c))
    return np.array(merged, dtype=np.int64).reshape(-1, 3)


@pytest.mark.parametrize("k,w", [(16, 5), (12, 8)])
def test_l1_l2_and_vote_against_definitions(oracle_lib, tmp_path, k, w):
    rng = np.random.default_rng(100 + k)
    fa = str(tmp_path / "DB.fa")
    seqs = _write_db(fa, rng)
    oi = oracle_lib.index(fa, k, w)
    H, S, WP, ST = oi.dump()
    assert np.all(np.diff(S.astype(np.int64) * (1 << 32) + WP) > 0)      # position order, no two entries at one place
    counts = dict(zip(*np.unique(H, return_counts=True)))
    n_windows = n_cand = n_dupwin = 0
    for q in _reads(seqs, rng, 36, (1200, 2000, 3500, 5000)):
        r = oi.map_read(q)
        qh, qs = r["sketch_hash"], r["sketch_strand"]
        assert np.all(np.diff(qh.astype(np.int64)) > 0)
        # A5: the seed hits are every occurrence of every sketch hash below the frequency threshold, sorted by (contig, position)
        keep = np.isin(H, np.array([h for h in qh if counts.get(h, 0) < oi.freq_threshold], dtype=np.uint32))
        order = np.lexsort((WP[keep], S[keep]))
        assert np.array_equal(S[keep][order], r["hit_contig"]) and np.array_equal(WP[keep][order], r["hit_wpos"])
        # A6
        cu = _interval_union(r["hit_contig"], r["hit_wpos"], max(1, r["min_hits"]), len(q))
        assert np.array_equal(cu, r["cand"].astype(np.int64).reshape(-1, 3))
        # A7 / A8
        accepted = {(int(m[0]), int(m[1])): int(m[5]) for m in r["map"]}
        for ci, cand in enumerate(r["cand"]):
            sh, mean, ob, oe, votes, nw, nd = _brute_l2(H, S, WP, ST, qh, qs, cand, len(q), k, w)
            o = r["l2"][ci]
            assert int(o[2]) == sh, (cand, o, sh)
            if sh > 0:
                assert (int(o[1]), int(o[3]), int(o[4])) == (mean, ob, oe), (cand, o, mean, ob, oe)
                if (int(cand[0]), mean) in accepted:
                    assert accepted[(int(cand[0]), mean)] == (1 if votes > 0 else -1)
            n_windows += nw; n_cand += 1
            n_dupwin += nd
    oi.close()
    assert n_cand > 60 and n_windows > 100_000 and n_dupwin > 2_000, (n_cand, n_windows, n_dupwin)


def test_forced_frequency_threshold_changes_hits_consistently(oracle_lib, tmp_path):
    """the oracle's threshold setter (used by the GPU density tests): hits = occurrences of sketch hashes with count < threshold"""
    rng = np.random.default_rng(5)
    fa = str(tmp_path / "DB.fa")
    seqs = _write_db(fa, rng)
    oi = oracle_lib.index(fa, 11, 6)
    H, S, WP, ST = oi.dump()
    counts = dict(zip(*np.unique(H, return_counts=True)))
    q = _reads(seqs, rng, 1, (4000,))[0]
    for thr in (2, 3, 5, 1 << 30):
        oi.set_freq_threshold(thr)
        r = oi.map_read(q)
        keep = np.isin(H, np.array([h for h in r["sketch_hash"] if counts.get(h, 0) < thr], dtype=np.uint32))
        assert int(keep.sum()) == len(r["hit_wpos"])
    oi.close()


def test_mapping_qualities_against_scipy(oracle_lib):
    """fields 13 and 14 of mapWrap::addMappingQualities (mapWrap.h:215-356) restated with scipy's (Boost-backed) binomial pmf"""
    from scipy.stats import binom
    rng = np.random.default_rng(9)
    k = 16
    n_lines = 0
    for _ in range(700):
        L = int(rng.integers(1000, 60_000))
        s = int(rng.integers(L // 12, L // 4))
        n_map = int(rng.integers(1, 9))
        hi = int(rng.integers(s // 8 + 1, s + 1))
        lines, shared = [], []
        for i in range(n_map):
            c = int(rng.integers(max(1, hi // 2), hi + 1))
            ident, _ = oracle_lib.identity(c, s, k)
            lines.append(f"read{_} {L} 0 {L - 1} + contig{i} 100000 {i * 7} {i * 7 + L - 1} {np.float32(ident):g} {c} {s}")
            shared.append(c)
        got = oracle_lib.add_mapq(k, lines)
        ids = [float(l.split(" ")[9]) / 100.0 for l in lines]
        max_id = np.exp(-(1 - max(ids)))
        nk = L - k + 1
        es = np.round(max_id ** k * nk)
        p = es / (nk + (nk - es))
        lik = np.array([binom.pmf(c, s, p) for c in shared])
        mq = lik / lik.sum()
        for ln, g, idv, q in zip(lines, got, ids, mq):
            f = g.split(" ")
            assert " ".join(f[:12]) == ln
            assert f[12] == f"{np.float32(np.exp(np.float64(-(1 - idv)))) * np.float32(100):g}"
            assert abs(float(f[13]) - q) <= 1e-5 * max(q, 1e-300) + 1e-12 or f[13] == f"{q:g}", (ln, f[13], q)
            n_lines += 1
    assert n_lines > 2500

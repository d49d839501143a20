"""Randomised differential test of the production configuration (fused seed filter, exact skip-ahead, device-made launch lists,
position directory): seeded random parameters (k, w, identity threshold, minimum read length), small references full of the awkward
cases (related contigs, duplications inside a contig, tandem repeats, homopolymers, N runs, lower case, contigs shorter than
w + k), reads of 60 ... 40 000 bases (shorter than k, shorter than w + k, longer than the 10 kb class) with 0-15 % errors, either
strand, some random, some with N — every mapping record must equal the oracle's (computeMap.hpp:90-538 restated in oracle/).
Round 2: MM_FUZZ_SEEDS=2500 MM_FUZZ_LONG_SEEDS=300 and, with MM_FUZZ_SEED_BASE=100000, 2000 + 250 more ran clean on an MI355X
(66 minutes together); round 5's tree: MM_FUZZ_SEEDS=1500 MM_FUZZ_LONG_SEEDS=150 MM_FUZZ_SEED_BASE=500000 clean in 16 minutes
and, on the rewritten seed filter, 2000 + 200 at base 700000 and 200 + 600 at base 900000 (profiles/r05_fuzz_campaign.txt); round 6's final tree (zone kernel, range kernel, K3's barrier placement):
MM_FUZZ_SEEDS=1500 MM_FUZZ_LONG_SEEDS=150 MM_FUZZ_SEED_BASE=700000, 1650 cases clean in 15 minutes, and 1500 + 200 at base 1100000 (new seeds) in 15; the suite keeps 24 + 6 seeds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8); COMP[list(b"ACGTN")] = list(b"TGCAN")


def _mutate(rng, s, sub, indel):
    out = []
    i = 0
    n = len(s)
    ev = rng.random(n)
    for i in range(n):
        e = ev[i]
        if e < sub: out.append(ACGT[rng.integers(4)])
        elif e < sub + indel / 2: continue                        # deletion
        elif e < sub + indel: out.append(s[i]); out.append(ACGT[rng.integers(4)])   # insertion
        else: out.append(s[i])
    return np.array(out, dtype=np.uint8)


def _reference(rng):
    contigs = []
    base = rng.choice(ACGT, size=int(rng.integers(30_000, 120_000)))
    contigs.append(base)
    rel = base.copy(); m = rng.random(len(rel)) < rng.choice([0.005, 0.03, 0.1]); rel[m] = rng.choice(ACGT, size=int(m.sum()))
    contigs.append(rel[int(rng.integers(0, 2000)):])
    d = rng.choice(ACGT, size=int(rng.integers(20_000, 80_000)))
    seg = d[1000:4000].copy()
    for at in rng.integers(5000, len(d) - 4000, size=3):
        d[at:at + 3000] = seg
    unit = rng.choice(ACGT, size=int(rng.integers(2, 12)))
    d[8000:8600] = np.resize(unit, 600)
    d[9000:9000 + int(rng.integers(20, 400))] = ord("A")
    d[12_000:12_000 + int(rng.integers(1, 300))] = ord("N")
    contigs.append(d)
    contigs.append(rng.choice(ACGT, size=int(rng.integers(5, 60))))          # shorter than w + k
    contigs.append(rng.choice(ACGT, size=int(rng.integers(2_000, 30_000))))
    lower = contigs[-1].copy()
    contigs.append(np.frombuffer(lower.tobytes().lower(), dtype=np.uint8))      # lower case = the same sequence (commonFunc.hpp:57)
    order = rng.permutation(len(contigs))
    return [contigs[i] for i in order]


def _reads(rng, contigs, n):
    big = [c for c in contigs if len(c) > 1500]
    out = []
    for i in range(n):
        L = int(np.exp(rng.uniform(np.log(60), np.log(40_000))))
        kind = rng.random()
        if kind < 0.08:
            s = rng.choice(ACGT, size=L)
        else:
            c = big[int(rng.integers(len(big)))]
            c = np.frombuffer(c.tobytes().upper(), dtype=np.uint8)
            L = min(L, len(c))
            p = int(rng.integers(0, len(c) - L + 1))
            rate = float(rng.choice([0.0, 0.02, 0.08, 0.15]))
            s = _mutate(rng, c[p:p + L], rate * 0.4, rate * 0.6) if rate > 0 and L < 12_000 else c[p:p + L].copy()
            if rate > 0 and L >= 12_000:
                m = rng.random(L) < rate * 0.5; s[m] = rng.choice(ACGT, size=int(m.sum()))
            if len(s) == 0: s = rng.choice(ACGT, size=70)
            if rng.random() < 0.5: s = COMP[s[::-1]]
            if rng.random() < 0.05: s[int(rng.integers(len(s)))] = ord("N")
        out.append(s.tobytes())
    return out


@pytest.mark.parametrize("seed", range(int(os.environ.get("MM_FUZZ_SEEDS", "24"))))   # MM_FUZZ_SEEDS=500 for a longer hunt
def test_random_configurations_match_oracle(oracle_lib, tmp_path, seed):
    from metamaps_amd import capi
    rng = np.random.default_rng(9000 + int(os.environ.get("MM_FUZZ_SEED_BASE", "0")) + seed)
    k = int(rng.integers(8, 25)); w = int(rng.integers(2, 26))
    pi = float(rng.choice([70.0, 80.0, 85.0, 92.0])); min_len = int(rng.choice([100, 500, 1000, 3000]))
    contigs = _reference(rng)
    fa = str(tmp_path / "DB.fa")
    with open(fa, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(f">C{i}|kraken:taxid|{100 + i}|x\n".encode())
            for j in range(0, len(c), 70): f.write(c[j:j + 70].tobytes() + b"\n")
    reads = _reads(rng, contigs, 140)
    ctx = capi.Context(0)
    S = ctx.seqset([c.tobytes() for c in contigs]); R = ctx.seqset(reads)
    idx = ctx.index(S, k, w)
    oi = oracle_lib.index(fa, k, w)
    assert idx.freq_threshold == oi.freq_threshold
    M = ctx.map_batch(idx, R, k, w, pi=pi, min_read_len=min_len)
    off, rec = M.fetch()
    n_mapped = n_rec = 0
    for r, q in enumerate(reads):
        a, b = int(off[r]), int(off[r + 1])
        if len(q) < max(min_len, k, w):                          # computeMap.hpp:137
            assert a == b, (seed, r)
            continue
        m = oi.map_read(q, pi)["map"]
        rr = rec[a:b]
        assert b - a == len(m), (seed, k, w, pi, r, len(q), b - a, len(m))
        assert np.array_equal(rr["ref_contig"], m[:, 0]) and np.array_equal(rr["ref_start"], m[:, 1]), (seed, r)
        assert np.array_equal(rr["shared"], m[:, 3]) and np.array_equal(rr["sketch"], m[:, 4]) and np.array_equal(rr["strand"], m[:, 5]), (seed, r)
        n_mapped += len(m) > 0; n_rec += len(m)
    print(f"seed {seed}: k={k} w={w} pi={pi} min_len={min_len}: {n_mapped} reads mapped, {n_rec} records")
    assert n_mapped >= 10
    oi.close(); M.close(); idx.close(); R.close(); S.close(); ctx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MM_FUZZ_LONG_SEEDS", "6"))))
def test_random_long_reads_match_oracle(oracle_lib, tmp_path, seed):
    """the long-read paths against the oracle: sketches beyond 16 384 minimizers (segmented sketch sort), the wide slot table of
    the pre-filter (sketches beyond 13 000 hashes), the dense K5 path with its lazy strand tie-break; reads of 30 ... 260 kb next to
    short ones in one batch, small windows so that the sketches get long, references with duplications and related contigs"""
    from metamaps_amd import capi
    rng = np.random.default_rng(7000 + int(os.environ.get("MM_FUZZ_SEED_BASE", "0")) + seed)
    k = int(rng.integers(12, 21)); w = int(rng.integers(3, 12)); pi = float(rng.choice([75.0, 80.0, 88.0]))
    contigs = []
    root = rng.choice(ACGT, size=int(rng.integers(300_000, 500_000)))
    contigs.append(root)
    for div in (0.01, 0.06):
        c = root.copy(); m = rng.random(len(c)) < div; c[m] = rng.choice(ACGT, size=int(m.sum())); contigs.append(c)
    d = rng.choice(ACGT, size=int(rng.integers(150_000, 300_000)))
    seg = d[10_000:40_000].copy()
    at = int(rng.integers(60_000, len(d) - 40_000)); d[at:at + 30_000] = seg        # a 30 kb duplication inside one contig
    d[5_000:5_000 + int(rng.integers(50, 2_000))] = ord("N")
    contigs.append(d)
    fa = str(tmp_path / "DB.fa")
    with open(fa, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(f">L{i}|kraken:taxid|{200 + i}|x\n".encode() + c.tobytes() + b"\n")
    reads = []
    for i in range(14):
        c = contigs[int(rng.integers(len(contigs)))]
        L = int(min(len(c), np.exp(rng.uniform(np.log(30_000), np.log(260_000))))) if i < 10 else int(rng.integers(500, 8_000))
        p = int(rng.integers(0, len(c) - L + 1))
        s = c[p:p + L].copy()
        rate = float(rng.choice([0.0, 0.03, 0.1]))
        m = rng.random(L) < rate; s[m] = rng.choice(ACGT, size=int(m.sum()))
        if rate > 0:                                              # a few indels as well, in blocks
            for _ in range(int(rng.integers(0, 6))):
                q = int(rng.integers(0, len(s) - 50)); s = np.concatenate([s[:q], s[q + int(rng.integers(1, 40)):]])
        if rng.random() < 0.5: s = COMP[s[::-1]]
        reads.append(s.tobytes())
    ctx = capi.Context(0)
    S = ctx.seqset([c.tobytes() for c in contigs]); R = ctx.seqset(reads)
    idx = ctx.index(S, k, w)
    oi = oracle_lib.index(fa, k, w)
    M = ctx.map_batch(idx, R, k, w, pi=pi, min_read_len=1000)
    off, rec = M.fetch()
    st = M.stats()
    n_rec = 0
    for r, q in enumerate(reads):
        a, b = int(off[r]), int(off[r + 1])
        if len(q) < max(1000, k, w):
            assert a == b
            continue
        m = oi.map_read(q, pi)["map"]
        rr = rec[a:b]
        assert b - a == len(m), (seed, k, w, pi, r, len(q), b - a, len(m))
        assert np.array_equal(rr["ref_contig"], m[:, 0]) and np.array_equal(rr["ref_start"], m[:, 1]), (seed, r)
        assert np.array_equal(rr["shared"], m[:, 3]) and np.array_equal(rr["sketch"], m[:, 4]) and np.array_equal(rr["strand"], m[:, 5]), (seed, r)
        n_rec += len(m)
    print(f"seed {seed}: k={k} w={w} pi={pi}: {n_rec} records, largest sketch {max(int(x) for x in rec['sketch']) if len(rec) else 0}, kept/raw hits {st['sum_hits_kept']}/{st['sum_hits']}")
    assert n_rec >= 8
    oi.close(); M.close(); idx.close(); R.close(); S.close(); ctx.close()

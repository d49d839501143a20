"""Edge cases and size-independent properties of the HIP path (through the C ABI)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from metamaps_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _rand(rnd, n):
    return bytes(rnd.choice(b"ACGT") for _ in range(n))


def _mutate(rnd, s, rate):
    out = bytearray()
    for c in s:
        u = rnd.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(rnd.choice(b"ACGT")); continue
        out.append(c)
        if u < rate:
            out.append(rnd.choice(b"ACGT"))
    return bytes(out)


@pytest.fixture(scope="module")
def small_world(ctx, oracle_lib, tmp_path_factory):
    rnd = random.Random(77)
    genomes = [_rand(rnd, 120_000) for _ in range(4)]
    d = tmp_path_factory.mktemp("edge")
    fa = d / "ref.fa"
    with open(fa, "wb") as f:
        for i, g in enumerate(genomes):
            f.write(b">C%d|kraken:taxid|%d|X\n" % (i, 100 + i) + g + b"\n")
        f.write(b">tiny|kraken:taxid|9|X\nACGTAC\n")            # shorter than k: metadata only
    S = ctx.seqset(genomes + [b"ACGTAC"])
    idx = ctx.index(S, 16, 10)
    oi = oracle_lib.index(str(fa), 16, 10)
    return {"genomes": genomes, "idx": idx, "oi": oi, "rnd": rnd}


def _check_against_oracle(ctx, world, reads, min_read_len=1000):
    R = ctx.seqset(reads)
    M = ctx.map_batch(world["idx"], R, 16, 10, pi=80.0, min_read_len=min_read_len)
    off, rec = M.fetch()
    for r, q in enumerate(reads):
        got = rec[off[r]:off[r + 1]]
        if len(q) < max(min_read_len, 16, 10):
            assert len(got) == 0
            continue
        m = world["oi"].map_read(q, 80.0)["map"]
        assert len(got) == len(m), (r, len(q))
        assert np.array_equal(got["ref_contig"], m[:, 0]) and np.array_equal(got["ref_start"], m[:, 1]), r
        assert np.array_equal(got["shared"], m[:, 3]) and np.array_equal(got["sketch"], m[:, 4]) and np.array_equal(got["strand"], m[:, 5]), r
    st = M.stats()
    M.close(); R.close()
    return st


def test_ragged_and_degenerate_reads(ctx, small_world):
    g, rnd = small_world["genomes"], small_world["rnd"]
    reads = [
        b"",                                              # empty record
        b"ACGT",                                          # shorter than k
        g[0][1000:1900],                                  # below --minReadLen
        b"N" * 3000,                                      # every k-mer symmetric: sketch size 0 (computeMap.hpp:302)
        b"A" * 4000,                                      # homopolymer: one hash
        b"ACGT" * 1000,                                   # palindromic tandem repeat
        g[1][5000:9000],                                  # exact copy
        g[1][5000:9000].lower(),                          # lower case is upper-cased first
        _mutate(rnd, g[2][20000:32000], 0.12),            # ONT-like
        g[3][100:3100][:1500] + b"N" * 200 + g[3][1800:3100],   # N run inside
        g[0][-2500:],                                     # hangs over the contig end
        g[0][:2500],                                      # at the contig start
        _rand(rnd, 5000),                                 # unrelated
        _mutate(rnd, g[2][40000:90000], 0.10),            # 50 kb read: large sketch (LDS size classes, compact L2 limits)
    ]
    st = _check_against_oracle(ctx, small_world, reads)
    assert st["n_reads"] == len(reads) and st["n_reads_mapped"] >= 6


def test_all_reads_too_short_and_empty_batch(ctx, small_world):
    st = _check_against_oracle(ctx, small_world, [b"ACGT", b"", small_world["genomes"][0][:900]])
    assert st["n_reads_long_enough"] == 0 and st["n_mappings"] == 0
    R = ctx.seqset([])
    M = ctx.map_batch(small_world["idx"], R, 16, 10)
    off, rec = M.fetch()
    assert len(off) == 1 and len(rec) == 0
    M.add_qualities(16)
    M.close(); R.close()


def test_min_read_len_and_identity_threshold_flags(ctx, small_world):
    g, rnd = small_world["genomes"], small_world["rnd"]
    reads = [_mutate(rnd, g[i % 4][3000 * i:3000 * i + 2200], 0.1) for i in range(12)]
    _check_against_oracle(ctx, small_world, reads, min_read_len=2000)
    R = ctx.seqset(reads)
    for pi in (70.0, 90.0, 99.0):                        # threshold only changes which candidates survive
        M = ctx.map_batch(small_world["idx"], R, 16, 10, pi=pi, min_read_len=1000)
        off, rec = M.fetch()
        for r, q in enumerate(reads):
            m = small_world["oi"].map_read(q, pi)["map"]
            assert np.array_equal(rec[off[r]:off[r + 1]]["ref_start"], m[:, 1]), (pi, r)
        M.close()
    R.close()


def test_results_do_not_depend_on_batching(ctx):
    """Size-independent property at a larger scale: mapping a batch == mapping its parts (reads are independent,
    computeMap.hpp:180-200), and mapping twice is bit-identical."""
    ref = ctx.synth_reference(seed=11, n_species=64, strains_per_species=4, genome_len=400_000, strain_divergence=0.02, genus_divergence=0.1)
    idx = ctx.index(ref, 16, 8)
    parts, recs = [], []
    for i, n in enumerate((3000, 1700)):
        r, _ = ctx.synth_reads(ref, seed=50 + i, n_reads=n, read_len=7000 + 2500 * i, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=60)
        parts.append(r)
    whole = ctx.seqset([p.fetch(i, int(L)) for p in parts for i, L in enumerate(p.lengths())])
    Mw = ctx.map_batch(idx, whole, 16, 8); Mw.add_qualities(16)
    offw, recw = Mw.fetch()
    Mw2 = ctx.map_batch(idx, whole, 16, 8); Mw2.add_qualities(16)
    offw2, recw2 = Mw2.fetch()
    assert np.array_equal(offw, offw2) and recw.tobytes() == recw2.tobytes()
    base = 0
    for p in parts:
        M = ctx.map_batch(idx, p, 16, 8); M.add_qualities(16)
        off, rec = M.fetch()
        n = p.count
        a, b = int(offw[base]), int(offw[base + n])
        assert np.array_equal(off, offw[base:base + n + 1] - a)
        sub = recw[a:b].copy(); sub["read"] -= base
        assert sub.tobytes() == rec.tobytes()
        # mapping qualities of a read sum to one (mapWrap.h:300-306)
        sums = np.add.reduceat(rec["mapq"], off[:-1][np.diff(off) > 0])
        assert np.allclose(sums, 1.0, atol=1e-9)
        base += n
        M.close()
    assert len(recw) > 10_000
    Mw.close(); Mw2.close(); whole.close(); idx.close(); ref.close()


def test_seqset_save_load_round_trip(tmp_path):
    """packed reference file: every base, exception run (N, lower case handled at upload) and length survives"""
    from metamaps_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(3)
    seqs = [bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8)) for n in (1, 15, 16, 17, 1000, 40_001)]
    seqs[4] = seqs[4][:100] + b"N" * 50 + b"RYK" + seqs[4][153:]
    seqs.append(b"")
    S = ctx.seqset(seqs)
    path = str(tmp_path / "s.seqset")
    S.save(path)
    T = ctx.load_seqset(path)
    assert T.count == S.count and T.total_bases == S.total_bases and np.array_equal(T.lengths(), S.lengths())
    for i, q in enumerate(seqs):
        assert T.fetch(i, len(q)) == S.fetch(i, len(q)) == q.upper()
    open(path, "r+b").write(b"XX")                      # a damaged file is refused
    with pytest.raises(Exception):
        ctx.load_seqset(path)
    S.close(); T.close(); ctx.close()


def test_seqset_slice_and_concat(ctx):
    """mm_seqset_slice / mm_seqset_concat (device-side, no repacking): the sequences, their exception runs (N, IUPAC, lower case
    upper-cased) and the minimizers computed from them equal those of sets uploaded from the same strings"""
    from metamaps_amd import capi
    rng = np.random.default_rng(12)
    seqs = []
    for i in range(37):
        n = int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 100, 1000, 4097]))
        s = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes())
        if n > 40 and i % 3 == 0:
            a = int(rng.integers(0, n - 20)); s[a:a + 17] = b"N" * 17
        if n > 40 and i % 4 == 1:
            s[5:9] = b"RYnn"
        if n > 16 and i % 5 == 2:
            s[n - 1:n] = b"N"                                      # an exception on the last base of a sequence
        seqs.append(bytes(s))
    whole = ctx.seqset(seqs)
    want = [s.upper() for s in seqs]
    k, w = 16, 5
    off_w, h_w, wp_w, st_w = ctx.minimizers(whole, k, w)
    cuts = [(0, 37), (0, 0), (5, 1), (3, 20), (20, 17), (36, 1), (37, 0)]
    for a, n in cuts:
        sl = whole.slice(a, n)
        assert sl.count == n and list(sl.lengths()) == [len(x) for x in want[a:a + n]]
        for j in range(n):
            assert sl.fetch(j, len(want[a + j])) == want[a + j], (a, n, j)
        off, h, wp, st = ctx.minimizers(sl, k, w)
        lo, hi = int(off_w[a]), int(off_w[a + n])
        assert np.array_equal(np.diff(off), np.diff(off_w[a:a + n + 1]))
        assert np.array_equal(h, h_w[lo:hi]) and np.array_equal(wp, wp_w[lo:hi]) and np.array_equal(st, st_w[lo:hi])
        sl.close()
    parts = [whole.slice(0, 7), whole.slice(7, 0), whole.slice(7, 13), ctx.seqset(seqs[20:30]), whole.slice(30, 7)]
    cat = capi.SeqSet.concat(ctx, parts)
    assert cat.count == 37 and cat.total_bases == whole.total_bases
    for j in range(37):
        assert cat.fetch(j, len(want[j])) == want[j], j
    off, h, wp, st = ctx.minimizers(cat, k, w)
    assert np.array_equal(off, off_w) and np.array_equal(h, h_w) and np.array_equal(wp, wp_w) and np.array_equal(st, st_w)
    idx_a, idx_b = ctx.index(whole, k, w), ctx.index(cat, k, w)
    ea, eb = idx_a.entries(), idx_b.entries()
    assert all(np.array_equal(x, y) for x, y in zip(ea, eb)) and len(ea[0]) > 100
    for x in parts + [cat, whole]:
        x.close()
    idx_a.close(); idx_b.close()


def test_release_cached_returns_memory_to_the_driver(ctx):
    """mm_ctx_release_cached: what an index build leaves in the context's cache (its temporaries) goes back to the driver on request —
    a host does this before worker contexts start beside the resident indexes, instead of waiting for an allocation to fail"""
    ref = ctx.synth_reference(seed=3, n_species=24, strains_per_species=4, genome_len=500_000, strain_divergence=0.02, genus_divergence=0.08)
    idx = ctx.index(ref, 16, 8)
    idx.close()
    ctx.synchronize()
    free_before = ctx.device_info()["hbm_free"]
    ctx.release_cached()
    free_after = ctx.device_info()["hbm_free"]
    assert free_after > free_before + (64 << 20)                 # the sort buffers of a 48 Mbp index alone are > 100 MB
    idx = ctx.index(ref, 16, 8)                                  # and the context works as before
    assert idx.info()["n_entries"] > 1_000_000
    idx.close(); ref.close()


def test_long_sequences_are_packed_in_pieces(ctx, tmp_path):
    """mm_seqset_upload packs a sequence in pieces of 4 Mbases on several threads; a run of non-ACGT characters that crosses a piece boundary
    is one exception run again (counted in the persistent form of the set, mm_seqset_save), and the sequence comes back as it went in"""
    from metamaps_amd import capi
    rng = np.random.default_rng(5)
    P = 4 << 20
    n = 2 * P + 12345
    s = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes())
    s[P - 7:P + 9] = b"N" * 16                                   # across the first boundary
    s[2 * P - 1:2 * P] = b"R"                                    # the last base of a piece ...
    s[2 * P:2 * P + 1] = b"R"                                    # ... and the first of the next: one run of two
    s[2 * P + 1:2 * P + 2] = b"Y"                                # another byte right behind: its own run
    s[100:103] = b"nnn"
    short = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 1000).tobytes())
    S = ctx.seqset([short, bytes(s), short])
    assert S.fetch(1, n) == bytes(s).upper() and S.fetch(0, 1000) == short and S.fetch(2, 1000) == short
    S.save(str(tmp_path / "a.seqset"))
    L = ctx.load_seqset(str(tmp_path / "a.seqset"))
    assert L.fetch(1, n) == bytes(s).upper() and L.total_bases == S.total_bases
    import struct
    raw = open(str(tmp_path / "a.seqset"), "rb").read()
    n_seq, total, n_words, n_exc = struct.unpack_from("<4q", raw, 16)
    assert n_seq == 3 and n_exc == 4                             # nnn | 16 N | RR | Y — not 5 (a run split at the boundary) or 6
    S.close(); L.close()

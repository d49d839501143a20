"""GPU parity tests: every stage of the HIP path, reached through the C ABI (metamaps_amd/capi.py →
libmetamaps_hip.so), against the oracle on the same seeded inputs.  Integer results must be bit-exact."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMP = {65: 84, 84: 65, 67: 71, 71: 67}


@pytest.fixture(scope="module")
def ctx():
    from metamaps_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def adversarial_sequences(seed, n, kmax=21):
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        L = rnd.randint(kmax, 700)
        mode = rnd.random()
        if mode < 0.3:
            s = bytes(rnd.choice(b"ACGT") for _ in range(L))
        elif mode < 0.5:
            unit = bytes(rnd.choice(b"ACGT") for _ in range(rnd.randint(1, 6)))
            s = (unit * (L // len(unit) + 1))[:L]
        elif mode < 0.65:
            s = bytes(rnd.choice(b"ACGTNNN") for _ in range(L))
        elif mode < 0.8:
            h = bytes(rnd.choice(b"ACGT") for _ in range(L // 2))
            s = h + bytes(COMP[c] for c in reversed(h))
        elif mode < 0.9:
            s = bytes(rnd.choice(b"acgtRYn") for _ in range(L))
        else:
            s = bytes(rnd.choice(b"AC") for _ in range(L))
        out.append(s)
    out += [b"ACGT", b"A" * 40, b"N" * 50, b"ACGTACGTACGTACGTA", bytes(random.Random(3).choice(b"ACGT") for _ in range(9000)),
            b"AC" * 3000, b"ACG" * 2500 + b"T" * 3000, b"A" * 6000]      # > 1024 minimizers per 2048-position tile (staging overflow path)
    return out


def test_seqset_roundtrip(ctx):
    seqs = adversarial_sequences(1, 50)
    s = ctx.seqset(seqs)
    assert s.count == len(seqs)
    assert list(s.lengths()) == [len(q) for q in seqs]
    for i, q in enumerate(seqs):
        assert s.fetch(i, len(q)) == q.upper()
    s.close()


@pytest.mark.parametrize("k,w", [(16, 8), (16, 13), (16, 1), (16, 20), (5, 3), (21, 11), (32, 16), (8, 100)])
def test_minimizers_match_oracle(ctx, oracle_lib, k, w):
    seqs = adversarial_sequences(100 + k * 31 + w, 120)
    s = ctx.seqset(seqs)
    off, h, wp, st = ctx.minimizers(s, k, w)
    nonempty = 0
    for i, q in enumerate(seqs):
        oh, ow, os_ = oracle_lib.minimizers(q, k, w) if len(q) >= max(k, w) else ([], [], [])
        a, b = int(off[i]), int(off[i + 1])
        assert b - a == len(oh), (i, k, w, q[:60])
        assert np.array_equal(h[a:b], oh) and np.array_equal(wp[a:b], ow) and np.array_equal(st[a:b], os_), (i, k, w)
        nonempty += len(oh) > 0
    assert nonempty > 50
    s.close()


def _read_fasta(path):
    names, seqs, cur = [], [], []
    for ln in open(path, "rb"):
        if ln.startswith(b">"):
            if names:
                seqs.append(b"".join(cur))
            names.append(ln[1:].split()[0].decode()); cur = []
        else:
            cur.append(ln.strip())
    seqs.append(b"".join(cur))
    return names, seqs


def _read_fastq(path):
    names, seqs = [], []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().strip(); f.readline(); f.readline()
            names.append(h[1:].split()[0].decode()); seqs.append(s)
    return names, seqs


@pytest.fixture(scope="module")
def mini(tmp_path_factory, oracle_lib):
    """BASELINE config 0 scaled for test time: 10 genomes x 60 kb, 150 reads of 3 kb."""
    from metamaps_amd import synth
    d = tmp_path_factory.mktemp("mini")
    db = synth.make_db(str(d / "db"), n_genomes=10, genome_len=60_000, seed=7)
    rd = synth.make_reads(db, str(d / "reads.fq"), n_reads=150, read_len=3000, seed=3)
    return {"dir": str(d), "db": db, "reads": rd["path"]}


@pytest.mark.parametrize("k,w,part_max", [(16, 13, None), (16, 8, None), (16, 8, 20000)])
def test_index_matches_oracle(ctx, oracle_lib, mini, k, w, part_max, monkeypatch):
    if part_max:                                       # force the partitioned (>2^31-entry) sort path
        monkeypatch.setenv("MM_INDEX_PART_MAX", str(part_max))
    names, contigs = _read_fasta(mini["db"].fasta)
    S = ctx.seqset(contigs)
    idx = ctx.index(S, k, w)
    oi = oracle_lib.index(mini["db"].fasta, k, w)
    info = idx.info()
    assert info["n_contigs"] == oi.n_contigs == len(contigs)
    assert info["n_entries"] == oi.n
    assert info["n_unique_hashes"] == oi.n_unique
    h, c, wp, st = idx.entries()
    oh, oc, ow, os_ = oi.dump()
    assert np.array_equal(h, oh) and np.array_equal(c, oc) and np.array_equal(wp, ow) and np.array_equal(st, os_)
    assert idx.freq_threshold == oi.freq_threshold
    if part_max:                                       # the hash-ordered table must work too: map a few reads
        rnames, reads = _read_fastq(mini["reads"])
        R = ctx.seqset(reads[:40])
        M = ctx.map_batch(idx, R, k, w)
        off, rec = M.fetch()
        for r, q in enumerate(reads[:40]):
            if len(q) >= 1000:
                m = oi.map_read(q)["map"]
                assert np.array_equal(rec[off[r]:off[r + 1]]["ref_start"], m[:, 1]), r
        M.close(); R.close()
    oi.close(); idx.close(); S.close()


def test_l2_workgroups_in_position_order_give_the_same_records(ctx, mini, monkeypatch):
    """K5's workgroups are launched in the order of their first candidate's position (l2_group_keys_kernel + radix sort; from 2 048 workgroups
    on, MM_L2_GROUP_SORT_MIN lowers that): results are indexed by candidate, so the records are those of the unsorted launch, byte for byte"""
    names, contigs = _read_fasta(mini["db"].fasta)
    rnames, reads = _read_fastq(mini["reads"])
    S, R = ctx.seqset(contigs), ctx.seqset(reads)
    idx = ctx.index(S, 16, 8)
    out = {}
    for tag, env in (("sorted", {"MM_L2_GROUP_SORT_MIN": "1"}), ("plain", {"MM_L2_NO_GROUP_SORT": "1"})):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        M = ctx.map_batch(idx, R, 16, 8); M.add_qualities(16)
        off, rec = M.fetch()
        out[tag] = (off.copy(), rec.tobytes(), len(rec))
        M.close()
        for k_ in env:
            monkeypatch.delenv(k_)
    assert np.array_equal(out["sorted"][0], out["plain"][0]) and out["sorted"][1] == out["plain"][1] and out["plain"][2] > 100
    idx.close(); R.close(); S.close()


def test_l2_launches_side_by_side_give_the_same_records(ctx, mini, monkeypatch):
    """the two launches of K5's 10 kb class (four-wave and two-wave workgroups) run side by side, the second on the context's auxiliary stream, taking their
    scratch slots from one pool: the records are those of the launches one behind the other (MM_L2_ONE_STREAM=1), byte for byte, also over repeated batches"""
    names, contigs = _read_fasta(mini["db"].fasta)
    rnames, reads = _read_fastq(mini["reads"])
    S, R = ctx.seqset(contigs), ctx.seqset(reads)
    idx = ctx.index(S, 16, 8)
    out = {}
    for tag, env in (("side_by_side", {}), ("one_stream", {"MM_L2_ONE_STREAM": "1"})):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        for rep in range(3):
            M = ctx.map_batch(idx, R, 16, 8); M.add_qualities(16)
            off, rec = M.fetch()
            st = M.stats()
            cur = (off.copy(), rec.tobytes(), len(rec))
            assert tag not in out or (np.array_equal(out[tag][0], cur[0]) and out[tag][1] == cur[1])
            out[tag] = cur
            M.close()
        for k_ in env:
            monkeypatch.delenv(k_)
    assert np.array_equal(out["side_by_side"][0], out["one_stream"][0]) and out["side_by_side"][1] == out["one_stream"][1] and out["one_stream"][2] > 100
    # tiny batches (a handful of waves per launch, far below the slot pool's cap): the pool is sized for the waves of BOTH launches, and with a
    # pool of eight slots (MM_L2_SLOTS, one per XCD) every wave of both launches queues for a slot — the records are the big batch's, read by read
    full = ctx.map_batch(idx, R, 16, 8); full.add_qualities(16)
    off_f, rec_f = full.fetch(); off_f, rec_f = off_f.copy(), rec_f.copy(); full.close()
    for env in ({}, {"MM_L2_SLOTS": "8"}):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        for first in (0, 7, 19):
            sub = ctx.seqset(reads[first:first + 3])
            Ms = ctx.map_batch(idx, sub, 16, 8); Ms.add_qualities(16)
            o, r = Ms.fetch()
            for i in range(3):
                want = rec_f[off_f[first + i]:off_f[first + i + 1]].copy(); got = r[o[i]:o[i + 1]].copy()
                want["read"] = 0; got["read"] = 0                   # (the read number is the batch's)
                assert want.tobytes() == got.tobytes(), (env, first, i)
            Ms.close(); sub.close()
        for k_ in env:
            monkeypatch.delenv(k_)
    idx.close(); R.close(); S.close()


@pytest.mark.parametrize("k,w,thr", [(16, 8, None), (16, 13, 3)])
def test_index_stored_and_loaded_equals_built(ctx, mini, tmp_path, k, w, thr):
    """persistent device index (mm_index_save / mm_index_load; mapWrap.h:358-405, :443-554): the loaded index has the
    built one's entries, duplicate distances, histogram and threshold, plans the same chunks and maps to the same records;
    foreign and truncated files are refused"""
    from metamaps_amd import capi
    names, contigs = _read_fasta(mini["db"].fasta)
    rnames, reads = _read_fastq(mini["reads"])
    S = ctx.seqset(contigs)
    built = ctx.index(S, k, w)
    if thr:
        built.set_freq_threshold(thr)
    path = str(tmp_path / "chunk.mmidx")
    built.save(path)
    assert os.path.getsize(path) > built.info()["n_entries"] * 8
    loaded = ctx.load_index(path)
    assert loaded.info() == built.info()
    for a, b in zip(built.entries(), loaded.entries()):
        assert np.array_equal(a, b)
    for a, b in zip(built.dup_neighbours(), loaded.dup_neighbours()):
        assert np.array_equal(a, b)
    for a, b in zip(built.freq_hist(), loaded.freq_hist()):
        assert np.array_equal(a, b)
    assert built.plan_chunks(1_000_000) == loaded.plan_chunks(1_000_000) and len(built.plan_chunks(1_000_000)) > 1
    R = ctx.seqset(reads)
    Mb, Ml = ctx.map_batch(built, R, k, w), ctx.map_batch(loaded, R, k, w)
    (ob, rb), (ol, rl) = Mb.fetch(), Ml.fetch()
    assert np.array_equal(ob, ol) and rb.tobytes() == rl.tobytes() and len(rb) > 100
    Mb.close(); Ml.close(); R.close(); loaded.close()
    data = open(path, "rb").read()
    # truncated, foreign magic, foreign version — and damage INSIDE intact framing (since format version 2 every array carries a checksum taken on the device, and the
    # element counts are held against the header before anything is allocated): one flipped bit in the middle of the entries, in the hash table and in the last
    # array; an entry count in the header that no longer matches the arrays
    def flipped(at):
        b = bytearray(data); b[at] ^= 0x10; return bytes(b)
    n_entries_at = 8 + 32 + 8                                        # dims[1] = N behind the magic and the eight header words
    bad_n = data[:n_entries_at] + (int.from_bytes(data[n_entries_at:n_entries_at + 8], "little") + 3).to_bytes(8, "little") + data[n_entries_at + 8:]
    for bad in (data[: len(data) - 5], data[: len(data) // 3], b"MMSEQSET" + data[8:], data[:8] + b"\x09" + data[9:],
                flipped(len(data) // 5), flipped(len(data) // 2), flipped(len(data) - 40), bad_n):
        open(path, "wb").write(bad)
        with pytest.raises(capi.MMError):
            ctx.load_index(path)
    open(path, "wb").write(data)
    again = ctx.load_index(path); assert again.info() == built.info(); again.close()      # (the undamaged bytes still load)
    with pytest.raises(capi.MMError):
        ctx.load_index(str(tmp_path / "absent.mmidx"))
    built.close(); S.close()


@pytest.mark.parametrize("k,w,force_thr,eager", [(16, 13, None, True), (16, 8, None, True), (16, 8, 3, True), (16, 8, None, False), (16, 8, None, "redo")])
def test_mapping_stages_match_oracle(ctx, oracle_lib, mini, k, w, force_thr, eager, monkeypatch):
    from metamaps_amd import capi
    if eager is True:   # resolve every duplicated-hash strand up front so that the whole sketch can be compared; the default
        monkeypatch.setenv("MM_EAGER_TIEBREAK", "1")   # resolves only reads whose strand vote is undecided without it
    elif eager == "redo":   # ... which is rare: force that path (host resolution after L2, candidates redone)
        monkeypatch.setenv("MM_FORCE_AMB_REDO", "1")
    monkeypatch.setenv("MM_NO_HIT_FILTER", "1")        # compare the raw seed-hit list; the filter has its own test below
    names, contigs = _read_fasta(mini["db"].fasta)
    rnames, reads = _read_fastq(mini["reads"])
    S = ctx.seqset(contigs)
    R = ctx.seqset(reads)
    idx = ctx.index(S, k, w)
    oi = oracle_lib.index(mini["db"].fasta, k, w)
    if force_thr is not None:      # exercise the frequency filter: the tiny DB never reaches the 0.001 % rule
        idx.set_freq_threshold(force_thr)
        import ctypes as C
        # the oracle index exposes no setter; skip hit comparison details by rebuilding expectations below
    M = ctx.map_batch(idx, R, k, w, pi=80.0, min_read_len=1000)
    M.add_qualities(k)
    st = M.stats()
    print("tie-break mode", eager, "ambiguous reads", st["n_ambiguous_sketch_reads"], "candidates redone", st["n_l2_wide_redo"])
    sk_off, sk_h, sk_s = M.debug_sketch()
    hit_off, hit_c, hit_w = M.debug_hits()
    cand_off, cand = M.debug_candidates()
    l2 = M.debug_l2(len(cand))
    mh = M.debug_min_hits()
    rec_off, rec = M.fetch()
    n_checked = n_mapped = 0
    for r, q in enumerate(reads):
        if len(q) < max(1000, k, w):
            assert sk_off[r + 1] == sk_off[r] and rec_off[r + 1] == rec_off[r]
            continue
        o = oi.map_read(q, 80.0)
        a, b = int(sk_off[r]), int(sk_off[r + 1])
        assert np.array_equal(sk_h[a:b], o["sketch_hash"]), r
        if eager is True:
            assert np.array_equal(sk_s[a:b], o["sketch_strand"]), r
        assert mh[r] == o["min_hits"] or b == a, r
        if force_thr is None:
            a, b = int(hit_off[r]), int(hit_off[r + 1])
            assert np.array_equal(hit_c[a:b], o["hit_contig"]) and np.array_equal(hit_w[a:b], o["hit_wpos"]), r
            a, b = int(cand_off[r]), int(cand_off[r + 1])
            assert np.array_equal(cand[a:b], o["cand"]), r
            got = l2[a:b]; exp = o["l2"]
            assert np.array_equal(got[:, 0], exp[:, 0]), r                                              # contig
            ok = got[:, 5] == 1                                    # candidates that pass the identity filter: every field exact
            assert np.array_equal(got[ok][:, [1, 2, 3, 4]], exp[ok][:, [1, 2, 3, 4]]), r                # meanPos, shared, optBeg, optEnd
            assert np.all(got[~ok][:, 2] <= exp[~ok][:, 2]), r     # dropped ones: the skip-ahead may stop below the true maximum
            assert int(ok.sum()) == len(o["map"]), r
            a, b = int(rec_off[r]), int(rec_off[r + 1])
            m = o["map"]
            assert b - a == len(m), r
            rr = rec[a:b]
            assert np.array_equal(rr["ref_contig"], m[:, 0]) and np.array_equal(rr["ref_start"], m[:, 1]), r
            assert np.array_equal(rr["shared"], m[:, 3]) and np.array_equal(rr["sketch"], m[:, 4]) and np.array_equal(rr["strand"], m[:, 5]), r
            if len(m):
                n_mapped += 1
                # mapping qualities through the reference's own text round trip
                lines = []
                for x in m:
                    ident, _ = oracle_lib.identity(int(x[3]), int(x[4]), k)
                    lines.append(f"q {len(q)} 0 {len(q) - 1} + c 1 {x[1]} {x[2]} {ident:g} {x[3]} {x[4]}")
                exp_mq = np.array([float(l.split(" ")[13]) for l in oracle_lib.add_mapq(k, lines)])
                got_mq = np.array([float(f"{v:g}") for v in rr["mapq"]])
                assert np.allclose(got_mq, exp_mq, rtol=1e-5, atol=1e-300), (r, got_mq, exp_mq)
        n_checked += 1
    assert n_checked > 100
    if force_thr is None:
        assert n_mapped > 80
        assert st["n_reads_mapped"] == n_mapped
    else:
        # with the filter on, hits must be a subset rule: every kept hash occurs < thr times; verify via counts
        assert st["sum_hits"] < 10**9
    oi.close(); M.close(); idx.close(); R.close(); S.close()


def test_em_matches_oracle(ctx, oracle_lib, mini, tmp_path):
    """EM iterations on the device vs the oracle's doEM on the same mappings file."""
    import subprocess, json, orc
    from metamaps_amd import emhost
    prefix = str(tmp_path / "out")
    subprocess.run([orc.CLI, "mapDirectly", "--all", "-r", mini["db"].fasta, "-q", mini["reads"], "-o", prefix], check=True,
                   capture_output=True, timeout=600)
    p = subprocess.run([orc.CLI, "classify", "--DB", mini["db"].dir, "--mappings", prefix], check=True, capture_output=True, timeout=600)
    info = json.loads(p.stderr.decode().strip().splitlines()[-1])
    prob = emhost.load_problem(prefix, mini["db"].dir)
    em = ctx.em(prob.read_off, prob.taxon, prob.mapq, prob.inv_nloc, len(prob.taxa))
    f, lls = emhost.run_em(lambda f: em.iterate_allreduce(f), len(prob.taxa))
    assert len(lls) == info["iterations"]
    assert np.allclose(lls, info["ll"], rtol=1e-12)
    post, best = em.posteriors(f)
    # reads2Taxon must be identical, posteriors within 1e-5 of the %f text
    exp_r2t = [l.rstrip("\n").split("\t") for l in open(prefix + ".EM.reads2Taxon")]
    got = [(prob.read_ids[r], prob.taxa[prob.taxon[best[r]]]) for r in range(len(prob.read_ids))]
    assert got == [tuple(x) for x in exp_r2t[:len(got)]]
    exp_post = np.array([float(l.split(" ")[13]) for l in open(prefix + ".EM")])
    assert np.allclose(post, exp_post, atol=1e-5)
    em.close()


def test_l2_skip_ahead_equals_full_slide(ctx, monkeypatch):
    """The exact skip-ahead of K5 against the plain full slide (which the tests above pin to the oracle) on a
    device-generated workload large enough to hit every path: thousands of candidates, rebuilds, block skips."""
    ref = ctx.synth_reference(seed=5, n_species=48, strains_per_species=4, genome_len=400_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, truth = ctx.synth_reads(ref, seed=9, n_reads=3000, read_len=8000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=40)
    idx = ctx.index(ref, 16, 8)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MM_L2_FULL", mode)
        M = ctx.map_batch(idx, reads, 16, 8)
        off, rec = M.fetch()
        res[mode] = (off.copy(), rec.copy(), M.stats())
        M.close()
    monkeypatch.delenv("MM_L2_FULL")
    assert np.array_equal(res["0"][0], res["1"][0])
    assert np.array_equal(res["0"][1], res["1"][1])
    s_skip, s_full = res["0"][2], res["1"][2]
    assert s_full["n_mappings"] > 5000 and s_skip["n_l2_rebuilds"] > 0
    assert s_skip["sum_l2_evals"] < s_full["sum_l2_evals"]
    idx.close(); reads.close(); ref.close()


def test_l2_device_made_workgroups_equal_host_made(ctx, monkeypatch):
    """K5's workgroups of the 10 kb class are put together by l2_group_kernel; MM_L2_HOST_GROUPS=1 makes them in the host loop, and
    MM_L2_NO_SMALL_GROUPS=1 sends remainders of one or two candidates to four-wave workgroups: identical records and work counters.
    Mixed read lengths, so that the host's classes (longer sketches) and the device's lists are both in use in one batch."""
    ref = ctx.synth_reference(seed=15, n_species=40, strains_per_species=5, genome_len=300_000, strain_divergence=0.02, genus_divergence=0.08)
    idx = ctx.index(ref, 16, 8)
    for read_len, len_min in ((7000, None), (30000, 800)):
        kw = dict(seed=19, n_reads=2500, read_len=read_len, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=30)
        if len_min: kw["read_len_min"] = len_min
        reads, _ = ctx.synth_reads(ref, **kw)
        res = {}
        for mode in ("device", "host", "host_nosmall", "device_nosmall"):
            if mode.startswith("host"): monkeypatch.setenv("MM_L2_HOST_GROUPS", "1")
            if mode.endswith("nosmall"): monkeypatch.setenv("MM_L2_NO_SMALL_GROUPS", "1")
            M = ctx.map_batch(idx, reads, 16, 8)
            off, rec = M.fetch()
            res[mode] = (off.copy(), rec.copy(), M.stats())
            M.close()
            monkeypatch.delenv("MM_L2_HOST_GROUPS", raising=False); monkeypatch.delenv("MM_L2_NO_SMALL_GROUPS", raising=False)
        for mode in ("host", "host_nosmall", "device_nosmall"):
            assert np.array_equal(res["device"][0], res[mode][0]), mode
            assert np.array_equal(res["device"][1], res[mode][1]), mode
            for key in ("n_candidates", "sum_l2_stream_entries", "sum_l2_evals", "n_l2_rebuilds"):
                assert res["device"][2][key] == res[mode][2][key], (mode, key)
        assert res["device"][2]["n_mappings"] > 2000
        reads.close()
    idx.close(); ref.close()


def test_phased_map_batch_calls_back_once_and_changes_nothing(ctx):
    """mm_map_batch_phased: the callbacks (sketches complete / last big kernel enqueued) run once each, in order, on the calling thread; same records"""
    import threading
    ref = ctx.synth_reference(seed=25, n_species=12, strains_per_species=3, genome_len=200_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=29, n_reads=600, read_len=6000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=10)
    idx = ctx.index(ref, 16, 8)
    calls = []
    A = ctx.map_batch(idx, reads, 16, 8)
    B = ctx.map_batch(idx, reads, 16, 8, at_seed_stage=lambda: calls.append((1, threading.get_ident())), at_last_kernel=lambda: calls.append((2, threading.get_ident())))
    assert calls == [(1, threading.get_ident()), (2, threading.get_ident())]
    (oa, ra), (ob, rb) = A.fetch(), B.fetch()
    assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and len(ra) > 500
    A.close(); B.close(); idx.close(); reads.close(); ref.close()


def test_wide_slot_table_of_the_hit_filter_changes_no_result(ctx, monkeypatch):
    """Reads beyond ~32 kb are filtered with 32 768 slots counted from occ[] instead of 8 192 slots from occ16[]: fewer chance hits
    survive, candidates and records stay what they are with the narrow table (MM_HF_WIDE_FROM=0), with no filter at all, and with a
    tiny stage (the write pass re-filters with the same table)."""
    ref = ctx.synth_reference(seed=45, n_species=30, strains_per_species=4, genome_len=500_000, strain_divergence=0.02, genus_divergence=0.08)
    idx = ctx.index(ref, 12, 8)                                   # k = 12: plenty of chance hits, so that the filter has work to do
    reads, _ = ctx.synth_reads(ref, seed=49, n_reads=300, read_len=120_000, read_len_min=3_000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=20)
    res = {}
    for mode in ("wide", "narrow", "none", "wide_tiny_stage"):
        if mode.startswith("wide"): monkeypatch.setenv("MM_HF_WIDE_FROM", "4000")   # (default: 13000 hashes, reads from ~58 kb on)
        if mode == "narrow": monkeypatch.setenv("MM_HF_WIDE_FROM", "0")
        if mode == "none": monkeypatch.setenv("MM_NO_HIT_FILTER", "1")
        if mode == "wide_tiny_stage": monkeypatch.setenv("MM_HF_STAGE_CAP", "16")
        M = ctx.map_batch(idx, reads, 12, 8)
        cand_off, cand = M.debug_candidates()
        off, rec = M.fetch()
        res[mode] = (cand_off.copy(), cand.copy(), off.copy(), rec.copy(), M.stats())
        M.close()
        for v in ("MM_HF_WIDE_FROM", "MM_NO_HIT_FILTER", "MM_HF_STAGE_CAP"): monkeypatch.delenv(v, raising=False)
    for mode in ("narrow", "none", "wide_tiny_stage"):
        for a, b in zip(res["wide"][:4], res[mode][:4]):
            assert np.array_equal(a, b), mode
    kept = {m: res[m][4]["sum_hits_kept"] for m in res}
    print("seed hits kept:", kept)
    assert kept["wide"] < kept["narrow"] < kept["none"] and kept["wide"] == kept["wide_tiny_stage"]
    assert res["wide"][4]["n_mappings"] > 300
    idx.close(); reads.close(); ref.close()


def test_long_sketches_segmented_sort_equals_bitonic_network(ctx, monkeypatch):
    """K2 for reads with more than 16 384 minimizers: segmented device radix sort + finish kernel against the bitonic network
    (MM_SKETCH_BITONIC=1): identical sketches (hash, strand after the tie-break), sizes, ambiguity flags and records."""
    ref = ctx.synth_reference(seed=55, n_species=8, strains_per_species=3, genome_len=600_000, strain_divergence=0.02, genus_divergence=0.08)
    idx = ctx.index(ref, 16, 5)                                   # w = 5: a 60 kb read already has ~20 000 minimizers
    reads, _ = ctx.synth_reads(ref, seed=59, n_reads=120, read_len=150_000, read_len_min=20_000, sub_rate=0.03, ins_rate=0.02, del_rate=0.03, frac_random=0.05, n_abundant=6)
    monkeypatch.setenv("MM_EAGER_TIEBREAK", "1")                  # every duplicated-hash strand resolved, so that whole sketches compare
    res = {}
    for mode in ("segmented", "bitonic"):
        if mode == "bitonic": monkeypatch.setenv("MM_SKETCH_BITONIC", "1")
        M = ctx.map_batch(idx, reads, 16, 5)
        sk_off, sk_h, sk_s = M.debug_sketch()
        off, rec = M.fetch()
        res[mode] = (sk_off.copy(), sk_h.copy(), sk_s.copy(), off.copy(), rec.copy(), M.stats())
        M.close()
        monkeypatch.delenv("MM_SKETCH_BITONIC", raising=False)
    for a, b in zip(res["segmented"][:5], res["bitonic"][:5]):
        assert np.array_equal(a, b)
    assert np.diff(res["segmented"][0]).max() > 16384 and res["segmented"][5]["n_mappings"] > 100
    assert res["segmented"][5]["n_ambiguous_sketch_reads"] == res["bitonic"][5]["n_ambiguous_sketch_reads"]
    idx.close(); reads.close(); ref.close()


def test_hit_prefilter_keeps_candidates_identical(ctx, monkeypatch):
    """K3c drops seed hits that cannot belong to a qualifying run; candidates and mappings must not change."""
    ref = ctx.synth_reference(seed=6, n_species=40, strains_per_species=4, genome_len=300_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=10, n_reads=2000, read_len=6000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.1, n_abundant=30)
    idx = ctx.index(ref, 16, 8)
    idx.set_freq_threshold(2**31 - 1)                  # keep every hash: many chance hits
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MM_NO_HIT_FILTER", mode)
        M = ctx.map_batch(idx, reads, 16, 8)
        coff, cand = M.debug_candidates()
        off, rec = M.fetch()
        res[mode] = (coff.copy(), cand.copy(), off.copy(), rec.copy(), M.stats())
        M.close()
    monkeypatch.delenv("MM_NO_HIT_FILTER")
    for i in range(4):
        assert np.array_equal(res["0"][i], res["1"][i]), i
    assert res["0"][4]["sum_hits_kept"] < res["1"][4]["sum_hits_kept"] == res["1"][4]["sum_hits"]
    assert res["0"][4]["n_candidates"] > 2000
    idx.close(); reads.close(); ref.close()


def test_em_from_mapping_equals_host_built_problem(ctx):
    """mm_em_create_from_mapping (device) against the same EM problem assembled on the host from fetched records with the
    reference's rules (fEM.h:234-353): multi-contig taxa, contigs shorter than the read, 6-digit mapping qualities."""
    from metamaps_amd import emhost
    ref = ctx.synth_reference(seed=8, n_species=24, strains_per_species=4, genome_len=150_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=12, n_reads=1500, read_len=5000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.1, n_abundant=30)
    idx = ctx.index(ref, 16, 8)
    M = ctx.map_batch(idx, reads, 16, 8)
    M.add_qualities(16)
    off, rec = M.fetch()
    C = ref.count
    contig_taxon = (np.arange(C) // 2).astype(np.int32)          # two contigs per taxon
    contig_len = ref.lengths().astype(np.int32).copy()
    contig_len[::5] = 3000                                        # pretend every fifth contig is shorter than the reads
    T = int(contig_taxon.max()) + 1
    rl = reads.lengths().astype(np.int64)
    # host construction
    taxon = contig_taxon[rec["ref_contig"]]
    mapq = emhost.parse6(rec["mapq"].astype(np.float64))
    inv = np.zeros(len(rec))
    for r in range(len(off) - 1):
        a, b = int(off[r]), int(off[r + 1])
        L = int(rl[r])
        cs = set(int(c) for c in rec["ref_contig"][a:b])
        for i in range(a, b):
            t = int(taxon[i]); n = 0
            for c in np.nonzero(contig_taxon == t)[0]:
                if contig_len[c] >= L: n += int(contig_len[c]) - L + 1
                elif int(c) in cs: n += 1
            inv[i] = 1.0 / n
    e_host = ctx.em(off, taxon, mapq, inv, T)
    e_dev = ctx.em_from_mapping(M, contig_taxon, contig_len, T)
    assert e_dev.n_entries == len(rec) and e_dev.n_reads == len(off) - 1
    assert np.array_equal(e_dev.taxon_counts(), np.bincount(taxon, minlength=T))
    f = np.where(np.bincount(taxon, minlength=T) > 0, 1.0, 0.0); f /= f.sum()
    for _ in range(3):
        ph, lh = e_host.iterate(f)
        pd_, ld = e_dev.iterate(f)
        assert np.array_equal(ph, pd_) and lh == ld
        f = ph / ph.sum()
    p1, b1 = e_host.posteriors(f); p2, b2 = e_dev.posteriors(f)
    assert np.array_equal(b1, b2)
    # (mapping qualities around 1e-35 round differently in numpy's and the device's 6-digit rounding: posteriors that small only)
    assert np.allclose(p1, p2, rtol=1e-12, atol=1e-30)
    e_host.close(); e_dev.close(); M.close(); idx.close(); reads.close(); ref.close()


def test_hit_filter_staging_overflow_path(ctx, monkeypatch):
    """survivors beyond the per-read staging capacity are re-filtered by the write kernel: same hits either way"""
    ref = ctx.synth_reference(seed=6, n_species=40, strains_per_species=4, genome_len=300_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=10, n_reads=1000, read_len=6000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.1, n_abundant=30)
    idx = ctx.index(ref, 16, 8)
    res = {}
    for cap in (None, "16"):
        if cap:
            monkeypatch.setenv("MM_HF_STAGE_CAP", cap)
        M = ctx.map_batch(idx, reads, 16, 8)
        off, c, w = M.debug_hits()
        ro, rec = M.fetch()
        res[cap] = (off.copy(), np.sort(c.astype(np.int64) << 32 | w), ro.copy(), rec.copy())   # hits per read are sorted later: compare as sets per batch
        M.close()
    monkeypatch.delenv("MM_HF_STAGE_CAP")
    assert np.array_equal(res[None][0], res["16"][0]) and np.array_equal(res[None][1], res["16"][1])
    assert np.array_equal(res[None][2], res["16"][2]) and np.array_equal(res[None][3], res["16"][3])
    idx.close(); reads.close(); ref.close()


def test_l1_wave_scan_equals_serial_loop(ctx, monkeypatch):
    """K4b one wavefront per read against the literal one-thread-per-read loop (MM_L1_SERIAL=1)."""
    ref = ctx.synth_reference(seed=6, n_species=40, strains_per_species=4, genome_len=300_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=10, n_reads=2000, read_len=6000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.1, n_abundant=30)
    idx = ctx.index(ref, 16, 8)
    res = {}
    for mode in ("wave", "serial"):
        if mode == "serial":
            monkeypatch.setenv("MM_L1_SERIAL", "1")
        M = ctx.map_batch(idx, reads, 16, 8)
        off, cand = M.debug_candidates()
        res[mode] = (off.copy(), cand.copy())
        M.close()
    monkeypatch.delenv("MM_L1_SERIAL")
    assert np.array_equal(res["wave"][0], res["serial"][0]) and np.array_equal(res["wave"][1], res["serial"][1])
    assert len(res["wave"][1]) > 3000
    idx.close(); reads.close(); ref.close()


def test_tiebreak_modes_agree(ctx, monkeypatch):
    """Duplicated hashes with differing strands (computeMap.hpp:292-295): resolving all of them up front, only those
    an undecided strand vote needs (default), or every one a vote reads (forced) must give the same records."""
    ref = ctx.synth_reference(seed=5, n_species=48, strains_per_species=4, genome_len=400_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=9, n_reads=4000, read_len=12000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=40)
    idx = ctx.index(ref, 16, 8)
    res = {}
    for mode, env in (("default", None), ("eager", "MM_EAGER_TIEBREAK"), ("redo", "MM_FORCE_AMB_REDO")):
        if env:
            monkeypatch.setenv(env, "1")
        M = ctx.map_batch(idx, reads, 16, 8)
        off, rec = M.fetch()
        res[mode] = (off.copy(), rec.copy(), M.stats())
        M.close()
        if env:
            monkeypatch.delenv(env)
    assert res["default"][2]["n_ambiguous_sketch_reads"] > 10
    assert res["redo"][2]["n_l2_wide_redo"] > 0                  # the host-resolution path really ran
    for mode in ("eager", "redo"):
        assert np.array_equal(res["default"][0], res[mode][0]) and np.array_equal(res["default"][1], res[mode][1]), mode
    idx.close(); reads.close(); ref.close()


def test_tiebreak_modes_agree_on_long_reads(ctx, monkeypatch):
    """the same for reads of the long-sketch paths (segmented sketch sort with per-entry marks, dense K5 path with the
    unresolved-strand feedback and its redo): default (lazy), everything up front, every touched read resolved and redone"""
    ref = ctx.synth_reference(seed=65, n_species=10, strains_per_species=4, genome_len=500_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=69, n_reads=160, read_len=130_000, read_len_min=50_000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=8)
    idx = ctx.index(ref, 16, 8)
    res = {}
    for mode, env in (("default", None), ("eager", "MM_EAGER_TIEBREAK"), ("redo", "MM_FORCE_AMB_REDO")):
        if env:
            monkeypatch.setenv(env, "1")
        M = ctx.map_batch(idx, reads, 16, 8)
        off, rec = M.fetch()
        res[mode] = (off.copy(), rec.copy(), M.stats())
        M.close()
        if env:
            monkeypatch.delenv(env)
    assert res["default"][2]["n_ambiguous_sketch_reads"] > 10 and res["default"][2]["n_mappings"] > 200
    assert res["redo"][2]["n_l2_wide_redo"] > 0                  # the host-resolution path really ran
    for mode in ("eager", "redo"):
        assert np.array_equal(res["default"][0], res[mode][0]) and np.array_equal(res["default"][1], res[mode][1]), mode
    idx.close(); reads.close(); ref.close()


@pytest.mark.parametrize("read_len,n_reads", [(20_000, 300), (45_000, 150), (90_000, 60)])
def test_l2_long_read_classes_equal_full_slide(ctx, monkeypatch, read_len, n_reads):
    """the sketch-size classes of K5 beyond the 10 kb case (masks for 32 768 streamed entries, blocks of several words,
    two candidates per workgroup; from ~58 kb on the long-read path of mm_l2_dense.hpp) against the literal full slide"""
    ref = ctx.synth_reference(seed=15, n_species=12, strains_per_species=4, genome_len=600_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=19, n_reads=n_reads, read_len=read_len, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=10)
    idx = ctx.index(ref, 16, 8)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MM_L2_FULL", mode)
        M = ctx.map_batch(idx, reads, 16, 8)
        off, rec = M.fetch()
        res[mode] = (off.copy(), rec.copy(), M.stats())
        M.close()
    monkeypatch.delenv("MM_L2_FULL")
    assert np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1])
    assert res["0"][2]["n_mappings"] > n_reads
    if read_len < 58_000:                                         # the LDS classes skip most windows ...
        assert res["0"][2]["sum_l2_evals"] < res["1"][2]["sum_l2_evals"]
    else:                                                         # ... the long-read path (state in global memory) slides over every window up to the point
        assert res["0"][2]["sum_l2_evals"] <= res["1"][2]["sum_l2_evals"]   # where nothing later can reach the best so far
        monkeypatch.setenv("MM_L2_DENSE_NO_STOP", "1")            # without that early end: every window, like the full slide
        M = ctx.map_batch(idx, reads, 16, 8)
        off, rec = M.fetch()
        assert np.array_equal(res["0"][0], off) and np.array_equal(res["0"][1], rec)
        assert M.stats()["sum_l2_evals"] == res["1"][2]["sum_l2_evals"] and res["0"][2]["sum_l2_evals"] < 0.9 * res["1"][2]["sum_l2_evals"]
        M.close()
        monkeypatch.delenv("MM_L2_DENSE_NO_STOP")
    idx.close(); reads.close(); ref.close()


def test_mixed_read_lengths_one_batch_equal_full_slide(ctx, monkeypatch):
    """reads of 1-60 kb in one batch (BASELINE config 3 shape): every K5 class is launched in the same map_batch and shares
    the code-word scratch; records must equal the literal full slide"""
    ref = ctx.synth_reference(seed=25, n_species=12, strains_per_species=4, genome_len=600_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=29, n_reads=1200, read_len=60_000, read_len_min=1_000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=10)
    lens = reads.lengths()
    assert lens.min() < 2_500 and lens.max() > 40_000 and ((lens > 14_000) & (lens < 30_000)).sum() > 50
    idx = ctx.index(ref, 16, 8)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MM_L2_FULL", mode)
        M = ctx.map_batch(idx, reads, 16, 8)
        off, rec = M.fetch()
        res[mode] = (off.copy(), rec.copy(), M.stats())
        M.close()
    monkeypatch.delenv("MM_L2_FULL")
    assert np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1])
    assert res["0"][2]["n_mappings"] > 1500
    idx.close(); reads.close(); ref.close()


@pytest.fixture(scope="module")
def dense(tmp_path_factory, oracle_lib):
    """The bench's regime at oracle size: a 30 Mbp reference under k = 10 saturates the hash space (a few hundred thousand distinct
    minimizer hashes for ~7 million index entries, ~20 occurrences per hash), so that a read draws tens of thousands of chance
    seed hits of which only the few per cent around its true locations matter — what K3 / K3c / K4 see at miniSeq+H density."""
    from metamaps_amd import synth
    d = tmp_path_factory.mktemp("dense")
    db = synth.make_db(str(d / "db"), n_genomes=30, genome_len=1_000_000, seed=21, contigs_per_genome=2)
    rd = synth.make_reads(db, str(d / "reads.fq"), n_reads=220, read_len=5000, seed=8, len_jitter=0.3)
    return {"dir": str(d), "db": db, "reads": rd["path"]}


@pytest.mark.parametrize("k,w,force_thr", [(10, 8, None), (10, 8, "cut30"), (11, 6, None)])
def test_mapping_at_bench_hit_density_matches_oracle(ctx, oracle_lib, dense, k, w, force_thr, monkeypatch):
    """Every stage against the oracle where the seed-hit filter has real work: > 90 % of the raw hits dropped, long occurrence
    lists, and (force_thr) a frequency threshold that actually cuts lists (computeMap.hpp:307-385, winSketch.hpp:452-494)."""
    names, contigs = _read_fasta(dense["db"].fasta)
    rnames, reads = _read_fastq(dense["reads"])
    S = ctx.seqset(contigs); R = ctx.seqset(reads)
    idx = ctx.index(S, k, w)
    oi = oracle_lib.index(dense["db"].fasta, k, w)
    info = idx.info()
    assert info["n_entries"] == oi.n and info["n_unique_hashes"] == oi.n_unique and idx.freq_threshold == oi.freq_threshold
    assert info["n_entries"] > 5 * info["n_unique_hashes"]        # saturated: more than five occurrences per hash on average
    if force_thr is not None:                                      # the threshold that cuts the lists holding ~30 % of the index entries
        cnt, nh = idx.freq_hist()
        cut = np.cumsum((cnt * nh)[::-1])[::-1] / float((cnt * nh).sum())       # share of the entries in lists of >= cnt[i] occurrences
        force_thr = int(cnt[np.argmax(cut <= 0.3)])
        idx.set_freq_threshold(force_thr); oi.set_freq_threshold(force_thr)
    monkeypatch.setenv("MM_EAGER_TIEBREAK", "1")
    exp = [oi.map_read(q, 80.0) if len(q) >= max(1000, k, w) else None for q in reads]
    out = {}
    for mode in ("filter", "raw"):
        if mode == "raw":
            monkeypatch.setenv("MM_NO_HIT_FILTER", "1")
        M = ctx.map_batch(idx, R, k, w, pi=80.0, min_read_len=1000)
        M.add_qualities(k)
        out[mode] = dict(st=M.stats(), sk=M.debug_sketch(), hits=M.debug_hits(), cand=M.debug_candidates(), mh=M.debug_min_hits(), rec=M.fetch())
        out[mode]["l2"] = M.debug_l2(len(out[mode]["cand"][1]))
        M.close()
    monkeypatch.delenv("MM_NO_HIT_FILTER")
    st = out["filter"]["st"]
    print("density", k, w, force_thr, {kk: st[kk] for kk in ("sum_sketch", "sum_hits", "sum_hits_kept", "n_candidates", "n_mappings")})
    if force_thr is None:
        assert st["sum_hits"] > 2_000_000 and st["sum_hits_kept"] * 10 < st["sum_hits"], st  # the filter drops > 90 %
    else:                                                          # most lists are cut by the threshold, the filter still drops most of the rest
        assert 1_000_000 < st["sum_hits"] and st["sum_hits_kept"] * 5 < st["sum_hits"], st
    for mode in ("filter", "raw"):
        sk_off, sk_h, sk_s = out[mode]["sk"]; hit_off, hit_c, hit_w = out[mode]["hits"]; cand_off, cand = out[mode]["cand"]
        rec_off, rec = out[mode]["rec"]; l2 = out[mode]["l2"]; mh = out[mode]["mh"]
        n_mapped = 0
        for r, q in enumerate(reads):
            o = exp[r]
            if o is None:
                assert rec_off[r + 1] == rec_off[r]
                continue
            a, b = int(sk_off[r]), int(sk_off[r + 1])
            assert np.array_equal(sk_h[a:b], o["sketch_hash"]) and np.array_equal(sk_s[a:b], o["sketch_strand"]), r
            assert mh[r] == o["min_hits"], r
            a, b = int(hit_off[r]), int(hit_off[r + 1])
            if mode == "raw":                                     # the complete seed-hit list, also under the forced threshold
                assert np.array_equal(hit_c[a:b], o["hit_contig"]) and np.array_equal(hit_w[a:b], o["hit_wpos"]), r
            else:                                                 # what the filter keeps is a sub-list of it, in order
                key_all = o["hit_contig"].astype(np.int64) << 32 | o["hit_wpos"]
                key_kept = hit_c[a:b].astype(np.int64) << 32 | hit_w[a:b]
                assert np.all(np.diff(key_kept) >= 0) and np.all(np.isin(key_kept, key_all)), r
            a, b = int(cand_off[r]), int(cand_off[r + 1])
            assert np.array_equal(cand[a:b], o["cand"]), r
            got, ex = l2[a:b], o["l2"]
            ok = got[:, 5] == 1
            assert np.array_equal(got[:, 0], ex[:, 0]) and np.array_equal(got[ok][:, [1, 2, 3, 4]], ex[ok][:, [1, 2, 3, 4]]), r
            assert np.all(got[~ok][:, 2] <= ex[~ok][:, 2]) and int(ok.sum()) == len(o["map"]), r
            a, b = int(rec_off[r]), int(rec_off[r + 1])
            m, rr = o["map"], rec[a:b]
            assert b - a == len(m), r
            assert np.array_equal(rr["ref_contig"], m[:, 0]) and np.array_equal(rr["ref_start"], m[:, 1]) and np.array_equal(rr["shared"], m[:, 3]), r
            assert np.array_equal(rr["sketch"], m[:, 4]) and np.array_equal(rr["strand"], m[:, 5]), r
            n_mapped += len(m) > 0
        assert n_mapped > 150, n_mapped
    oi.close(); idx.close(); R.close(); S.close()


def test_community_generator_and_parity_on_it(ctx, oracle_lib, tmp_path):
    """The SURVEY D1 community at toy size (lognormal genome lengths, 1-12 strains per species with block indels, human-like
    contigs with 45 % library repeats and N runs, shuffled contig order): structure checks, then the mapper against the oracle on
    the very sequences the device generated — repeats (long occurrence lists, duplicate hashes inside windows) and N runs included."""
    ref, genome = ctx.synth_community(seed=5, n_genomes=40, n_species=12, n_genera=4, median_len=120_000.0, sigma_len=0.6, min_len=5_000, max_len=600_000,
                                      strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=6,
                                      human_contigs=3, human_bases=1_500_000, repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=40, total_bases_target=0)
    lens = ref.lengths()
    assert len(lens) == 43 and len(genome) == 43 and int((genome == 40).sum()) == 3 and set(genome[genome < 40]) == set(range(40))
    assert lens.min() >= 4_000 and len(set(lens.tolist())) > 20                     # lognormal lengths (+- block indels), not a constant
    seqs = [ref.fetch(i, int(lens[i])) for i in range(len(lens))]
    human = [s for s, g in zip(seqs, genome) if g == 40]
    n_frac = sum(s.count(b"N") for s in human) / sum(len(s) for s in human)
    assert 0.008 < n_frac < 0.012 and all(s.startswith(b"N") and s.endswith(b"N") for s in human)
    assert all(set(s) <= set(b"ACGT") for s, g in zip(seqs, genome) if g < 40)
    fa = str(tmp_path / "DB.fa")
    with open(fa, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(f">C{i}|kraken:taxid|{int(genome[i]) + 1}|x\n".encode() + s + b"\n")
    k, w = 16, 8
    idx = ctx.index(ref, k, w)
    oi = oracle_lib.index(fa, k, w)
    h, c, wp, st = idx.entries(); oh, oc, ow, os_ = oi.dump()
    assert np.array_equal(h, oh) and np.array_equal(c, oc) and np.array_equal(wp, ow) and np.array_equal(st, os_)
    cnt, nh = idx.freq_hist()
    assert cnt.max() > 30                                                             # repeat granules: hashes with dozens of occurrences
    reads, truth = ctx.synth_reads(ref, seed=3, n_reads=300, read_len=4000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=43)
    assert int((genome[truth[truth >= 0]] == 40).sum()) > 5                            # some reads come from the human-like contigs
    M = ctx.map_batch(idx, reads, k, w); M.add_qualities(k)
    off, rec = M.fetch()
    rl = reads.lengths()
    n_map = 0
    for r in range(300):
        q = reads.fetch(r, int(rl[r]))
        m = oi.map_read(q)["map"] if len(q) >= 1000 else np.zeros((0, 6), dtype=np.int32)
        got = rec[off[r]:off[r + 1]]
        assert len(got) == len(m), (r, len(got), len(m))
        assert np.array_equal(got["ref_contig"], m[:, 0]) and np.array_equal(got["ref_start"], m[:, 1]) and np.array_equal(got["shared"], m[:, 3]), r
        assert np.array_equal(got["strand"], m[:, 5]), r
        n_map += len(m) > 0
    assert n_map > 250
    oi.close(); M.close(); idx.close(); reads.close(); ref.close()


def test_l2_dense_path_equals_lds_classes(ctx, monkeypatch):
    """The long-read K5 path (window state in global memory, every window evaluated in the reference's order: mm_l2_dense.hpp) forced
    onto reads of every length (MM_L2_DENSE_FROM=1) against the LDS classes with their exact skip-ahead — which the tests above pin to
    the oracle — on a workload with thousands of candidates, zone shifts and duplicated hashes inside windows."""
    ref, genome = ctx.synth_community(seed=11, n_genomes=60, n_species=15, n_genera=5, median_len=300_000.0, sigma_len=0.5, min_len=20_000, max_len=900_000,
                                      strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=6,
                                      human_contigs=2, human_bases=3_000_000, repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=30, total_bases_target=0)
    reads, truth = ctx.synth_reads(ref, seed=9, n_reads=2500, read_len=12000, read_len_min=1500, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=62)
    idx = ctx.index(ref, 16, 8)
    res = {}
    for mode in ("lds", "dense"):
        if mode == "dense":
            monkeypatch.setenv("MM_L2_DENSE_FROM", "1")
        M = ctx.map_batch(idx, reads, 16, 8)
        off, rec = M.fetch()
        res[mode] = (off.copy(), rec.copy(), M.stats())
        M.close()
    monkeypatch.delenv("MM_L2_DENSE_FROM")
    assert np.array_equal(res["lds"][0], res["dense"][0]) and np.array_equal(res["lds"][1], res["dense"][1])
    assert res["lds"][2]["n_mappings"] > 4000
    assert res["dense"][2]["sum_l2_evals"] > 3 * res["lds"][2]["sum_l2_evals"]      # every window against the skip-ahead's few
    idx.close(); reads.close(); ref.close()


def test_adversarial_minimizers_match_committed_golden(ctx):
    """K1 on the adversarial winnowing set against tests/golden/core_golden.json (oracle output pinned in the repository: N runs,
    lower case, palindromes, tandem repeats, len == k, len == k + w - 1, w from 1 to 100, k from 5 to 32)"""
    import json, sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_core_golden import a2_digest
    from adversarial import adversarial_cases
    g = json.load(open(os.path.join(here, "golden", "core_golden.json")))["a2"]
    by_kw = {}
    for name, seq, k, w in adversarial_cases():
        by_kw.setdefault((k, w), []).append((name, seq))
    n = 0
    for (k, w), lst in sorted(by_kw.items()):
        S = ctx.seqset([s for _, s in lst])
        off, h, wp, st = ctx.minimizers(S, k, w)
        for i, (name, _) in enumerate(lst):
            a, b = int(off[i]), int(off[i + 1])
            assert [b - a, a2_digest(h[a:b], wp[a:b], st[a:b])] == g[name], (name, k, w)
            n += 1
        S.close()
    assert n == len(g)


def test_host_statistics_match_committed_golden():
    """mm_min_hits_relaxed and the accept threshold behind mm_identity (host side of the library) against the committed tables"""
    import json
    from metamaps_amd import capi
    import ctypes as C
    here = os.path.dirname(os.path.abspath(__file__))
    g = json.load(open(os.path.join(here, "golden", "core_golden.json")))
    L = capi.lib()
    steps = np.array(g["min_hits"]["steps_up_at"])
    for s in list(range(1, 300)) + list(range(300, 12001, 211)) + [12000]:
        assert L.mm_min_hits_relaxed(s, 16, 80.0) == int((steps <= s).sum()), s
    a, b = C.c_float(), C.c_float()
    for s, v in list(zip(g["accept_min"]["s"], g["accept_min"]["min_shared"]))[::5]:
        L.mm_identity(v, s, 16, C.byref(a), C.byref(b))
        assert b.value >= 80.0
        if v > 0:
            L.mm_identity(v - 1, s, 16, C.byref(a), C.byref(b))
            assert b.value < 80.0, (s, v)


def test_streaming_seed_filter_equals_one_read_per_workgroup(ctx, dense, monkeypatch):
    """K3: the resident-workgroup form (default: reads handed out by a ticket, the next read's look-ups in flight under this read's
    LDS phases) against the one-workgroup-per-read form (MM_SF_ONESHOT=1) and against the two-pass kernels (MM_NO_FUSED_FILTER=1), at
    bench hit density: the filtered seed hits, hit for hit, the raw hit counts, candidates and records; with a batch smaller than
    the grid, with reads of other classes in between (skipped by this kernel) and with a stage too small for some reads (fallback)"""
    names, contigs = _read_fasta(dense["db"].fasta)
    rnames, reads = _read_fastq(dense["reads"])
    rng = np.random.default_rng(3)
    long_read = contigs[0][:40_000]                               # beyond SF_SMAX: another class, the streaming kernel skips it
    mixed = []
    for i, q in enumerate(reads):
        mixed.append(q)
        if i % 37 == 5:
            mixed.append(long_read)
        if i % 53 == 7:
            mixed.append(b"ACGT" * 10)                            # too short: no sketch
    S = ctx.seqset(contigs)
    idx = ctx.index(S, 10, 8)
    for batch, cap in ((mixed, None), (reads[:7], None), (mixed, "40")):
        R = ctx.seqset(batch)
        got = {}
        for mode, env in (("stream", {}), ("oneshot", {"MM_SF_ONESHOT": "1"}), ("twopass", {"MM_NO_FUSED_FILTER": "1"})):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            if cap:
                monkeypatch.setenv("MM_HF_STAGE_CAP", cap)
            M = ctx.map_batch(idx, R, 10, 8, pi=80.0, min_read_len=1000)
            got[mode] = dict(st=M.stats(), hits=M.debug_hits(), cand=M.debug_candidates(), rec=M.fetch())
            M.close()
            for k_ in list(env) + ["MM_HF_STAGE_CAP"]:
                monkeypatch.delenv(k_, raising=False)
        a = got["stream"]
        assert a["st"]["sum_hits"] > 100_000 and a["st"]["sum_hits_kept"] > 0
        for other in ("oneshot", "twopass"):
            b = got[other]
            assert a["st"]["sum_hits"] == b["st"]["sum_hits"] and a["st"]["sum_hits_kept"] == b["st"]["sum_hits_kept"], other
            for x, y in zip(a["hits"], b["hits"]):
                assert np.array_equal(x, y), other
            assert np.array_equal(a["cand"][0], b["cand"][0]) and np.array_equal(a["cand"][1], b["cand"][1]), other
            assert np.array_equal(a["rec"][0], b["rec"][0]) and a["rec"][1].tobytes() == b["rec"][1].tobytes(), other
        R.close()
    idx.close(); S.close()


def test_map_batch_reusing_sketches_equals_map_batch(ctx, oracle_lib, mini, monkeypatch):
    """mm_map_batch_reusing (minimizers + sketches taken from an earlier mapping of the same reads — the second and later index chunks
    of --maxmemory) gives the records, candidates and sketches of mm_map_batch, in all three strand tie-break modes; a donor of other
    reads or other parameters is refused"""
    from metamaps_amd import capi
    names, contigs = _read_fasta(mini["db"].fasta)
    rnames, reads = _read_fastq(mini["reads"])
    k, w = 16, 8
    half = len(contigs) // 2
    A, B = ctx.seqset(contigs[:half]), ctx.seqset(contigs[half:])
    ia, ib = ctx.index(A, k, w), ctx.index(B, k, w)
    R = ctx.seqset(reads)
    for env in ({}, {"MM_EAGER_TIEBREAK": "1"}, {"MM_FORCE_AMB_REDO": "1"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        Ma = ctx.map_batch(ia, R, k, w)
        Mb = ctx.map_batch(ib, R, k, w)
        Mr = ctx.map_batch(ib, R, k, w, sketch_of=Ma)
        (ob, rb), (orr, rr) = Mb.fetch(), Mr.fetch()
        assert np.array_equal(ob, orr) and rb.tobytes() == rr.tobytes() and len(rb) > 100
        for x, y in zip(Mb.debug_sketch(), Mr.debug_sketch()):
            assert np.array_equal(x, y)
        for x, y in zip(Mb.debug_candidates(), Mr.debug_candidates()):
            assert np.array_equal(x, y)
        assert Mr.stats()["ms_sketch"] == 0.0 and Mb.stats()["ms_sketch"] > 0.0
        Mr2 = ctx.map_batch(ia, R, k, w, sketch_of=Mr)            # a mapping made from copied sketches is a donor like any other
        assert Mr2.fetch()[1].tobytes() == Ma.fetch()[1].tobytes()
        Sk = ctx.sketch_batch(R, k, w)                            # mm_sketch_batch: K1 + K2 alone, tied to no index; holds no records
        assert len(Sk.fetch()[1]) == 0 and Sk.stats()["sum_sketch"] == Ma.stats()["sum_sketch"] and Sk.stats()["n_candidates"] == 0
        for ix, Mfull in ((ia, Ma), (ib, Mb)):
            Ms = ctx.map_batch(ix, R, k, w, sketch_of=Sk)
            assert Ms.fetch()[1].tobytes() == Mfull.fetch()[1].tobytes() and Ms.stats()["ms_sketch"] == 0.0
            for x, y in zip(Mfull.debug_sketch(), Ms.debug_sketch()):
                assert np.array_equal(x, y)
            Ms.close()
        Sk.close()
        other = capi.Context(0)                                   # a donor of another context of the device: everything is copied, nothing held jointly
        Mo = other.map_batch(ib, R, k, w, sketch_of=Ma)
        assert Mo.fetch()[1].tobytes() == rb.tobytes()
        Mo.close(); other.close()
        for m_ in (Mb, Mr, Mr2):
            m_.close()
        for k_ in env:
            monkeypatch.delenv(k_)
        R2 = ctx.seqset(reads[:-1])
        with pytest.raises(capi.MMError):
            ctx.map_batch(ib, R2, k, w, sketch_of=Ma)
        with pytest.raises(capi.MMError):
            ctx.map_batch(ib, R, k, w, min_read_len=500, sketch_of=Ma)
        R2.close(); Ma.close()
    for x in (ia, ib, A, B, R):
        x.close()


def test_duplicate_neighbour_distances_and_scan_fallback(ctx, monkeypatch):
    """"Is another occurrence of this hash inside the window?" (slidingMap.hpp:139-214) is answered from the index's same-hash neighbour
    distances (mm_index.hpp: dup_bits / dup_rank / dup_dist) and by a scan of the window only where a stored distance is saturated and
    the window reaches that far.  MM_DUP_SAT lowers the saturation value at index build, so that a repeat-rich reference (45 % library
    repeats, tandem copies inside contigs) sends most questions through the saturated branches: 1 = every distance saturated (all scans,
    the behaviour before the distances existed), 7 / 300 = a mix.  Every K5 form — LDS classes with skip-ahead, the literal full slide,
    the long-read path — must give the records of the default index (which test_community_generator_and_parity_on_it pins to the oracle)."""
    ref, genome = ctx.synth_community(seed=21, n_genomes=40, n_species=10, n_genera=4, median_len=250_000.0, sigma_len=0.5, min_len=20_000, max_len=700_000,
                                      strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=6,
                                      human_contigs=3, human_bases=6_000_000, repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=12, total_bases_target=0)
    reads, truth = ctx.synth_reads(ref, seed=23, n_reads=1500, read_len=40_000, read_len_min=1_500, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.03, n_abundant=43)
    res = {}
    for sat in ("default", "1", "7", "300"):
        if sat != "default": monkeypatch.setenv("MM_DUP_SAT", sat)
        idx = ctx.index(ref, 16, 8)
        monkeypatch.delenv("MM_DUP_SAT", raising=False)
        if sat == "default": assert idx.info()["n_dup_flagged"] > 10_000   # pairs of same-hash entries inside one contig
        for mode, env in (("lds", {}), ("full", {"MM_L2_FULL": "1"}), ("dense", {"MM_L2_DENSE_FROM": "1"})):
            if sat in ("7", "300") and mode == "full": continue
            for k_, v_ in env.items(): monkeypatch.setenv(k_, v_)
            M = ctx.map_batch(idx, reads, 16, 8)
            off, rec = M.fetch()
            res[(sat, mode)] = (off.copy(), rec.copy(), M.stats()["n_mappings"])
            M.close()
            for k_ in env: monkeypatch.delenv(k_)
        idx.close()
    base = res[("default", "lds")]
    assert base[2] > 1500
    for key, got in res.items():
        assert np.array_equal(base[0], got[0]) and np.array_equal(base[1], got[1]), key
    reads.close(); ref.close()


def test_duplicate_neighbour_distances_match_definition(ctx):
    """the index's same-hash neighbour distances (mm_index.hpp) against their definition, entry by entry: within a contig, the
    distance in entries to the nearest earlier / later entry with the same hash; flags PW_DP / PW_DN exactly where one exists"""
    ref, genome = ctx.synth_community(seed=31, n_genomes=12, n_species=4, n_genera=2, median_len=120_000.0, sigma_len=0.5, min_len=20_000, max_len=300_000,
                                      strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=6,
                                      human_contigs=2, human_bases=1_500_000, repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=8, total_bases_target=0)
    idx = ctx.index(ref, 16, 8)
    hsh, ct, wp, st = idx.entries()
    pd, nd = idx.dup_neighbours()
    n = len(hsh)
    order = np.lexsort((np.arange(n), hsh, ct))                   # by contig, hash, entry number
    same = (ct[order][1:] == ct[order][:-1]) & (hsh[order][1:] == hsh[order][:-1])
    exp_p = np.zeros(n, dtype=np.int64); exp_n = np.zeros(n, dtype=np.int64)
    d = (order[1:] - order[:-1])[same]
    exp_p[order[1:][same]] = d
    exp_n[order[:-1][same]] = d
    assert same.sum() == idx.info()["n_dup_flagged"] and same.sum() > 5_000
    assert d.max() > 65535                                        # some neighbours lie beyond the stored range ...
    assert np.array_equal(pd, np.minimum(exp_p, 65535)) and np.array_equal(nd, np.minimum(exp_n, 65535))   # ... and are stored saturated
    idx.close(); ref.close()


def test_l2_scratch_slots_hand_over(ctx, monkeypatch):
    """K5's code words and class masks live in scratch slots that the waves of a launch take and give back (mm_l2.hpp: one slot per
    resident wave instead of one per wave of the launch).  Reads of 1-60 kb (every class of the skip kernels in one batch, far more
    waves than slots in the 10 kb class): the default, a pool of 48 slots that every launch's waves queue for (MM_L2_SLOTS), and one
    slot per wave of the launch without any hand-over (MM_L2_NO_SLOTS, the layout before) must give identical records."""
    ref = ctx.synth_reference(seed=35, n_species=12, strains_per_species=4, genome_len=600_000, strain_divergence=0.02, genus_divergence=0.08)
    reads, _ = ctx.synth_reads(ref, seed=39, n_reads=6000, read_len=60_000, read_len_min=1_000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=10)
    idx = ctx.index(ref, 16, 8)
    res = {}
    for mode, env in (("default", {}), ("few", {"MM_L2_SLOTS": "48"}), ("per_wave", {"MM_L2_NO_SLOTS": "1"})):
        for k_, v_ in env.items(): monkeypatch.setenv(k_, v_)
        for rep in range(2):                                      # twice: the flags of the first launch must all have come back
            M = ctx.map_batch(idx, reads, 16, 8)
            off, rec = M.fetch()
            res[(mode, rep)] = (off.copy(), rec.copy(), M.stats())
            M.close()
        for k_ in env: monkeypatch.delenv(k_)
    base = res[("default", 0)]
    assert base[2]["n_candidates"] > 16_000 and base[2]["n_l2_rebuilds"] > 0
    for key, got in res.items():
        assert np.array_equal(base[0], got[0]) and np.array_equal(base[1], got[1]), key
    idx.close(); reads.close(); ref.close()


def test_zone_kernel_equals_rank_code_kernel(ctx, monkeypatch):
    """K5 runs as the zone kernel (mm_l2z.hpp: membership through a bit table + compaction, window states from threshold masks of a band of
    128 ranks, the band predicted from L1's seed-hit count) by default; l2_kernel (a rank code per streamed entry, MM_L2_V1=1) is the same
    algorithm in its first form.  Reads of 1-60 kb on a repeat-rich reference (duplicate hashes inside windows, DP/DN flags): the default, the zone
    kernel without the predicted band (MM_L2_NO_FUSE: the band's masks from a second pass), the zone kernel for the 10 kb class only
    (MM_L2_V1_LONG), the zone kernel searching its index ranges itself (MM_L2_NO_RANGES; by default l2_ranges_kernel makes them for all candidates at once), all with a lowered saturation of the duplicate distances (MM_DUP_SAT: the scan fall-back), and l2_kernel must give
    identical records; the L2 tuples of every candidate are compared as well."""
    res = {}
    for sat in ("default", "40"):
        if sat != "default": monkeypatch.setenv("MM_DUP_SAT", sat)
        ref, _genome = ctx.synth_community(seed=23, n_genomes=40, n_species=10, n_genera=4, median_len=250_000.0, sigma_len=0.5, min_len=20_000, max_len=700_000,
                                           strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=6,
                                           human_contigs=3, human_bases=6_000_000, repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=12, total_bases_target=0)
        reads, _ = ctx.synth_reads(ref, seed=41, n_reads=4000, read_len=60_000, read_len_min=1_000, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=43)
        idx = ctx.index(ref, 16, 8)
        monkeypatch.delenv("MM_DUP_SAT", raising=False)
        for mode, env in (("zone", {}), ("zone_two_pass", {"MM_L2_NO_FUSE": "1"}), ("zone_short", {"MM_L2_V1_LONG": "1"}), ("zone_short_two_pass", {"MM_L2_V1_LONG": "1", "MM_L2_NO_FUSE": "1"}),
                          ("zone_own_ranges", {"MM_L2_NO_RANGES": "1"}), ("rank_codes", {"MM_L2_V1": "1"})):
            for k_, v_ in env.items(): monkeypatch.setenv(k_, v_)
            M = ctx.map_batch(idx, reads, 16, 8)
            off, rec = M.fetch()
            st = M.stats()
            res[(sat, mode)] = (off.copy(), rec.copy(), M.debug_l2(st["n_candidates"]), st)
            M.close()
            for k_ in env: monkeypatch.delenv(k_)
        base = res[(sat, "rank_codes")]
        assert base[3]["n_candidates"] > 10_000 and base[3]["n_mappings"] > 5_000
        for mode in ("zone", "zone_two_pass", "zone_short", "zone_short_two_pass", "zone_own_ranges"):
            got = res[(sat, mode)]
            assert np.array_equal(base[0], got[0]) and np.array_equal(base[1], got[1]), (sat, mode)
            acc = base[2][:, 5] == 1
            assert np.array_equal(base[2][acc], got[2][acc]), (sat, mode)
        idx.close(); reads.close(); ref.close()

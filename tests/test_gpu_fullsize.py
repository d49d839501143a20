"""Parity at BASELINE.json's full single-GPU size, on the workload `value` is quoted on: the SURVEY D1 community reference
(12 000 microbial genomes in 3 000 species of 1-12 strains + 24 human-like contigs with 45 % library repeats and N runs,
26.76 Gbp, k = 16, w = 8; bench.py builds the same one) — configs[1] —, on configs[2] as one GPU can stand in for eight (the batch as
eight read shards mapped in turn, the EM's per-rank partial sums added on the host), and on configs[3] at the size of its per-GPU
share: 125 000 mixed 1-50 kb PacBio-error reads (10^6 / 8 GPUs) against the same reference split by the --maxmemory chunk rule into
resident chunk indexes, and into more chunks that are built, mapped and dropped in turn (multi-pass streaming; configs[4] at ITS size,
the 300 Gbp reference, is tests/test_gpu_refseq_scale.py).  No oracle run is feasible at 26.76 Gbp, so what is checked is what must hold at any size:

  configs[1]  truth recovery at species level; determinism; shard invariance (the multi-GPU partition of the reads);
              the seed-hit pre-filter is exact (MM_NO_HIT_FILTER=1), the K5 sweep equals the full slide (MM_L2_FULL=1), eager = lazy
              strand tie-break; record invariants; EM: frequencies sum to 1, log-likelihood never decreases, posteriors sum to 1,
              abundant genomes come out on top
  configs[2]  the 100 000 reads as eight contiguous shards (dist.shard_range): records of the shards one behind the other ==
              records of the whole batch, byte for byte; the EM run as eight ranks would run it — every shard its own EM problem, one E+M
              step each per iteration, the eight partial (T + 1)-vectors added on the host (what ncclAllReduce does between physical
              GPUs, fEM.h:583-600) — follows the single-rank trajectory to 1e-12 and ends at the same frequencies; a shard's one-rank
              RCCL iteration (MM_EM_FORCE_COLLECTIVE=1: P1-P3' | ncclAllReduce | finalize) == its plain partial sums, normalised
  configs[3]  the chunk rule on the whole index gives >= 3 chunks; mapping against the chunk indexes and merging read-wise in chunk
              order (unifyFiles, mapWrap.h:128-145) == mapping against the whole index when no hash is cut by freqThreshold (a hash
              spread over several chunks meets a different count in each, so only then is equality exact); with the reference's
              per-chunk thresholds from the accumulated histogram (winSketch.hpp:452-494): records in chunk order, mapping
              qualities over the union sum to 1, truth recovery as unchunked
  configs[4]  the same reads through >= 8 chunks, one index on the device at a time (built, mapped, dropped), records gathered on the
              host and merged (mm_mapping_from_parts): identical to the resident-chunk result

Stage 1 holds the index of the whole reference (149 GB); stage 2 drops it and builds chunk indexes from device-side slices of the
packed reference (mm_seqset_slice).  The tests run in file order."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, W = 16, 8
COMM = dict(seed=20260928, n_genomes=12000, n_species=3000, n_genera=600, median_len=2.0e6, sigma_len=0.6, min_len=5_000, max_len=12_000_000,
            strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
            human_contigs=24, human_bases=int(3.1e9), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000,
            total_bases_target=26_762_276_280)
N_READS, RLEN = 100_000, 10_000                               # configs[1]: the bench batch at its size
N_MIXED, MIXED_MIN, MIXED_MAX = 125_000, 1_000, 50_000        # configs[3]: the per-GPU share of 10^6 reads over 8 GPUs, mapped as TWO calls of 62 500 reads (0.8 Gbp each, what bench.py --config 3 maps per step beside the four resident chunk indexes: 227 of 288 GiB; the CLI maps such reads in batches of <= 0.256 Gbp)
N_HALVES = 2
INT_MAX = 2**31 - 1
GIB = 1 << 30


def _map(ctx, idx, reads, env=None, qualities=True):
    old = {}
    for k_, v in (env or {}).items():
        old[k_] = os.environ.get(k_); os.environ[k_] = v
    try:
        M = ctx.map_batch(idx, reads, K, W)
        if qualities:
            M.add_qualities(K)
        off, rec = M.fetch()
        rec = rec.copy(); st = M.stats()
        M.close()
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    return off, rec, st


def _halves(world):
    """the mixed batch as N_HALVES device-side slices: [(first read, SeqSet)]"""
    return world["mixed_halves"]


def _join(parts):
    """[(first read, off, rec)] of consecutive read ranges -> (off, rec) of the whole batch"""
    offs, recs, base = [np.zeros(1, dtype=parts[0][1].dtype)], [], 0
    for first, off, rec in parts:
        rec = rec.copy(); rec["read"] += first
        offs.append(off[1:] + base); recs.append(rec); base += len(rec)
    return np.concatenate(offs), np.concatenate(recs)


def _subset(ctx, reads, which):
    rl = reads.lengths()
    return ctx.seqset([reads.fetch(int(i), int(rl[i])) for i in which])


@pytest.fixture(scope="module")
def world():
    from metamaps_amd import capi
    ctx = capi.Context(0)
    ref, genome = ctx.synth_community(**COMM)
    species = capi.Context.synth_community_species(**COMM)
    contig_species = np.where(genome < COMM["n_genomes"], species[np.minimum(genome, COMM["n_genomes"] - 1)], -2)   # -2: human-like
    idx = ctx.index(ref, K, W)
    reads, truth = ctx.synth_reads(ref, seed=4242, n_reads=N_READS, read_len=RLEN, sub_rate=0.04, ins_rate=0.03, del_rate=0.05,
                                   frac_random=0.05, n_abundant=100)
    off, rec, stats = _map(ctx, idx, reads)
    mixed, mtruth = ctx.synth_reads(ref, seed=777, n_reads=N_MIXED, read_len=MIXED_MAX, read_len_min=MIXED_MIN, sub_rate=0.02, ins_rate=0.08, del_rate=0.02,
                                    frac_random=0.05, n_abundant=100)
    bounds = [N_MIXED * i // N_HALVES for i in range(N_HALVES + 1)]
    halves = [(a, mixed.slice(a, b - a)) for a, b in zip(bounds, bounds[1:])]
    w = dict(ctx=ctx, ref=ref, genome=genome, species=species, contig_species=contig_species, idx=idx, reads=reads, truth=truth, off=off, rec=rec, stats=stats,
             mixed=mixed, mixed_halves=halves, mtruth=mtruth, stage=1, chunk_idx=[])
    yield w
    for _a, h in halves:
        h.close()
    for ix in w["chunk_idx"]:
        ix.close()
    if w["idx"] is not None:
        w["idx"].close()
    mixed.close(); reads.close(); ref.close(); ctx.close()


# ---------------------------------------------------------------------------------------------- configs[1], the bench workload
def test_index_is_the_bench_index(world):
    info = world["idx"].info()
    assert info["n_contigs"] == COMM["n_genomes"] + COMM["human_contigs"]
    assert abs(world["ref"].total_bases - 26_762_276_280) < 50_000_000
    assert info["n_entries"] > 5_500_000_000 and info["n_unique_hashes"] > 500_000_000
    assert info["hbm_bytes"] < 200 * GIB                          # resident, replicated per GPU
    assert 1000 < world["idx"].freq_threshold < 3000              # the library repeats push it up (101 on the uniform shape)
    st = world["stats"]
    assert st["n_reads_long_enough"] == N_READS and st["sum_hits_kept"] < st["sum_hits"] // 10


def _species_recovery(world, off, rec, truth, random_may_map=0.0):
    n_map = np.diff(off)
    from_genome = truth >= 0
    # random sequence never maps at 10 kb; a 1-2 kb read of random sequence (sketch of ~300 hashes, minimumHits 3-4) now and then
    # finds a chance candidate that passes the identity bound at this hash density — the reference's rule, not an artefact
    assert (n_map[~from_genome] > 0).mean() <= random_may_map, (n_map[~from_genome] > 0).mean()
    mapped = from_genome & (n_map > 0)
    good = 0
    cs = world["contig_species"]
    for r in np.nonzero(mapped)[0]:
        seg = rec[off[r]:off[r + 1]]
        best = seg[np.argmax(seg["mapq"])]
        good += cs[int(best["ref_contig"])] == cs[int(truth[r])]
    return mapped.sum() / max(1, from_genome.sum()), good / max(1, int(mapped.sum()))


def test_truth_recovery_at_species_level(world):
    frac_mapped, frac_right = _species_recovery(world, world["off"], world["rec"], world["truth"])
    assert frac_mapped > 0.99 and frac_right > 0.99, (frac_mapped, frac_right)


def test_record_invariants(world):
    off, rec = world["off"], world["rec"]
    assert (rec["shared"] <= rec["sketch"]).all() and (rec["shared"] > 0).all()
    assert np.isin(rec["strand"], (-1, 1)).all()
    assert (np.diff(rec["read"]) >= 0).all()
    sums = np.add.reduceat(rec["mapq"], off[:-1][np.diff(off) > 0])
    assert np.allclose(sums, 1.0, atol=1e-9)
    key = rec["read"].astype(np.int64) << 44 | rec["ref_contig"].astype(np.int64) << 30 | rec["ref_start"].astype(np.int64)
    assert len(np.unique(key)) == len(key) and (np.diff(key) > 0).all()   # (read, contig, position) order, no duplicate


def test_idempotent_and_shard_invariant(world):
    ctx, idx = world["ctx"], world["idx"]
    off2, rec2, _ = _map(ctx, idx, world["reads"])
    assert (off2 == world["off"]).all() and rec2.tobytes() == world["rec"].tobytes()
    n = 6000
    a, b = _subset(ctx, world["reads"], range(0, n // 2)), _subset(ctx, world["reads"], range(n // 2, n))
    (oa, ra, _), (ob, rb, _) = _map(ctx, idx, a), _map(ctx, idx, b)
    a.close(); b.close()
    rb = rb.copy(); rb["read"] += n // 2
    whole = world["rec"][:world["off"][n]]
    assert len(ra) + len(rb) == len(whole)
    assert np.concatenate([ra, rb]).tobytes() == whole.tobytes()


@pytest.mark.parametrize("env", [{"MM_NO_HIT_FILTER": "1"}, {"MM_L2_FULL": "1"}, {"MM_EAGER_TIEBREAK": "1"}], ids=lambda e: next(iter(e)))
def test_kernel_variants_agree_at_full_density(world, env):
    n = 1500
    sub = _subset(world["ctx"], world["reads"], range(n))
    off, rec, _ = _map(world["ctx"], world["idx"], sub, env)
    sub.close()
    assert (off == world["off"][:n + 1]).all()
    assert rec.tobytes() == world["rec"][:world["off"][n]].tobytes()


def test_em_properties(world):
    ctx = world["ctx"]
    n_taxa = COMM["n_genomes"] + 1                                # one taxon per microbial genome + the human-like one
    contig_len = world["ref"].lengths().astype(np.int64)
    M = ctx.map_batch(world["idx"], world["reads"], K, W)
    M.add_qualities(K)
    em = ctx.em_from_mapping(M, world["genome"], contig_len, n_taxa)
    counts = em.taxon_counts()
    f = (counts > 0).astype(np.float64); f /= f.sum()
    lls = []
    for _ in range(6):
        f, ll = em.iterate(f)
        f = f / f.sum()
        lls.append(ll)
        assert abs(f.sum() - 1) < 1e-12 and (f >= 0).all()
    assert all(b >= a - 1e-6 * abs(a) for a, b in zip(lls, lls[1:])), lls
    f_run, lls_run = em.run((counts > 0) / max(1, int((counts > 0).sum())))
    assert len(lls_run) >= 2 and np.all(np.diff(lls_run) >= -1e-6 * np.abs(lls_run[:-1])) and abs(f_run.sum() - 1) < 1e-9
    post, best = em.posteriors(f_run)
    off = world["off"]
    sums = np.add.reduceat(post, off[:-1][np.diff(off) > 0])
    assert np.allclose(sums, 1.0, atol=1e-9)
    # the genomes most reads were drawn from carry the highest frequencies: top-10 by truth is within the top-25 by f, species-wise
    truth = world["truth"]
    src = np.bincount(world["genome"][truth[truth >= 0]], minlength=n_taxa)
    sp_of = np.append(world["species"], -2)                       # genome -> species; the human-like taxon last
    top_truth = {int(sp_of[g]) for g in np.argsort(-src)[:10]}
    top_f = {int(sp_of[g]) for g in np.argsort(-f_run)[:25]}
    assert len(top_truth - top_f) <= 1, (top_truth, top_f)
    em.close(); M.close()


# ---------------------------------------------------------------------------------------------- configs[2] on one GPU
def test_eight_read_shards_equal_the_whole_batch(world, monkeypatch):
    """configs[2] (100k x 10 kb reads sharded over 8 GPUs, EM sufficient statistics all-reduced per iteration) as far as one GPU can carry it"""
    from metamaps_amd import capi
    from metamaps_amd.dist import shard_range, em_distributed
    ctx, idx, reads = world["ctx"], world["idx"], world["reads"]
    n_taxa = COMM["n_genomes"] + 1
    contig_len = world["ref"].lengths().astype(np.int64)
    shards, ems, recs, offs = [], [], [], []
    for rk in range(8):
        lo, hi = shard_range(N_READS, rk, 8)
        sl = reads.slice(lo, hi - lo)
        M = ctx.map_batch(idx, sl, K, W); M.add_qualities(K)
        off, rec = M.fetch()
        offs.append((lo, off.copy(), rec.copy()))
        ems.append(ctx.em_from_mapping(M, world["genome"], contig_len, n_taxa))
        M.release_intermediates()
        shards.append((sl, M))
    off, rec = _join(offs)
    assert np.array_equal(off, world["off"]) and rec.tobytes() == world["rec"].tobytes()
    # single rank: the whole batch
    M = ctx.map_batch(idx, reads, K, W); M.add_qualities(K)
    em = ctx.em_from_mapping(M, world["genome"], contig_len, n_taxa)
    counts = em.taxon_counts()
    assert np.array_equal(counts, sum(e.taxon_counts() for e in ems))
    seen = (counts > 0).astype(np.float64)
    f_one, lls_one = em.run(seen / seen.sum())
    # eight ranks: every rank its E+M step on its own reads, the partial sums added (here: on the host), normalisation and stop rule as fEM.h:606-639
    def step(f):
        tot, ll = np.zeros(n_taxa), 0.0
        for e in ems:
            part, l = e.iterate(f)
            tot += part; ll += l
        return tot, ll
    f_eight, lls_eight = em_distributed(step, lambda v: v, seen)
    assert len(lls_eight) == len(lls_one) >= 2
    assert np.allclose(lls_eight, lls_one, rtol=1e-12, atol=0), (lls_eight, lls_one)
    assert np.allclose(f_eight, f_one, rtol=1e-9, atol=1e-15)
    post1, best1 = em.posteriors(f_one)
    best8, base = [], 0
    for e, (_lo, _o, r_) in zip(ems, offs):                       # best mapping per read: an entry index of the shard's own EM problem
        b = e.posteriors(f_eight)[1]
        best8.append(np.where(b >= 0, b + base, -1)); base += len(r_)
    best8 = np.concatenate(best8)
    differ = np.nonzero(best8 != best1)[0]                        # reads2Taxon: the same best mapping for every read (two mappings whose posteriors agree to 1e-9 may swap)
    assert len(differ) <= 5 and all(abs(post1[best8[r]] - post1[best1[r]]) < 1e-9 for r in differ), differ[:10]
    # the RCCL form of one rank's iteration (one-rank communicator, the collective kept): == its plain partial sums, normalised
    monkeypatch.setenv("MM_EM_FORCE_COLLECTIVE", "1")
    ctx.comm_init(capi.Context.comm_unique_id(), 0, 1)
    f0 = seen / seen.sum()
    for e in ems[:2]:
        part, ll = e.iterate(f0)
        fn, lla = e.iterate_allreduce(f0)
        assert np.allclose(fn, part / part.sum(), rtol=1e-12, atol=0) and abs(lla - ll) <= 1e-12 * abs(ll)
    monkeypatch.delenv("MM_EM_FORCE_COLLECTIVE")
    for e in ems:
        e.close()
    for sl, Ms in shards:
        Ms.close(); sl.close()
    em.close(); M.close()


# ---------------------------------------------------------------------------------------------- configs[3] / [4] shapes
def _microbial_subset(world, n):
    """reads of the mixed batch that stem from microbial genomes or from nowhere (a read from a human-like contig draws 1e7+ seed
    hits once no hash is cut, see below), longest first excluded: the first n such reads"""
    t = world["mtruth"]
    ok = np.nonzero((t < 0) | (world["genome"][np.maximum(t, 0)] < COMM["n_genomes"]))[0]
    return ok[:n]


def test_mixed_lengths_unchunked(world):
    """stage 1 (whole index resident): the mixed 1-50 kb PacBio batch, (a) as the bench maps it, (b) a sub-batch with freqThreshold
    off — the reference value for the chunked runs below — and the chunk plans of --maxmemory 70 GiB / 25 GiB"""
    ctx, idx, mixed = world["ctx"], world["idx"], world["mixed"]
    rl = mixed.lengths()
    assert rl.min() >= MIXED_MIN * 0.8 and rl.max() > 40_000 and np.median(rl) < 12_000      # log-uniform: half the reads below ~7 kb
    off, rec = _join([(a,) + _map(ctx, idx, h)[:2] for a, h in _halves(world)])   # two calls of 62 500 reads
    assert len(off) == N_MIXED + 1
    world["mixed_off"], world["mixed_rec"] = off, rec
    frac_mapped, frac_right = _species_recovery(world, off, rec, world["mtruth"], random_may_map=0.05)
    assert frac_mapped > 0.97 and frac_right > 0.98, (frac_mapped, frac_right)
    sums = np.add.reduceat(rec["mapq"], off[:-1][np.diff(off) > 0])
    assert np.allclose(sums, 1.0, atol=1e-9)
    which = _microbial_subset(world, 1500)
    assert len(which) == 1500
    sub = _subset(ctx, mixed, which)
    thr = idx.freq_threshold
    idx.set_freq_threshold(INT_MAX)
    world["sub_which"] = which
    world["sub_off"], world["sub_rec"], _ = _map(ctx, idx, sub)
    idx.set_freq_threshold(thr)
    sub.close()
    world["plan3"] = idx.plan_chunks(70 * GIB)
    world["plan8"] = idx.plan_chunks(25 * GIB)
    assert 3 <= len(world["plan3"]) <= 6 and 8 <= len(world["plan8"]) <= 16, (world["plan3"], world["plan8"])
    assert world["plan3"][0] == 0 and all(b > a for a, b in zip(world["plan3"], world["plan3"][1:]))


def _chunk_bounds(plan, n_contigs):
    return [(a, (plan[i + 1] if i + 1 < len(plan) else n_contigs) - a) for i, a in enumerate(plan)]


def _accumulated_thresholds(hists, uniques):
    """the reference's per-chunk freqThreshold: the occurrence histogram is never cleared between chunks (winSketch.hpp:452-494)"""
    from metamaps_amd import capi
    acc, thr, out = {}, INT_MAX, []
    for (counts, nh), u in zip(hists, uniques):
        for c, n in zip(counts.tolist(), nh.tolist()):
            acc[c] = acc.get(c, 0) + n
        cc = np.array(sorted(acc), dtype=np.int64); hh = np.array([acc[c] for c in cc.tolist()], dtype=np.int64)
        thr = capi.lib().mm_freq_threshold_from_hist(cc.ctypes.data, hh.ctypes.data, len(cc), u, thr)
        out.append(int(thr))
    return out


def test_resident_chunks_equal_whole_index(world):
    """stage 2, config 3's shape: the whole index goes, >= 3 chunk indexes built from device-side slices stay resident"""
    from metamaps_amd import capi
    ctx, ref, mixed = world["ctx"], world["ref"], world["mixed"]
    world["idx"].close(); world["idx"] = None; world["stage"] = 2
    bounds = _chunk_bounds(world["plan3"], ref.count)
    hists, uniq = [], []
    for a, n in bounds:
        sl = ref.slice(a, n)
        ix = ctx.index(sl, K, W, auto_threshold=False)
        sl.close()
        world["chunk_idx"].append(ix)
        hists.append(ix.freq_hist()); uniq.append(ix.info()["n_unique_hashes"])
    assert sum(ix.info()["n_contigs"] for ix in world["chunk_idx"]) == ref.count
    base = [a for a, _ in bounds]
    # (i) no hash cut: chunked == unchunked, record for record
    sub = _subset(ctx, mixed, world["sub_which"])
    parts = []
    for ix in world["chunk_idx"]:
        ix.set_freq_threshold(INT_MAX)
        parts.append(ctx.map_batch(ix, sub, K, W))
    U = capi.Mapping.concat(ctx, parts, base); U.add_qualities(K)
    off, rec = U.fetch()
    assert np.array_equal(off, world["sub_off"]) and len(rec) > 3000
    for fld in ("read", "ref_contig", "ref_start", "shared", "sketch", "strand"):
        assert np.array_equal(rec[fld], world["sub_rec"][fld]), fld
    assert np.allclose(rec["mapq"], world["sub_rec"]["mapq"], rtol=1e-12, atol=0)
    for p in parts:
        p.close()
    U.close(); sub.close()
    # (ii) the reference's per-chunk thresholds: records in chunk order, qualities over the union, truth as unchunked
    thr = _accumulated_thresholds(hists, uniq)
    world["thr3"] = thr
    assert thr == sorted(thr) and 100 < thr[0] and 1000 < thr[-1] < 3000, thr   # the histogram accumulates over the chunks: the cut rises towards the whole index's
    for ix, t in zip(world["chunk_idx"], thr):
        ix.set_freq_threshold(t)
    joined = []
    for a, half in _halves(world):
        parts = []
        for ix in world["chunk_idx"]:                             # (chunks 2.. reuse the sketches of the first mapping, as the CLI does; the
            parts.append(ctx.map_batch(ix, half, K, W, sketch_of=parts[0] if parts else None))   #  streamed run below computes them per chunk: the two must agree)
            if len(parts) > 1:
                parts[-1].release_intermediates()                 # (the first part keeps the sketches the others borrow)
        U = capi.Mapping.concat(ctx, parts, base); U.add_qualities(K)
        o_, r_ = U.fetch()
        joined.append((a, o_.copy(), r_.copy()))
        for p in parts:
            p.close()
        U.close()
    off, rec = _join(joined)
    world["res3_off"], world["res3_rec"] = off, rec
    key = rec["read"].astype(np.int64) << 44 | rec["ref_contig"].astype(np.int64) << 30 | rec["ref_start"].astype(np.int64)
    assert (np.diff(key) > 0).all()                               # within a read: chunk order = contig order, positions ascending
    sums = np.add.reduceat(rec["mapq"], off[:-1][np.diff(off) > 0])
    assert np.allclose(sums, 1.0, atol=1e-9)
    frac_mapped, frac_right = _species_recovery(world, off, rec, world["mtruth"], random_may_map=0.05)
    assert frac_mapped > 0.97 and frac_right > 0.98, (frac_mapped, frac_right)
    def _same(r):
        a, b = rec[off[r]:off[r + 1]], world["mixed_rec"][world["mixed_off"][r]:world["mixed_off"][r + 1]]
        return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("ref_contig", "ref_start", "shared"))
    same = sum(_same(r) for r in range(0, N_MIXED, 7))
    assert same > 0.9 * len(range(0, N_MIXED, 7))                 # (thresholds differ per chunk, so a few reads may differ from the unchunked run)


def test_streamed_chunks_equal_resident_chunks(world):
    """config 4's shape on one device: the SAME chunk rule cuts (plan3) and thresholds, but one index on the device at a time —
    built, every batch mapped, records to the host, index dropped — then mm_mapping_from_parts: identical to the resident run;
    and >= 8 chunks of the 25 GiB plan streamed the same way with freqThreshold off: identical to the whole-index result"""
    from metamaps_amd import capi
    ctx, ref, mixed = world["ctx"], world["ref"], world["mixed"]
    assert world["stage"] == 2
    for ix in world["chunk_idx"]:
        ix.close()
    world["chunk_idx"] = []
    bounds = _chunk_bounds(world["plan3"], ref.count)
    host = [[] for _ in _halves(world)]
    for (a, n), t in zip(bounds, world["thr3"]):
        sl = ref.slice(a, n); ix = ctx.index(sl, K, W, auto_threshold=False); sl.close()
        ix.set_freq_threshold(t)
        for hi_, (_first, half) in enumerate(_halves(world)):     # every read batch against the chunk that is on the device
            M = ctx.map_batch(ix, half, K, W)
            o, r = M.fetch(); host[hi_].append((o.copy(), r.copy()))
            M.close()
        ix.close()
    joined = []
    for (first, half), parts_h in zip(_halves(world), host):
        V = capi.Mapping.from_parts(ctx, half.lengths(), parts_h, [a for a, _ in bounds], K, W); V.add_qualities(K)
        o_, r_ = V.fetch()
        joined.append((first, o_.copy(), r_.copy()))
        V.close()
    off, rec = _join(joined)
    assert np.array_equal(off, world["res3_off"]) and rec.tobytes() == world["res3_rec"].tobytes()
    bounds = _chunk_bounds(world["plan8"], ref.count)
    assert len(bounds) >= 8
    sub = _subset(ctx, mixed, world["sub_which"])
    host = []
    sk = ctx.sketch_batch(sub, K, W)                              # minimizers + sketches once for all chunks (mm_sketch_batch), as the chunk-major runs do
    for a, n in bounds:
        sl = ref.slice(a, n); ix = ctx.index(sl, K, W, auto_threshold=False); sl.close()
        ix.set_freq_threshold(INT_MAX)
        M = ctx.map_batch(ix, sub, K, W, sketch_of=sk)
        o, r = M.fetch(); host.append((o.copy(), r.copy()))
        M.close(); ix.close()
    sk.close()
    V = capi.Mapping.from_parts(ctx, sub.lengths(), host, [a for a, _ in bounds], K, W); V.add_qualities(K)
    off, rec = V.fetch()
    assert np.array_equal(off, world["sub_off"])
    for fld in ("read", "ref_contig", "ref_start", "shared", "sketch", "strand"):
        assert np.array_equal(rec[fld], world["sub_rec"][fld]), fld
    assert np.allclose(rec["mapq"], world["sub_rec"]["mapq"], rtol=1e-12, atol=0)
    V.close(); sub.close()

"""BASELINE configs[4] at its size on one GPU: 100k x 10 kb-shaped reads against a full-RefSeq-scale database whose index does not fit
288 GB — multi-pass HBM streaming (SURVEY D1 "DB-refseq-scale": 140 000 genomes, ~300 Gbp, w = 6 at default flags; the index of the whole
reference would take ~1.6 TB by the size model of DESIGN.md section 3).  What `bench.py --config 5` times, as a test:

  * the reference (134 400 microbial genomes in 33 600 species + 24 human-like contigs, 299.7 Gbp) is generated on the device (75 GB packed);
  * the --maxmemory chunk rule (winSketch.hpp:274-329, limit 150 GiB) runs on the indexes of contig ranges (metamaps_amd/chunkplan.py; the CLI
    does the same in C++): ~20 chunks of ~15 Gbp;
  * ONE full pass: every chunk index is built (per-chunk freqThreshold from the histogram accumulated so far, winSketch.hpp:452-494), a batch of
    25 000 ONT-error reads is mapped against it (minimizers and sketches once for the pass: mm_sketch_batch / mm_map_batch_reusing), the index is
    dropped; the records stay on the device and are merged in chunk order (unifyFiles, mapWrap.h:128-145), mapping qualities over the union, EM.

No oracle run is feasible at this size; checked is what must hold at any size: the plan covers every contig exactly once and no chunk index
exceeds the device; thresholds do not fall as the histogram accumulates; the records of sampled chunks equal mapping the same reads directly
against that chunk's index (own sketches, no reuse); qualities sum to 1 over the union; reads come back to the species they were drawn from;
the EM's likelihood does not fall; and a chunk build takes about a second (round 4's allocator regression: 7 s per build when every build
fetched its ~135 GB fresh from the driver)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, W = 16, 6
SCALE = 11.2
COMM = dict(seed=20260928, n_genomes=int(12000 * SCALE), n_species=int(3000 * SCALE), n_genera=int(600 * SCALE), median_len=2.0e6, sigma_len=0.6, min_len=5_000,
            max_len=12_000_000, strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
            human_contigs=24, human_bases=int(3.1e9), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000,
            total_bases_target=int(26_762_276_280 * SCALE))
N_READS, RLEN = 25_000, 10_000
MAXMEM_GIB, RANGE_GBP = 150, 8
GIB = 1 << 30


@pytest.fixture(scope="module")
def run():
    from metamaps_amd import capi
    from metamaps_amd.chunkplan import plan_chunks_by_ranges, chunk_bounds, AccumulatedThreshold
    t0 = time.time()
    ctx = capi.Context(0)
    total = ctx.device_info()["hbm_total"]
    if total and total < 200 * GIB:                               # a 300 Gbp reference (75 GB packed) + ~135 GB chunk indexes: the fixture is sized for an MI355X
        ctx.close()
        pytest.skip(f"needs a ~288 GB device (this one has {total / GIB:.0f} GiB)")
    ref, genome = ctx.synth_community(**COMM)
    species = capi.Context.synth_community_species(**COMM)
    contig_len = ref.lengths().astype(np.int32)
    t_ref = time.time() - t0
    plan, pinfo = plan_chunks_by_ranges(ctx, ref, contig_len, K, W, MAXMEM_GIB * GIB, int(RANGE_GBP * 1e9))
    bounds = chunk_bounds(plan, ref.count)
    t_plan = time.time() - t0 - t_ref
    reads, truth = ctx.synth_reads(ref, seed=4243, n_reads=N_READS, read_len=RLEN, sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
    # ---- one pass
    tp = time.time()
    sk = ctx.sketch_batch(reads, K, W)
    acc = AccumulatedThreshold()
    parts, thrs, t_build, infos, direct = [], [], [], [], {}
    sampled = {3, len(bounds) // 2, len(bounds) - 1}
    for ci, (a, n) in enumerate(bounds):
        tb = time.time()
        sl = ref.slice(a, n); ix = ctx.index(sl, K, W, auto_threshold=False); sl.close()
        ctx.synchronize(); t_build.append(time.time() - tb)
        thrs.append(acc.next(ix)); infos.append(ix.info())
        M = ctx.map_batch(ix, reads, K, W, sketch_of=sk)
        if ci in sampled:
            o_, r_ = M.fetch()
            D = ctx.map_batch(ix, reads, K, W)                     # the same reads against this chunk alone, their sketches computed again
            od, rd = D.fetch()
            direct[ci] = (o_.copy(), r_.copy(), od.copy(), rd.copy())
            D.close()
        M.release_intermediates()
        parts.append(M)
        ix.close()
    sk.close()
    U = capi.Mapping.concat(ctx, parts, [a for a, _ in bounds])
    for p in parts:
        p.close()
    U.add_qualities(K)
    off, rec = U.fetch()
    t_pass = time.time() - tp
    info = ctx.device_info()
    out = dict(ctx=ctx, ref=ref, genome=genome, species=species, contig_len=contig_len, plan=plan, bounds=bounds, pinfo=pinfo, reads=reads, truth=truth, thrs=thrs, t_build=t_build,
               infos=infos, direct=direct, U=U, off=off.copy(), rec=rec.copy(), device=info, t_ref=t_ref, t_plan=t_plan, t_pass=t_pass)
    print(f"refseq-scale: reference {ref.total_bases / 1e9:.1f} Gbp in {t_ref:.1f} s, plan of {len(bounds)} chunks in {t_plan:.1f} s, pass {t_pass:.1f} s "
          f"(builds {min(t_build):.2f}-{max(t_build):.2f} s), {len(rec)} records")
    yield out
    U.close(); reads.close(); ref.close(); ctx.close()


def test_reference_is_refseq_scale_and_the_plan_covers_it(run):
    ref, bounds = run["ref"], run["bounds"]
    assert ref.count == COMM["n_genomes"] + COMM["human_contigs"] and abs(ref.total_bases - COMM["total_bases_target"]) < 0.01 * COMM["total_bases_target"]
    assert 12 <= len(bounds) <= 40, len(bounds)
    covered = np.zeros(ref.count, dtype=np.int32)
    for a, n in bounds:
        assert n > 0
        covered[a:a + n] += 1
    assert (covered == 1).all()                                    # every contig in exactly one chunk, chunks in contig order
    assert run["plan"][0] == 0 and all(b > a for a, b in zip(run["plan"], run["plan"][1:]))
    total = run["device"]["hbm_total"]
    size_model_whole = sum(i["hbm_bytes"] for i in run["infos"])
    assert size_model_whole > total                                # the chunk indexes together do not fit the device: streaming is forced, not chosen
    for i, (a, n) in zip(run["infos"], bounds):
        assert i["n_contigs"] == n and i["hbm_bytes"] < 0.75 * total, i   # one chunk index at a time beside the 75 GB packed reference
    assert sum(i["n_entries"] for i in run["infos"]) > 6e10


def test_thresholds_follow_the_accumulated_histogram(run):
    thrs = run["thrs"]
    assert thrs == sorted(thrs) and thrs[0] >= 20 and thrs[-1] > thrs[0], thrs   # (winSketch.hpp:452-494: the histogram is never cleared)


def test_sampled_chunks_equal_direct_mapping(run):
    assert len(run["direct"]) >= 2
    n_rec = 0
    for ci, (o, r, od, rd) in run["direct"].items():
        assert np.array_equal(o, od) and r.tobytes() == rd.tobytes(), ci
        n_rec += len(r)
    assert n_rec > 0


def test_union_records_and_qualities(run):
    off, rec = run["off"], run["rec"]
    assert len(off) == N_READS + 1 and len(rec) > N_READS // 2
    assert (rec["shared"] <= rec["sketch"]).all() and (rec["shared"] > 0).all() and np.isin(rec["strand"], (-1, 1)).all()
    key = rec["read"].astype(np.int64) << 48 | rec["ref_contig"].astype(np.int64) << 30 | rec["ref_start"].astype(np.int64)
    assert (np.diff(key) > 0).all()                               # (read, contig of the WHOLE reference, position): chunk order = contig order
    assert rec["ref_contig"].max() < run["ref"].count
    sums = np.add.reduceat(rec["mapq"], off[:-1][np.diff(off) > 0])
    assert np.allclose(sums, 1.0, atol=1e-9)                      # mapWrap.h:215-323 over the union of the chunks
    # the records of a sampled chunk are the union's records on that chunk's contigs
    for ci, (o, r, _od, _rd) in run["direct"].items():
        a, n = run["bounds"][ci]
        sel = (rec["ref_contig"] >= a) & (rec["ref_contig"] < a + n)
        assert sel.sum() == len(r)
        assert np.array_equal(rec["ref_contig"][sel] - a, r["ref_contig"]) and np.array_equal(rec["ref_start"][sel], r["ref_start"]) and np.array_equal(rec["shared"][sel], r["shared"])


def test_truth_recovery_at_species_level(run):
    off, rec, truth, genome, species = run["off"], run["rec"], run["truth"], run["genome"], run["species"]
    ng = COMM["n_genomes"]
    cs = np.where(genome < ng, species[np.minimum(genome, ng - 1)], -2)
    n_map = np.diff(off)
    from_genome = truth >= 0
    assert (n_map[~from_genome] > 0).mean() <= 0.02                # (random sequence against 300 Gbp at w = 6: the window size is chosen for that, map_stats.hpp:226)
    mapped = from_genome & (n_map > 0)
    good = 0
    for r in np.nonzero(mapped)[0]:
        seg = rec[off[r]:off[r + 1]]
        best = seg[np.argmax(seg["mapq"])]
        good += cs[int(best["ref_contig"])] == cs[int(truth[r])]
    frac_mapped, frac_right = mapped.sum() / max(1, from_genome.sum()), good / max(1, int(mapped.sum()))
    assert frac_mapped > 0.98 and frac_right > 0.98, (frac_mapped, frac_right)


def test_em_over_the_union(run):
    ctx, U = run["ctx"], run["U"]
    n_taxa = COMM["n_genomes"] + 1
    em = ctx.em_from_mapping(U, run["genome"], run["contig_len"].astype(np.int64), n_taxa)
    seen = (em.taxon_counts() > 0).astype(np.float64)
    f, lls = em.run(seen / seen.sum())
    assert len(lls) >= 2 and np.all(np.diff(lls) >= -1e-6 * np.abs(lls[:-1])) and abs(f.sum() - 1) < 1e-9
    em.close()


def test_chunk_builds_take_about_a_second(run):
    tb = run["t_build"]
    # the first build of the pass may fetch blocks of a size the range indexes of the plan never asked for from the driver (cleared as they are
    # handed out, ~25 GB/s); from then on a chunk's arrays land in the previous chunk's pooled blocks
    # (a regression guard with generous bounds, not a benchmark — round 4's allocator regression was 7 s per build; a build right behind another
    #  process's release of the device waits for the driver's wipe, profiles/r05_index_build_processes.txt: 5-6 s — so the median decides, not the worst)
    assert float(np.median(tb)) <= 3.0, tb
    assert run["t_pass"] < 240, run["t_pass"]

"""Several GPUs through the drop-in CLI and the RCCL path of the library.

The GPU test box has one MI355X, so:
  * the RCCL code path runs with a one-rank communicator (ncclCommInitRank + ncclAllReduce are really executed);
  * the multi-device orchestration of the CLI (order-preserving batch hand-out, chunk indexes spread over devices, records
    gathered to a batch's owner, rounds of streamed chunks) runs with several LOGICAL devices on the one physical GPU
    (`--devices 0,0,0`: one context each — exactly the code a real `--gpus 3` runs, minus peer hardware);
  * `--gpus 2` on two physical devices runs wherever two are visible and is skipped otherwise.
Every result is compared with the oracle CLI (bit-exact text, floats 1e-5)."""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_cli import CLI, _cmp_table

pytestmark = pytest.mark.gpu


def _classify_files_equal(pa, pb):
    _cmp_table(pa + ".EM", pb + ".EM", " ", {13})
    assert open(pa + ".EM.reads2Taxon").read() == open(pb + ".EM.reads2Taxon").read()
    _cmp_table(pa + ".EM.reads2Taxon.krona", pb + ".EM.reads2Taxon.krona", "\t", {2})
    _cmp_table(pa + ".EM.WIMP", pb + ".EM.WIMP", "\t", {4, 5})
    _cmp_table(pa + ".EM.lengthAndIdentitiesPerMappingUnit", pb + ".EM.lengthAndIdentitiesPerMappingUnit", "\t", {3})
    _cmp_table(pa + ".EM.contigCoverage", pb + ".EM.contigCoverage", "\t", {6})


def _map_files_equal(pa, pb):
    _cmp_table(pa, pb, " ", {13})
    for suf in (".meta", ".meta.unmappedReadsLengths"):
        assert open(pa + suf).read() == open(pb + suf).read(), suf


@pytest.fixture(scope="module")
def small_run(tmp_path_factory, oracle_lib):
    """a 10-genome DB, 400 reads in two query files (so that batches, files and chunk passes all have borders), oracle outputs"""
    import orc
    from metamaps_amd import synth
    d = tmp_path_factory.mktemp("multi")
    db = synth.make_db(str(d / "db"), n_genomes=10, genome_len=60_000, seed=7)
    r1 = synth.make_reads(db, str(d / "r1.fq"), n_reads=260, read_len=3000, seed=3)
    r2 = synth.make_reads(db, str(d / "r2.fq"), n_reads=140, read_len=2500, seed=4)
    ora = {}
    for tag, extra in (("plain", []), ("chunks", ["--maxmemory-bytes", "1000000"])):
        o1, o2 = str(d / f"cpu_{tag}1"), str(d / f"cpu_{tag}2")
        subprocess.run([orc.CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", r1["path"] + "," + r2["path"], "-o", o1 + "," + o2] + extra,
                       check=True, capture_output=True, timeout=900)
        ora[tag] = (o1, o2)
    subprocess.run([orc.CLI, "classify", "--DB", db.dir, "--mappings", ora["plain"][0], "--minreads", "3"], check=True, capture_output=True, timeout=900)
    return {"dir": d, "db": db, "queries": r1["path"] + "," + r2["path"], "oracle": ora}


def _gpu_map(run, tag, extra):
    d = run["dir"]
    o1, o2 = str(d / f"gpu_{tag}1"), str(d / f"gpu_{tag}2")
    # MM_CLI_BATCH_READS: batches of 64 reads, so that several batches are in flight on every worker
    env = dict(os.environ, MM_CLI_BATCH_READS="64")
    p = subprocess.run([CLI, "mapDirectly", "--all", "-r", run["db"].fasta, "-q", run["queries"], "-o", o1 + "," + o2] + extra,
                       capture_output=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return o1, o2, p.stdout.decode()


@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0"])
def test_replicated_index_batches_over_devices(small_run, devices):
    """index replicated per device, read batches handed to whichever worker is free, output in input order"""
    o1, o2, _ = _gpu_map(small_run, "rep" + devices.replace(",", ""), ["--devices", devices])
    _map_files_equal(o1, small_run["oracle"]["plain"][0])
    _map_files_equal(o2, small_run["oracle"]["plain"][1])


@pytest.mark.parametrize("devices,mode,gather", [("0,0", "--shard-index", None), ("0,0,0", "--shard-index", None), ("0,0", "--stream-chunks", None), ("0,0,0", "--stream-chunks", None),
                                                 ("0", "--shard-index", None), ("0,0,0", "--shard-index", "--host-gather"), ("0,0", "--stream-chunks", "--host-gather")])
def test_chunk_indexes_spread_over_devices(small_run, devices, mode, gather):
    """--maxmemory chunks, chunk c on device c mod N (all resident, or N at a time), every batch visits every device, records
    gathered on the batch's owner — device to device (the parts stay on the device that mapped them; logical devices of one GPU: copies
    inside mm_mapping_concat; physical devices: RCCL, tools/scale_check.sh) or, --host-gather, through host memory as in rounds 1-3:
    same files as the oracle under the same --maxmemory"""
    o1, o2, out = _gpu_map(small_run, mode.strip("-")[:5] + devices.replace(",", "") + (gather or "").strip("-")[:4],
                           ["--devices", devices, mode, "--maxmemory-bytes", "1000000"] + ([gather] if gather else []))
    assert sum(1 for l in out.splitlines() if l.startswith("INFO, index chunk")) >= 3
    _map_files_equal(o1, small_run["oracle"]["chunks"][0])
    _map_files_equal(o2, small_run["oracle"]["chunks"][1])


def test_replicated_chunks_over_devices(small_run):
    """--maxmemory with every chunk index on every device (BASELINE config 4's shape: miniSeq+H chunks all fit one GPU)"""
    o1, o2, _ = _gpu_map(small_run, "repchunks", ["--devices", "0,0", "--maxmemory-bytes", "1000000"])
    _map_files_equal(o1, small_run["oracle"]["chunks"][0])
    _map_files_equal(o2, small_run["oracle"]["chunks"][1])


def test_classify_through_rccl_one_rank(small_run):
    """`classify --gpus 1` with MM_EM_FORCE_COLLECTIVE=1: the EM loop goes through ncclCommInitRank / kernel A | ncclAllReduce | kernel B (one rank)
    and writes the oracle's files; without the switch the one-rank communicator takes the resident kernel: the same files"""
    o1, _, _ = _gpu_map(small_run, "cls", [])
    for env in (dict(os.environ, MM_EM_FORCE_COLLECTIVE="1"), dict(os.environ)):
        p = subprocess.run([CLI, "classify", "--DB", small_run["db"].dir, "--mappings", o1, "--minreads", "3", "--gpus", "1"], capture_output=True, timeout=900, env=env)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        _classify_files_equal(o1, small_run["oracle"]["plain"][0])


def test_two_physical_gpus(small_run):
    """--gpus 2: mapDirectly (replicated and sharded) and classify (reads sharded, RCCL all-reduce over xGMI) == --gpus 1 == oracle"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    for tag, extra, ref in (("g2", ["--gpus", "2"], "plain"), ("g2s", ["--gpus", "2", "--shard-index", "--maxmemory-bytes", "1000000"], "chunks")):
        o1, o2, _ = _gpu_map(small_run, tag, extra)
        _map_files_equal(o1, small_run["oracle"][ref][0])
        _map_files_equal(o2, small_run["oracle"][ref][1])
    o1, _, _ = _gpu_map(small_run, "g2c", ["--gpus", "2"])
    p = subprocess.run([CLI, "classify", "--DB", small_run["db"].dir, "--mappings", o1, "--minreads", "3", "--gpus", "2"], capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    _classify_files_equal(o1, small_run["oracle"]["plain"][0])


def test_allreduce_iteration_equals_local_iteration(monkeypatch):
    """mm_comm_init(nranks = 1), then mm_em_iterate_allreduce (device partial sums -> ncclAllReduce -> normalise) must equal
    mm_em_iterate + the same normalisation on the host, bit for bit; mm_comm_allreduce_f64 is the identity"""
    from metamaps_amd import capi
    rng = np.random.default_rng(11)
    n_reads, n_taxa = 5000, 37
    per = rng.integers(1, 9, size=n_reads)
    off = np.concatenate([[0], np.cumsum(per)]).astype(np.int64)
    ne = int(off[-1])
    taxon = rng.integers(0, n_taxa, size=ne).astype(np.int32)
    mapq = rng.random(ne)
    inv = 1.0 / rng.integers(1000, 5_000_000, size=ne).astype(np.float64)
    f = np.full(n_taxa, 1.0 / n_taxa)
    ctx_a, ctx_b = capi.Context(0), capi.Context(0)
    ctx_b.comm_init(capi.Context.comm_unique_id(), 0, 1)
    ea, eb = ctx_a.em(off, taxon, mapq, inv, n_taxa), ctx_b.em(off, taxon, mapq, inv, n_taxa)
    v = rng.random(n_taxa + 1)
    w = v.copy(); ctx_b.comm_allreduce(w)
    assert np.array_equal(v, w)
    for _ in range(6):
        part, ll_a = ea.iterate(f)
        tot = 0.0
        for x in part:                                            # the library normalises with a sequential sum (fEM.h:606-615)
            tot += float(x)
        f_a = part / tot
        f_b, ll_b = eb.iterate_allreduce(f)
        assert np.array_equal(f_a, f_b) and ll_a == ll_b
        f = f_b
    # the device-resident loop (mm_em_run), with and without the communicator, against the host-driven loop above
    from metamaps_amd import emhost
    monkeypatch.setenv("MM_EM_FORCE_COLLECTIVE", "1")             # the loop with the ncclAllReduce inside, also on this one rank
    f_ref, lls_ref = emhost.run_em(lambda x: ea.iterate_allreduce(x), n_taxa)
    for e in (ea, eb):
        f_run, lls_run = e.run(np.full(n_taxa, 1.0 / n_taxa))
        assert len(lls_run) == len(lls_ref) and np.allclose(lls_run, lls_ref, rtol=1e-13, atol=0)
        assert np.allclose(f_run, f_ref, rtol=1e-12, atol=1e-300)
    ea.close(); eb.close(); ctx_a.close(); ctx_b.close()



def _em_problem(seed, n_reads, n_taxa, sigma=2.0, tied=True):
    """skewed abundances, ambiguous reads, a pair of exactly tied taxa (identical mapping lists), one very abundant taxon (many sum items)"""
    rng = np.random.default_rng(seed)
    ab = rng.lognormal(0, sigma, n_taxa); ab /= ab.sum()
    true = rng.choice(n_taxa, size=n_reads, p=ab)
    n_extra = rng.integers(0, 6, size=n_reads)
    off = np.concatenate([[0], np.cumsum(1 + n_extra)]).astype(np.int64)
    taxon = np.empty(int(off[-1]), dtype=np.int32)
    taxon[off[:-1]] = true
    for r in np.nonzero(n_extra)[0]:
        taxon[off[r] + 1:off[r + 1]] = rng.integers(0, n_taxa, size=n_extra[r])
    if tied and n_taxa > 4:                                       # every mapping on taxon 1 also goes, identically, to taxon 2: exactly tied sums
        taxon[taxon == 2] = 3
        t1 = taxon == 1
        idx = np.nonzero(t1)[0]
        taxon2 = taxon.copy()
        # rebuild with a twin entry behind every taxon-1 entry
        rd = np.searchsorted(off, idx, side="right") - 1
        add = np.bincount(rd, minlength=n_reads)
        noff = np.concatenate([[0], np.cumsum(np.diff(off) + add)]).astype(np.int64)
        nt = np.empty(int(noff[-1]), dtype=np.int32)
        src = np.repeat(np.arange(len(taxon)), 1 + t1.astype(np.int64))
        nt[:] = taxon2[src]
        twin = np.concatenate([[False], src[1:] == src[:-1]])
        nt[twin] = 2
        taxon, off = nt, noff
        mapq_src = rng.uniform(0.05, 1.0, len(taxon2))[src]
    else:
        mapq_src = rng.uniform(0.05, 1.0, len(taxon))
    inv = 1.0 / rng.integers(1000, 5_000_000, size=n_taxa).astype(np.float64)
    inv[2] = inv[1]
    return off, taxon, mapq_src, inv[taxon], n_taxa


@pytest.mark.parametrize("n_reads,n_taxa", [(60_000, 700), (900, 40), (3, 5)])
def test_em_resident_kernel_equals_its_phases_as_launches(n_reads, n_taxa, monkeypatch):
    """the whole EM run as ONE resident kernel (MM_EM_RESIDENT=1: grid barriers between E step, per-taxon sums and normalisation + stop rule) gives, bit for bit,
    what the same phases give as separate launches (the default since the two were measured against each other), also when a barrier gives up mid-run (MM_EM_BARRIER_TICKS=1: the run goes on
    phase by phase from the last completed iteration); other grid sizes (another summation shape of the log-likelihood) agree to 1e-12; with a
    one-rank communicator (kernel A | ncclAllReduce | kernel B per iteration) as well; exactly tied taxa stay exactly tied in every form"""
    from metamaps_amd import capi, emhost
    off, taxon, mapq, inv, T = _em_problem(7 + n_reads, n_reads, n_taxa)
    f0 = np.full(T, 1.0 / T)

    def run(env, comm=False):
        for kk in ("MM_EM_SPLIT", "MM_EM_RESIDENT", "MM_EM_GRID", "MM_EM_BARRIER_TICKS", "MM_EM_FORCE_COLLECTIVE"):
            monkeypatch.delenv(kk, raising=False)
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        if comm:
            monkeypatch.setenv("MM_EM_FORCE_COLLECTIVE", "1")     # (a one-rank communicator alone takes the resident kernel: nothing to exchange)
        ctx = capi.Context(0)
        if comm:
            ctx.comm_init(capi.Context.comm_unique_id(), 0, 1)
        e = ctx.em(off, taxon, mapq, inv, T)
        f, lls = e.run(f0)
        f5, lls5 = e.run(f0, max_iter=2)
        fc, llsc, stopped = e.continue_run(1000)
        post, best = e.posteriors(f)
        e.close(); ctx.close()
        assert stopped and np.array_equal(np.concatenate([lls5, llsc]), lls) and np.array_equal(fc, f)
        return f, lls, best

    G = {"MM_EM_GRID": "96"}                                      # (the two forms default to different grids; the log-likelihood is summed in the grid's shape)
    f_a, ll_a, best_a = run({"MM_EM_RESIDENT": "1", **G})
    assert len(ll_a) >= 3 and abs(f_a.sum() - 1) < 1e-12
    if T > 4:
        assert f_a[1] == f_a[2] and f_a[1] > 0                    # the twins
    f_b, ll_b, best_b = run(G)                                    # the default form: one launch per phase
    assert np.array_equal(f_a, f_b) and np.array_equal(ll_a, ll_b) and np.array_equal(best_a, best_b)
    f_3, ll_3, best_3 = run({"MM_EM_SPLIT": "2", **G})            # P1 | P2 + P3 (the workgroup that finishes its items last runs P3: measured slower, kept as the record)
    assert np.array_equal(f_a, f_3) and np.array_equal(ll_a, ll_3) and np.array_equal(best_a, best_3)
    f_c, ll_c, best_c = run({"MM_EM_RESIDENT": "1", "MM_EM_BARRIER_TICKS": "1", **G})
    assert np.array_equal(f_a, f_c) and np.array_equal(ll_a, ll_c)
    f_0, ll_0, _ = run({})                                        # the default grid
    assert len(ll_0) == len(ll_a) and np.allclose(ll_0, ll_a, rtol=1e-12, atol=0) and np.allclose(f_0, f_a, rtol=1e-10, atol=1e-300)
    f_d, ll_d, _ = run({"MM_EM_DBG": "3"})                        # P1 thread-per-read (the form blocks too large for the LDS buffers take): another summation order of ll only
    monkeypatch.delenv("MM_EM_DBG", raising=False)
    assert len(ll_d) == len(ll_a) and np.allclose(ll_d, ll_a, rtol=1e-12, atol=0) and np.allclose(f_d, f_a, rtol=1e-10, atol=1e-300)
    for grid in ("1", "7", "256"):
        f_g, ll_g, _ = run({"MM_EM_GRID": grid} if grid == "7" else {"MM_EM_GRID": grid, "MM_EM_RESIDENT": "1"})
        assert len(ll_g) == len(ll_a) and np.allclose(ll_g, ll_a, rtol=1e-12, atol=0) and np.allclose(f_g, f_a, rtol=1e-10, atol=1e-300)
        if T > 4:
            assert f_g[1] == f_g[2]
    for env in ({}, {"MM_EM_SPLIT": "2"}, {"MM_EM_RESIDENT": "1"}, {"MM_EM_RESIDENT": "1", "MM_EM_BARRIER_TICKS": "1"}):
        f_m, ll_m, _ = run(env, comm=True)
        assert len(ll_m) == len(ll_a) and np.allclose(ll_m, ll_a, rtol=1e-12, atol=0) and np.allclose(f_m, f_a, rtol=1e-10, atol=1e-300)
        if T > 4:
            assert f_m[1] == f_m[2]
    monkeypatch.delenv("MM_EM_FORCE_COLLECTIVE", raising=False); monkeypatch.delenv("MM_EM_RESIDENT", raising=False); monkeypatch.delenv("MM_EM_BARRIER_TICKS", raising=False)
    ctx = capi.Context(0)                                         # a one-rank communicator without the switch: no collective, bit for bit the plain run
    ctx.comm_init(capi.Context.comm_unique_id(), 0, 1)
    e = ctx.em(off, taxon, mapq, inv, T)
    f_1, ll_1 = e.run(f0)
    e.close(); ctx.close()
    assert np.array_equal(f_1, f_0) and np.array_equal(ll_1, ll_0)
    # against the host-driven loop (mm_em_iterate per iteration, numpy normalisation and stop rule)
    ctx = capi.Context(0)
    e = ctx.em(off, taxon, mapq, inv, T)
    f_ref, lls_ref = emhost.run_em(lambda x: e.iterate_allreduce(x), T)
    e.close(); ctx.close()
    assert len(lls_ref) == len(ll_a) and np.allclose(lls_ref, ll_a, rtol=1e-12, atol=0) and np.allclose(f_ref, f_a, rtol=1e-10, atol=1e-300)


def test_records_gathered_from_parts_equal_device_concat(oracle_lib, monkeypatch):
    """mm_mapping_from_parts (host-side parts of chunks mapped elsewhere) == mm_mapping_concat, incl. mapping qualities"""
    from metamaps_amd import capi, synth
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        db = synth.make_db(os.path.join(d, "db"), n_genomes=8, genome_len=50_000, seed=3)
        rd = synth.make_reads(db, os.path.join(d, "r.fq"), n_reads=150, read_len=3000, seed=2)
        reads = [l.strip() for i, l in enumerate(open(rd["path"], "rb")) if i % 4 == 1]
        contigs = [s.tobytes() for s in db.contig_seqs]
    ctx, ctx2 = capi.Context(0), capi.Context(0)
    k, w = 16, 10
    half = len(contigs) // 2
    R = ctx.seqset(reads)
    parts, host = [], []
    for a, b in ((0, half), (half, len(contigs))):
        S = ctx.seqset(contigs[a:b]); idx = ctx.index(S, k, w)
        M = ctx.map_batch(idx, R, k, w)
        parts.append(M); host.append(M.fetch())
    U = capi.Mapping.concat(ctx, parts, [0, half]); U.add_qualities(k)
    lens = R.lengths()
    V = capi.Mapping.from_parts(ctx2, lens, host, [0, half], k, w); V.add_qualities(k)
    (oa, ra), (ob, rb) = U.fetch(), V.fetch()
    assert np.array_equal(oa, ob) and len(ra) > 100
    for fld in ("read", "ref_contig", "ref_start", "shared", "sketch", "strand", "mapq"):
        assert np.array_equal(ra[fld], rb[fld]), fld
    # mm_mapping_gather: no communicator (local merge), a one-rank communicator, and — MM_GATHER_SELF_SEND=1 — the owner's own parts through
    # ncclSend / ncclRecv (offsets, then records): the RCCL exchange of the sharded-index mode, as far as one GPU can run it
    ctx3 = capi.Context(0)
    ctx3.comm_init(capi.Context.comm_unique_id(), 0, 1)
    parts3 = []
    for a, b in ((0, half), (half, len(contigs))):
        S = ctx3.seqset(contigs[a:b]); idx3 = ctx3.index(S, k, w)
        R3 = ctx3.seqset(reads)
        parts3.append(ctx3.map_batch(idx3, R3, k, w))
    for c, pp, env in ((ctx, parts, None), (ctx3, parts3, None), (ctx3, parts3, "1")):
        if env:
            monkeypatch.setenv("MM_GATHER_SELF_SEND", env)
        G = capi.Mapping.gather(c, 0, lens, pp, [0, 1], [0, 0], [0, half], k, w); G.add_qualities(k)
        monkeypatch.delenv("MM_GATHER_SELF_SEND", raising=False)
        og, rg = G.fetch()
        assert np.array_equal(oa, og) and ra.tobytes() == rg.tobytes()
        assert G.stats()["sum_hits"] == U.stats()["sum_hits"]
        G.close()
    Y = capi.Mapping.concat(ctx2, parts, [0, half]); Y.add_qualities(k)      # parts of another context of the device
    oy, ry = Y.fetch()
    assert np.array_equal(oa, oy) and ra.tobytes() == ry.tobytes()
    ctx.close(); ctx2.close(); ctx3.close()


# ---- the multi-rank classify bookkeeping (shard ranges, shard-local offsets, best-mapping rebasing, final f), everything around the
# collective: `--em-host-reduce` adds the ranks' partial sums on the host in rank order (what the all-reduce delivers), so several
# ranks may share the one GPU of the test box (run_em_sharded, csrc/host/metamaps_main.cpp; fEM.h:583-600, :1229)
def _classify(prefix, db, extra, env=None):
    p = subprocess.run([CLI, "classify", "--DB", db, "--mappings", prefix, "--minreads", "3"] + extra, capture_output=True, timeout=900,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout.decode()


def _em_log(out):
    return [l for l in out.splitlines() if l.startswith("EM round") or "Log likelihood" in l or "Improvement" in l or "Relative" in l]


def _copy_run(src, dst, n_reads=None):
    """the mapping run `src` as prefix `dst`; n_reads: only the lines of the first n mapped reads (the .meta follows)"""
    import shutil
    lines = open(src).read().splitlines(keepends=True)
    meta = dict(l.split() for l in open(src + ".meta"))
    if n_reads is not None:
        ids, keep = [], []
        for l in lines:
            rid = l.split(" ", 1)[0]
            if not ids or ids[-1] != rid:
                if len(ids) == n_reads:
                    break
                ids.append(rid)
            keep.append(l)
        dropped = int(meta["ReadsMapped"]) - len(ids)
        meta["ReadsMapped"] = str(len(ids)); meta["TotalReads"] = str(int(meta["TotalReads"]) - dropped)
        lines = keep
    open(dst, "w").write("".join(lines))
    open(dst + ".meta", "w").write("".join(f"{k} {v}\n" for k, v in meta.items()))
    shutil.copy(src + ".meta.unmappedReadsLengths", dst + ".meta.unmappedReadsLengths")


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_classify_shards_with_host_reduce_equal_oracle(small_run, devices):
    """G = 2, 3 EM ranks on one device: .EM, reads2Taxon, WIMP, coverage == one rank == oracle; the EM log has the same rounds"""
    o1, _, _ = _gpu_map(small_run, "hr", [])
    d = small_run["dir"]
    one, many = str(d / "hr_one"), str(d / ("hr_" + devices.replace(",", "")))
    _copy_run(o1, one); _copy_run(o1, many)
    log_one = _em_log(_classify(one, small_run["db"].dir, []))
    log_many = _em_log(_classify(many, small_run["db"].dir, ["--devices", devices, "--em-host-reduce"]))
    assert len(log_one) == len(log_many) > 4 and [l for l in log_one if l.startswith("EM round")] == [l for l in log_many if l.startswith("EM round")]
    _classify_files_equal(many, one)
    _classify_files_equal(many, small_run["oracle"]["plain"][0])


@pytest.mark.parametrize("n_reads", [1, 2, 3, 4, 7, 10])
def test_classify_shard_boundaries(small_run, n_reads):
    """few reads, so that shards are empty (NR < G) or one read long and every cut falls on another read boundary:
    G = 2, 3 (host reduce) == one rank, reads2Taxon byte for byte, posteriors and frequencies 1e-5"""
    o1, _, _ = _gpu_map(small_run, "hb", [])
    d = small_run["dir"]
    one = str(d / f"hb_{n_reads}_one")
    _copy_run(o1, one, n_reads)
    _classify(one, small_run["db"].dir, [])
    assert sum(1 for _ in open(one + ".EM.reads2Taxon")) >= n_reads
    for devices in ("0,0", "0,0,0"):
        many = str(d / f"hb_{n_reads}_{devices.replace(',', '')}")
        _copy_run(o1, many, n_reads)
        _classify(many, small_run["db"].dir, ["--devices", devices, "--em-host-reduce"])
        _classify_files_equal(many, one)


def test_classify_em_log_in_slices(small_run):
    """the device-resident loop run in slices of 3 iterations (mm_em_run + mm_em_continue) prints and computes what one call does;
    an iteration cap that is no multiple of the enqueue group ends exactly there"""
    o1, _, _ = _gpu_map(small_run, "sl", [])
    d = small_run["dir"]
    a, b, c = str(d / "sl_a"), str(d / "sl_b"), str(d / "sl_c")
    for x in (a, b, c):
        _copy_run(o1, x)
    log_a = _em_log(_classify(a, small_run["db"].dir, ["--gpus", "1"]))
    for sl in ("1", "2", "3", "4"):                               # (round-3 advisor: a slice whose last iteration is the converging one must not run on)
        log_b = _em_log(_classify(b, small_run["db"].dir, ["--gpus", "1"], {"MM_EM_SLICE": sl, "MM_EM_FORCE_COLLECTIVE": "1"} if sl == "3" else {"MM_EM_SLICE": sl}))
        assert log_a == log_b and len(log_a) > 4, sl
        for suf in (".EM", ".EM.WIMP", ".EM.reads2Taxon"):
            assert open(a + suf).read() == open(b + suf).read(), (suf, sl)
    log_c = _em_log(_classify(c, small_run["db"].dir, ["--gpus", "1"], {"MM_EM_MAX_ITER": "3"}))
    assert [l for l in log_c if l.startswith("EM round")] == ["EM round 0", "EM round 1", "EM round 2"]


def test_em_run_stops_at_max_iter_and_continues():
    """mm_em_run(max_iter = 5) does 5 iterations, not a whole enqueue group of 8 (round-2 advisor finding); mm_em_continue goes on
    from there to what an uninterrupted run gives"""
    from metamaps_amd import capi, emhost
    rng = np.random.default_rng(5)
    n_reads, n_taxa = 3000, 23
    ab = rng.lognormal(0, 2.0, n_taxa); ab /= ab.sum()           # skewed abundances + ambiguous reads: a dozen EM rounds
    true = rng.choice(n_taxa, size=n_reads, p=ab)
    off, taxon, mapq = [0], [], []
    for r in range(n_reads):
        others = rng.choice(n_taxa, size=rng.integers(1, 6), replace=False)
        ts = [int(true[r])] + [int(o) for o in others if o != true[r]]
        taxon += ts; mapq += [0.5] + list(rng.uniform(0.35, 0.5, len(ts) - 1)); off.append(len(taxon))
    off = np.array(off, dtype=np.int64); taxon = np.array(taxon, dtype=np.int32); mapq = np.array(mapq)
    inv = np.full(len(taxon), 1e-6)
    ctx = capi.Context(0)
    e = ctx.em(off, taxon, mapq, inv, n_taxa)
    f0 = np.full(n_taxa, 1.0 / n_taxa)
    f_all, lls_all = e.run(f0)
    assert len(lls_all) > 6
    f_ref, lls_ref = emhost.run_em(lambda x: e.iterate_allreduce(x), n_taxa)
    assert len(lls_ref) == len(lls_all)
    assert len(lls_all) >= 11
    for cap in (1, 5, 9):
        if cap + 2 >= len(lls_all):
            continue
        f5, lls5 = e.run(f0, max_iter=cap)
        assert len(lls5) == cap and np.array_equal(lls5, lls_all[:cap])
        f_host = f0
        for _ in range(cap):
            f_host, _ll = e.iterate_allreduce(f_host)
        assert np.allclose(f5, f_host, rtol=1e-12, atol=1e-300)
        f_c, lls_c, stopped = e.continue_run(2)
        assert np.array_equal(lls_c, lls_all[cap:cap + 2]) and not stopped
        f_c, lls_c, stopped = e.continue_run(1000)
        assert stopped and np.array_equal(np.concatenate([lls5, lls_all[cap:cap + 2], lls_c])[:len(lls_all)], lls_all)
        assert np.array_equal(f_c, f_all)
        f_c2, lls_c2, stopped = e.continue_run(10)               # a stopped run stays stopped
        assert stopped and len(lls_c2) == 0 and np.array_equal(f_c2, f_all)
    e.close(); ctx.close()


def test_classify_parse_and_format_threads(small_run):
    """classify reads and tokenises the mappings file in pieces that begin on read boundaries and formats its per-line / per-read files in ranges of
    reads (round 4): one piece / range (MM_CLASSIFY_THREADS=1), three, and as many as the file allows (MM_CLASSIFY_THREADS=64 on a small file: pieces of a
    few reads, some of them empty) write the same bytes — also when the file has empty lines and no line break at its end"""
    o1, _, _ = _gpu_map(small_run, "cth", [])
    d = small_run["dir"]
    outs = {}
    for th in ("1", "3", "64"):
        x = str(d / f"cth_{th}")
        _copy_run(o1, x)
        if th != "1":                                             # the same mappings with oddities the pieces have to cope with
            pass
        _classify(x, small_run["db"].dir, [], {"MM_CLASSIFY_THREADS": th})
        outs[th] = {suf: open(x + suf).read() for suf in (".EM", ".EM.reads2Taxon", ".EM.reads2Taxon.krona", ".EM.WIMP", ".EM.lengthAndIdentitiesPerMappingUnit", ".EM.contigCoverage")}
    assert outs["1"] == outs["3"] == outs["64"] and len(outs["1"][".EM"]) > 10_000
    # empty lines between reads and a missing final line break: still the same reads
    txt = open(o1).read().rstrip("\n").split("\n")
    cut = [i for i in range(1, len(txt)) if txt[i].split(" ")[0] != txt[i - 1].split(" ")[0]]
    odd = []
    for i, ln in enumerate(txt):
        if i in cut[::5]:
            odd.append("")
        odd.append(ln)
    for th in ("1", "64"):
        x = str(d / f"cth_odd_{th}")
        _copy_run(o1, x)
        open(x, "w").write("\n".join(odd))
        _classify(x, small_run["db"].dir, [], {"MM_CLASSIFY_THREADS": th})
        for suf in (".EM.reads2Taxon", ".EM.WIMP", ".EM.reads2Taxon.krona"):
            assert open(x + suf).read() == outs["1"][suf], (suf, th)
        assert open(x + ".EM").read() == outs["1"][".EM"]

"""End to end through the drop-in CLI (metamaps_amd/csrc/metamaps, which uses only the C ABI) against the
oracle CLI on the same files: every output file of mapDirectly and classify.  Integers and text byte-exact;
mapping qualities / posteriors / frequencies within 1e-5 (BASELINE.json north_star)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "metamaps_amd", "csrc", "metamaps")


def _close(a: str, b: str, rel=1e-5, abs_=2e-6) -> bool:
    if a == b:
        return True
    x, y = float(a), float(b)
    return abs(x - y) <= abs_ + rel * max(abs(x), abs(y))


def _cmp_table(fa, fb, sep, numeric_cols):
    la, lb = open(fa).read().splitlines(), open(fb).read().splitlines()
    assert len(la) == len(lb), (fa, len(la), len(lb))
    for i, (x, y) in enumerate(zip(la, lb)):
        fx, fy = x.split(sep), y.split(sep)
        assert len(fx) == len(fy), (fa, i)
        for c, (u, v) in enumerate(zip(fx, fy)):
            if c in numeric_cols:
                assert _close(u, v), (fa, i, c, u, v)
            else:
                assert u == v, (fa, i, c, u, v)


@pytest.mark.parametrize("w_flag", [[], ["-w", "8"]])
def test_cli_outputs_match_oracle(oracle_lib, tmp_path, w_flag):
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=10, genome_len=60_000, seed=7)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=200, read_len=3000, seed=3)
    pa, pb = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    for exe, pre in ((CLI, pa), (orc.CLI, pb)):
        subprocess.run([exe, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", pre] + w_flag, check=True, capture_output=True, timeout=900)
        subprocess.run([exe, "classify", "--DB", db.dir, "--mappings", pre, "--minreads", "3"], check=True, capture_output=True, timeout=900)
    _cmp_table(pa, pb, " ", {13})
    for suf in (".meta", ".meta.unmappedReadsLengths"):
        assert open(pa + suf).read() == open(pb + suf).read(), suf
    pa_par = [l for l in open(pa + ".parameters") if not l.startswith("outFileName")]
    pb_par = [l for l in open(pb + ".parameters") if not l.startswith("outFileName")]
    assert pa_par == pb_par
    _cmp_table(pa + ".EM", pb + ".EM", " ", {13})
    assert open(pa + ".EM.reads2Taxon").read() == open(pb + ".EM.reads2Taxon").read()
    _cmp_table(pa + ".EM.reads2Taxon.krona", pb + ".EM.reads2Taxon.krona", "\t", {2})
    _cmp_table(pa + ".EM.WIMP", pb + ".EM.WIMP", "\t", {4, 5})
    _cmp_table(pa + ".EM.lengthAndIdentitiesPerMappingUnit", pb + ".EM.lengthAndIdentitiesPerMappingUnit", "\t", {3})
    _cmp_table(pa + ".EM.contigCoverage", pb + ".EM.contigCoverage", "\t", {6})
    assert sum(1 for _ in open(pa + ".EM.contigCoverage")) > 50
    assert sum(1 for _ in open(pa)) > 150
    _cmp_unknown_species(pa, pb)


CLASSIFY_SUFFIXES = (".EM", ".EM.reads2Taxon", ".EM.reads2Taxon.krona", ".EM.WIMP", ".EM.lengthAndIdentitiesPerMappingUnit", ".EM.contigCoverage", ".EM.evidenceUnknownSpecies")


@pytest.mark.parametrize("devices", [None, "0,0"])
def test_cli_then_classify_in_one_process_writes_the_same_files(tmp_path, devices):
    """`mapDirectly ... --then-classify DBDIR` (classify in the mapping process, on the files it has just written, with the live contexts) == `mapDirectly`
    followed by `classify` as two processes — every file byte for byte; two query files -> two prefixes, both classified; with two logical devices
    the reads are sharded over them and the shards' sums added on the host (--em-host-reduce: RCCL refuses two ranks on one physical device), as `classify --devices 0,0` does"""
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=10, genome_len=60_000, seed=7)
    r1 = synth.make_reads(db, str(tmp_path / "r1.fq"), n_reads=220, read_len=3000, seed=3)
    r2 = synth.make_reads(db, str(tmp_path / "r2.fq"), n_reads=90, read_len=2500, seed=4)
    q = r1["path"] + "," + r2["path"]
    dev = ["--devices", devices, "--em-host-reduce"] if devices else []
    two = [str(tmp_path / "two_a"), str(tmp_path / "two_b")]
    one = [str(tmp_path / "one_a"), str(tmp_path / "one_b")]
    subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", q, "-o", ",".join(two)] + dev, check=True, capture_output=True, timeout=900)
    subprocess.run([CLI, "classify", "--DB", db.dir, "--mappings", ",".join(two), "--minreads", "3"] + dev, check=True, capture_output=True, timeout=900)
    p = subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", q, "-o", ",".join(one), "--then-classify", db.dir, "--minreads", "3"] + dev,
                       capture_output=True, timeout=900, env=dict(os.environ, MM_CLI_TIMING="1"))
    assert p.returncode == 0, p.stderr.decode()[-1500:]
    assert "INFO, lap 9 classify" in p.stderr.decode()
    for a, b in zip(one, two):
        for suf in ("", ".meta", ".meta.unmappedReadsLengths") + CLASSIFY_SUFFIXES:
            assert open(a + suf, "rb").read() == open(b + suf, "rb").read(), suf
        assert os.path.getsize(a + ".EM.WIMP") > 200
    # the in-process classify takes the lines it formatted from memory (text + parsed fields per batch); in batches of 64 reads (several parts per file), with the text of a batch formatted by several threads (MM_CLI_FORMAT_PART)
    # and with the file read back instead (MM_CLI_CLASSIFY_FROM_FILE=1, what round 5 did) the files are the same again
    for tag, env in (("small_batches", {"MM_CLI_BATCH_READS": "64"}), ("from_file", {"MM_CLI_CLASSIFY_FROM_FILE": "1"}), ("format_parts", {"MM_CLI_FORMAT_PART": "40"})):
        alt = [str(tmp_path / f"{tag}_a"), str(tmp_path / f"{tag}_b")]
        p = subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", q, "-o", ",".join(alt), "--then-classify", db.dir, "--minreads", "3"] + dev,
                           capture_output=True, timeout=900, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        for a, b in zip(alt, two):
            for suf in ("", ".meta", ".meta.unmappedReadsLengths") + CLASSIFY_SUFFIXES:
                assert open(a + suf, "rb").read() == open(b + suf, "rb").read(), (tag, suf)


def test_cli_pooled_blocks_serve_the_workers(tmp_path):
    """mm_slab.hpp on the device: with the index-scale threshold lowered to 2 MiB (MM_INDEX_SCALE_MB) the buffers a 10 Mbp index build lets go
    of are pooled, and the worker contexts' allocations are cut out of them (MM_ALLOC_TRACE shows the pieces).  Files equal those of a run
    without slabs, byte for byte; several batches per worker, so pieces go back to their slabs and out again."""
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=20, genome_len=500_000, seed=11)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=600, read_len=5000, seed=5)
    outs = {}
    for name, extra in (("slabs", {}), ("plain", {"MM_NO_SLABS": "1"})):
        pre = str(tmp_path / name)
        env = dict(os.environ, MM_INDEX_SCALE_MB="2", MM_ALLOC_TRACE="1", MM_CLI_BATCH_READS="60", **extra)
        p = subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", pre, "--workers-per-gpu", "3"], capture_output=True, timeout=900, env=env)
        assert p.returncode == 0, p.stderr.decode()[-600:]
        err = p.stderr.decode()
        n_pieces = err.count("MM_ALLOC_TRACE slab piece")
        assert (n_pieces > 0) == (name == "slabs"), (name, n_pieces)
        if name == "slabs":
            assert "big block of" in err or "direct hipMalloc" in err     # the build did go through the index-scale path
        subprocess.run([CLI, "classify", "--DB", db.dir, "--mappings", pre, "--minreads", "3"], check=True, capture_output=True, timeout=900, env=env)
        outs[name] = {suf: open(pre + suf, "rb").read() for suf in ("", ".meta", ".EM", ".EM.WIMP", ".EM.reads2Taxon")}
    assert outs["slabs"] == outs["plain"]
    assert outs["slabs"][""].count(b"\n") > 400


def _cmp_unknown_species(pa, pb, expect_tests=True):
    # NA and integer columns as text, the six std::to_string doubles numerically
    suf = ".EM.evidenceUnknownSpecies"
    rows_a = [l.rstrip("\n").split("\t") for l in open(pa + suf)]
    rows_b = [l.rstrip("\n").split("\t") for l in open(pb + suf)]
    assert len(rows_a) == len(rows_b) > 1 and rows_a[0] == rows_b[0] and len(rows_a[0]) == 13
    for ra, rb in zip(rows_a[1:], rows_b[1:]):
        for c, (u, v) in enumerate(zip(ra, rb)):
            if c in (4, 5, 6, 9, 11, 12) and u != "NA" and v != "NA":
                assert abs(float(u) - float(v)) <= 2e-6, (ra, rb, c)
            else:
                assert u == v, (ra, rb, c)
    if expect_tests:                                             # the run exercised both tests, not only the NA branches
        assert any(r[6] != "NA" for r in rows_a[1:]) and any(r[12] not in ("NA", "1") for r in rows_a[1:])


def _run_pair(tmp_path, extra, n_reads=200, gpu_only=()):
    import json
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=10, genome_len=60_000, seed=7)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=n_reads, read_len=3000, seed=3)
    pa, pb = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    info = {}
    for exe, pre in ((CLI, pa), (orc.CLI, pb)):
        p = subprocess.run([exe, "mapDirectly", "-r", db.fasta, "-q", rd["path"], "-o", pre] + extra + (list(gpu_only) if exe == CLI else []),
                           check=True, capture_output=True, timeout=900)
        info[pre] = p
    n_gpu_chunks = sum(1 for l in info[pa].stdout.decode().splitlines() if l.startswith("INFO, index chunk"))
    n_cpu_chunks = json.loads(info[pb].stderr.decode().strip().splitlines()[-1])["chunks"]
    _cmp_table(pa, pb, " ", {13})
    for suf in (".meta", ".meta.unmappedReadsLengths"):
        assert open(pa + suf).read() == open(pb + suf).read(), suf
    pa_par = [l for l in open(pa + ".parameters") if not l.startswith("outFileName")]
    pb_par = [l for l in open(pb + ".parameters") if not l.startswith("outFileName")]
    assert pa_par == pb_par
    return pa, n_gpu_chunks, n_cpu_chunks


def test_cli_best_only_reporting(oracle_lib, tmp_path):
    """without --all only mappings within 1.0 of the read's best identity are reported (computeMap.hpp:546-587)"""
    pa, g, c = _run_pair(tmp_path, [])
    assert g == c == 1
    pall = str(tmp_path / "gpu_all")
    subprocess.run([CLI, "mapDirectly", "--all", "-r", str(tmp_path / "db" / "DB.fa"), "-q", str(tmp_path / "reads.fq"), "-o", pall],
                   check=True, capture_output=True, timeout=900)
    assert sum(1 for _ in open(pa)) < sum(1 for _ in open(pall))          # the filter removed something


@pytest.mark.parametrize("limit,all_flag", [(4_000_000, ["--all"]), (2_000_000, ["--all"]), (1_000_000, ["--all"]), (1_000_000, [])])
def test_cli_maxmemory_chunks(oracle_lib, tmp_path, limit, all_flag):
    """--maxmemory: same chunk boundaries as the reference's streaming rule (winSketch.hpp:274-329), per-chunk
    freqThreshold from the accumulated histogram, read-wise merge in chunk order, mapQ over the union."""
    pa, g, c = _run_pair(tmp_path, all_flag + ["--maxmemory-bytes", str(limit)])
    assert g == c and g >= 2, (g, c)


def test_cli_maxmemory_contig_too_large(tmp_path):
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=4, genome_len=60_000, seed=7)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=10, read_len=3000, seed=3)
    p = subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", str(tmp_path / "x"), "--maxmemory-bytes", "300000"],
                       capture_output=True, timeout=900)
    assert p.returncode != 0 and b"too large" in p.stderr


@pytest.mark.parametrize("full", [[], ["--full-index"]])
@pytest.mark.parametrize("extra", [[], ["--maxmemory-bytes", "1000000"]])
def test_cli_index_then_map_against_index(tmp_path, extra, full):
    """`index` + `mapAgainstIndex` give byte-identical output to `mapDirectly`, from the packed reference (index rebuilt) and from the
    persistent device index of `index --full-index` (IDX.N.mmidx, nothing rebuilt: SURVEY N2, mapWrap.h:358-405, :443-554), also with the
    chunk indexes of a --maxmemory run loaded one at a time (--stream-chunks)"""
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=10, genome_len=60_000, seed=7)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=200, read_len=3000, seed=3)
    direct, via = str(tmp_path / "direct"), str(tmp_path / "via")
    subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", direct] + extra, check=True, capture_output=True, timeout=900)
    p = subprocess.run([CLI, "index", "-r", db.fasta, "-i", str(tmp_path / "idx")] + extra + full, check=True, capture_output=True, timeout=900)
    stored = [l.split()[-1] for l in p.stdout.decode().splitlines() if l.startswith("Stored state in file")]
    assert len(stored) == (8 if extra else 1)
    assert all(f.endswith(".mmidx" if full else ".seqset") and os.path.getsize(f) > 0 for f in stored)
    assert open(str(tmp_path / "idx.index")).read().split()[0] == "1"
    for mode in ([], ["--stream-chunks"]) if extra else ([],):
        subprocess.run([CLI, "mapAgainstIndex", "--all", "-i", str(tmp_path / "idx"), "-q", rd["path"], "-o", via] + mode, check=True, capture_output=True, timeout=900)
        assert open(direct).read() == open(via).read()
        for suf in (".meta", ".meta.unmappedReadsLengths"):
            assert open(direct + suf).read() == open(via + suf).read(), suf
        a = [l for l in open(direct + ".parameters") if not l.startswith("outFileName")]
        b = [l for l in open(via + ".parameters") if not l.startswith("outFileName")]
        assert a == b
    if full:                                                       # a truncated index file is refused with a message, not mapped against
        data = open(stored[-1], "rb").read()
        open(stored[-1], "wb").write(data[: len(data) // 2])
        q = subprocess.run([CLI, "mapAgainstIndex", "--all", "-i", str(tmp_path / "idx"), "-q", rd["path"], "-o", via], capture_output=True, timeout=900)
        assert q.returncode != 0 and b"index file" in q.stderr
    # an incomplete index is refused (mapWrap.h:466-470)
    open(str(tmp_path / "idx.index"), "w").write("0\n")
    q = subprocess.run([CLI, "mapAgainstIndex", "--all", "-i", str(tmp_path / "idx"), "-q", rd["path"], "-o", via], capture_output=True, timeout=900)
    assert q.returncode != 0 and b"not complete" in q.stderr


def test_cli_midscale_matches_oracle(oracle_lib, tmp_path):
    """a larger case than the plumbing ones: 24 genomes x 1 Mbp (related pairs), 1500 reads of 10 kb with ONT-like errors,
    mapDirectly + classify against the oracle CLI"""
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=24, genome_len=1_000_000, seed=11, contigs_per_genome=3)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=1500, read_len=10_000, seed=5)
    pa, pb = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    for exe, pre, extra in ((CLI, pa, []), (orc.CLI, pb, ["-t", "16"])):
        subprocess.run([exe, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", pre] + extra, check=True, capture_output=True, timeout=1500)
        subprocess.run([exe, "classify", "--DB", db.dir, "--mappings", pre, "--minreads", "20"], check=True, capture_output=True, timeout=1500)
    _cmp_table(pa, pb, " ", {13})
    assert open(pa + ".meta").read() == open(pb + ".meta").read()
    _cmp_table(pa + ".EM", pb + ".EM", " ", {13})
    assert open(pa + ".EM.reads2Taxon").read() == open(pb + ".EM.reads2Taxon").read()
    _cmp_table(pa + ".EM.WIMP", pb + ".EM.WIMP", "\t", {4, 5})
    _cmp_unknown_species(pa, pb)
    assert sum(1 for _ in open(pa)) > 2000


_RANDOM_CASES = [
    # k, w, --pi, -m, read_len, len_jitter, sub, ins, del, --all
    (11, 5, 80, 500, 1500, 0.5, 0.03, 0.02, 0.03, True),
    (14, 3, 85, 1000, 2500, 0.3, 0.02, 0.02, 0.02, True),
    (19, 10, 75, 1000, 4000, 0.6, 0.05, 0.04, 0.05, False),
    (24, 16, 90, 2000, 6000, 0.2, 0.01, 0.01, 0.01, True),
    (16, 25, 80, 600, 3000, 0.8, 0.04, 0.03, 0.05, False),
    (12, 1, 82, 1000, 2000, 0.4, 0.03, 0.03, 0.03, True),
]


def _more_random_cases(n):
    """seeded random cases behind the hand-picked ones (MM_CLI_FUZZ_CASES=60 for a longer hunt; 60 ran clean in round 2)"""
    import random
    rng = random.Random(20260928)
    out = []
    for _ in range(n):
        k = rng.randint(8, 26); w = rng.choice([1, 2, 3, 5, 8, 11, 16, 24, 40])
        m = rng.choice([200, 500, 1000, 2500])
        out.append((k, w, rng.choice([70, 78, 80, 85, 92]), m, max(rng.choice([1200, 2500, 5000, 9000]), 2 * m),    # (no read long enough: classify of both CLIs refuses)
                    rng.choice([0.0, 0.3, 0.7]), rng.choice([0.0, 0.02, 0.05]), rng.choice([0.0, 0.02, 0.04]), rng.choice([0.0, 0.02, 0.05]), rng.random() < 0.6))
    return out


_RANDOM_CASES = _RANDOM_CASES + _more_random_cases(int(os.environ.get("MM_CLI_FUZZ_CASES", "4")))


@pytest.mark.parametrize("case", _RANDOM_CASES, ids=lambda c: f"k{c[0]}w{c[1]}pi{c[2]}m{c[3]}")
def test_cli_parameter_sweep_matches_oracle(oracle_lib, tmp_path, case):
    """mapDirectly + classify with k / w / identity threshold / minimum read length / read length spread / error rates
    away from the defaults (k above and below the 16-base fast path of K1, w = 1, sketches of very different sizes)"""
    import orc
    from metamaps_amd import synth
    k, w, pi, m, rl, jit, sub, ins, dele, all_ = case
    db = synth.make_db(str(tmp_path / "db"), n_genomes=8, genome_len=40_000, seed=100 + k)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=150, read_len=rl, seed=w, len_jitter=jit, sub=sub, ins=ins, dele=dele)
    pa, pb = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    opts = ["-k", str(k), "-w", str(w), "--pi", str(pi), "-m", str(m)] + (["--all"] if all_ else [])
    for exe, pre in ((CLI, pa), (orc.CLI, pb)):
        subprocess.run([exe, "mapDirectly", "-r", db.fasta, "-q", rd["path"], "-o", pre] + opts, check=True, capture_output=True, timeout=900)
        subprocess.run([exe, "classify", "--DB", db.dir, "--mappings", pre, "--minreads", "3"], check=True, capture_output=True, timeout=900)
    _cmp_table(pa, pb, " ", {13})
    for suf in (".meta", ".meta.unmappedReadsLengths"):
        assert open(pa + suf).read() == open(pb + suf).read(), suf
    assert [l for l in open(pa + ".parameters") if not l.startswith("outFileName")] == [l for l in open(pb + ".parameters") if not l.startswith("outFileName")]
    _cmp_table(pa + ".EM", pb + ".EM", " ", {13})
    assert open(pa + ".EM.reads2Taxon").read() == open(pb + ".EM.reads2Taxon").read()
    _cmp_table(pa + ".EM.WIMP", pb + ".EM.WIMP", "\t", {4, 5})
    _cmp_table(pa + ".EM.contigCoverage", pb + ".EM.contigCoverage", "\t", {6})
    _cmp_unknown_species(pa, pb, expect_tests=False)
    assert sum(1 for _ in open(pa)) > (30 if case in _RANDOM_CASES[:6] else 0)


@pytest.mark.parametrize("limit,range_bases,all_flag", [(1_000_000, 100_000, ["--all"]), (1_000_000, 30_000, []), (2_000_000, 10_000_000, ["--all"]), (4_000_000, 150_000, ["--all"])])
def test_cli_stream_chunks(oracle_lib, tmp_path, limit, range_bases, all_flag):
    """references whose chunk indexes do not fit HBM together (BASELINE config 5): --stream-chunks keeps one chunk index
    on the device at a time (all read batches are mapped against it, only the records are kept) and evaluates the chunk
    rule on contig ranges instead of an index of the whole reference; `--stream-range-bases` makes the ranges tiny here
    (smaller than a chunk, so that the range has to grow; several chunks per range; everything in one range).
    Same files as the oracle under the same --maxmemory, and as `index` + `mapAgainstIndex --stream-chunks`."""
    opts = all_flag + ["--maxmemory-bytes", str(limit)]
    pa, g, c = _run_pair(tmp_path, opts, gpu_only=["--stream-chunks", "--stream-range-bases", str(range_bases)])
    assert g == c and g >= 2, (g, c)
    db_fa, reads = str(tmp_path / "db" / "DB.fa"), str(tmp_path / "reads.fq")
    subprocess.run([CLI, "index", "-r", db_fa, "-i", str(tmp_path / "idx"), "--maxmemory-bytes", str(limit), "--stream-chunks", "--stream-range-bases", str(range_bases)],
                   check=True, capture_output=True, timeout=900)
    via = str(tmp_path / "via")
    subprocess.run([CLI, "mapAgainstIndex", "-i", str(tmp_path / "idx"), "-q", reads, "-o", via, "--stream-chunks"] + all_flag, check=True, capture_output=True, timeout=900)
    assert open(pa).read() == open(via).read()
    assert open(pa + ".meta").read() == open(via + ".meta").read()


def test_cli_stream_chunks_two_query_files(tmp_path):
    """several query files share each chunk pass (mapWrap.h:417-430): same output as the resident mode"""
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=10, genome_len=60_000, seed=7)
    r1 = synth.make_reads(db, str(tmp_path / "r1.fq"), n_reads=120, read_len=3000, seed=3)
    r2 = synth.make_reads(db, str(tmp_path / "r2.fq"), n_reads=80, read_len=2500, seed=4)
    outs = {}
    for tag, extra in (("res", []), ("str", ["--stream-chunks", "--stream-range-bases", "120000"])):
        o1, o2 = str(tmp_path / f"{tag}1"), str(tmp_path / f"{tag}2")
        subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", r1["path"] + "," + r2["path"], "-o", o1 + "," + o2, "--maxmemory-bytes", "1000000"] + extra,
                       check=True, capture_output=True, timeout=900)
        outs[tag] = (open(o1).read(), open(o2).read(), open(o1 + ".meta").read(), open(o2 + ".meta").read())
    assert outs["res"] == outs["str"] and len(outs["res"][0]) > 1000 and len(outs["res"][1]) > 1000


def test_cli_mixed_long_reads_match_oracle(oracle_lib, tmp_path):
    """BASELINE config 3's read shape at oracle scale: lengths spread over 1-60 kb with PacBio-like errors (2 % del, 8 % ins,
    2 % sub; simulate.pl:57), so that every sketch-size class of K5 (3 072 / 7 168 / 16 384 and beyond) and the multi-bin windows
    of the seed-hit filter meet the oracle directly, not only the GPU's own full slide"""
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=12, genome_len=400_000, seed=23, contigs_per_genome=2)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=260, read_len=9000, seed=9, len_jitter=1.0, sub=0.02, ins=0.08, dele=0.02)
    # (reads beyond ~145 kb are refused, test_long_read_limit_is_an_error: keep this set below 100 kb)
    recs = open(rd["path"]).read().split("\n")
    with open(rd["path"], "w") as f:
        for i in range(0, len(recs) - 3, 4):
            if len(recs[i + 1]) <= 100_000:
                f.write("\n".join(recs[i:i + 4]) + "\n")
    lens = [len(l) for i, l in enumerate(open(rd["path"])) if i % 4 == 1]
    assert 50_000 < max(lens) <= 100_000 and sum(1 for x in lens if x > 20_000) > 25 and min(lens) < 1500
    pa, pb = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    for exe, pre, extra in ((CLI, pa, []), (orc.CLI, pb, ["-t", "16"])):
        subprocess.run([exe, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", pre, "-w", "8"] + extra, check=True, capture_output=True, timeout=240)
        subprocess.run([exe, "classify", "--DB", db.dir, "--mappings", pre, "--minreads", "5"], check=True, capture_output=True, timeout=240)
    _cmp_table(pa, pb, " ", {13})
    assert open(pa + ".meta").read() == open(pb + ".meta").read()
    _cmp_table(pa + ".EM", pb + ".EM", " ", {13})
    assert open(pa + ".EM.reads2Taxon").read() == open(pb + ".EM.reads2Taxon").read()
    _cmp_table(pa + ".EM.WIMP", pb + ".EM.WIMP", "\t", {4, 5})
    assert sum(1 for _ in open(pa)) > 300


def test_longest_supported_reads_match_oracle(oracle_lib, tmp_path):
    """just below the limit: 120-140 kb reads (sketches of 26-31 thousand hashes, streams beyond the mask capacity of K5)"""
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=4, genome_len=400_000, seed=5, contigs_per_genome=1)
    seq = open(db.fasta).read().split("\n")
    genomes, cur = [], []
    for l in seq:
        if l.startswith(">"):
            if cur: genomes.append("".join(cur)); cur = []
        elif l: cur.append(l)
    if cur: genomes.append("".join(cur))
    import random
    rng = random.Random(3)
    with open(str(tmp_path / "long.fq"), "w") as f:
        for i, (g, L) in enumerate(((0, 140_000), (1, 120_000), (2, 131_000))):
            s0 = rng.randrange(0, len(genomes[g]) - L)
            r = list(genomes[g][s0:s0 + L])
            for j in range(0, L, 23): r[j] = "ACGT"[rng.randrange(4)]          # ~3 % substitutions
            f.write(f"@long{i}\n" + "".join(r) + "\n+\n" + "I" * L + "\n")
    pa, pb = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    for exe, pre in ((CLI, pa), (orc.CLI, pb)):
        subprocess.run([exe, "mapDirectly", "--all", "-r", db.fasta, "-q", str(tmp_path / "long.fq"), "-o", pre, "-w", "8"], check=True, capture_output=True, timeout=200)
    _cmp_table(pa, pb, " ", {13})
    assert sum(1 for _ in open(pa)) >= 3


def _genomes_of(fasta):
    genomes, cur = [], []
    for l in open(fasta).read().split("\n"):
        if l.startswith(">"):
            if cur: genomes.append("".join(cur)); cur = []
        elif l: cur.append(l)
    if cur: genomes.append("".join(cur))
    return genomes


@pytest.mark.parametrize("w_flag", [["-w", "8"], ["-w", "3"]])
def test_reads_of_any_length_match_oracle(oracle_lib, tmp_path, w_flag):
    """The reference sizes its sliding map from the read (computeMap.hpp:228-263, slidingMap.hpp:114-131): no length limit.
    Reads of 180 kb and 500 kb (sketches of 40 000 and 110 000 hashes at w = 8, beyond the LDS-resident K5 classes) map through the
    global-memory class exactly like the oracle, next to ordinary reads of the same batch; with w = 3 already the 60 kb read
    is in that class."""
    import orc, random
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=3, genome_len=700_000, seed=9, contigs_per_genome=1)
    genomes = [g for g in _genomes_of(db.fasta) if len(g) > 600_000]   # (make_db also writes a few-base contig)
    rng = random.Random(5)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    with open(str(tmp_path / "long.fq"), "w") as f:
        for i, (g, L, rc) in enumerate(((0, 180_000, False), (1, 8_000, False), (2, 500_000, True), (1, 60_000, False), (0, 33_000, True))):
            s0 = rng.randrange(0, len(genomes[g]) - L)
            r = list(genomes[g][s0:s0 + L])
            for j in range(0, L, 19): r[j] = "ACGT"[rng.randrange(4)]          # ~4 % substitutions
            if rc: r = [comp[c] for c in reversed(r)]
            f.write(f"@long{i}\n" + "".join(r) + "\n+\n" + "I" * L + "\n")
    pa, pb = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    for exe, pre in ((CLI, pa), (orc.CLI, pb)):
        # (MM_L2_GROUP_SORT_MIN=1: the workgroup lists of every K5 class, also the few groups of this batch, go through the ordering by candidate position)
        subprocess.run([exe, "mapDirectly", "--all", "-r", db.fasta, "-q", str(tmp_path / "long.fq"), "-o", pre] + w_flag, check=True, capture_output=True, timeout=900,
                       env=dict(os.environ, MM_L2_GROUP_SORT_MIN="1"))
    _cmp_table(pa, pb, " ", {13})
    for suf in (".meta", ".meta.unmappedReadsLengths"):
        assert open(pa + suf).read() == open(pb + suf).read(), suf
    mapped = {l.split(" ")[0] for l in open(pa)}
    assert mapped == {f"long{i}" for i in range(5)}


def test_cli_at_bench_hit_density_matches_oracle(oracle_lib, tmp_path):
    """mapDirectly + classify with k = 10 on a 24 Mbp reference: the 32-bit hash space is saturated as at miniSeq+H scale
    (tens of thousands of chance seed hits per read, the pre-filter drops > 90 %), every output file equal to the oracle's"""
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=24, genome_len=1_000_000, seed=33, contigs_per_genome=2)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=260, read_len=6000, seed=12, len_jitter=0.4)
    pa, pb = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    for exe, pre in ((CLI, pa), (orc.CLI, pb)):
        subprocess.run([exe, "mapDirectly", "--all", "-k", "10", "-w", "8", "-r", db.fasta, "-q", rd["path"], "-o", pre], check=True, capture_output=True, timeout=1800)
        subprocess.run([exe, "classify", "--DB", db.dir, "--mappings", pre, "--minreads", "3"], check=True, capture_output=True, timeout=900)
    _cmp_table(pa, pb, " ", {13})
    for suf in (".meta", ".meta.unmappedReadsLengths"):
        assert open(pa + suf).read() == open(pb + suf).read(), suf
    _cmp_table(pa + ".EM", pb + ".EM", " ", {13})
    assert open(pa + ".EM.reads2Taxon").read() == open(pb + ".EM.reads2Taxon").read()
    _cmp_table(pa + ".EM.WIMP", pb + ".EM.WIMP", "\t", {4, 5})
    _cmp_table(pa + ".EM.contigCoverage", pb + ".EM.contigCoverage", "\t", {6})
    assert sum(1 for _ in open(pa)) > 300


def _rewrite_fastq(src, dst, mode):
    """the reads of `src` in another layout: 'wrapped' (sequence and qualities over several lines, every third record), 'nasty'
    (quality lines that start with '@' or '>' or contain '+'), 'fasta' (no qualities, 70 columns)"""
    import random
    rng = random.Random(7)
    recs, lines = [], open(src).read().split("\n")
    for i in range(0, len(lines) - 3, 4):
        recs.append((lines[i], lines[i + 1]))
    with open(dst, "w") as f:
        for n, (h, s) in enumerate(recs):
            if mode == "fasta":
                f.write(">" + h[1:] + "\n" + "\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + "\n")
                continue
            q = "".join(rng.choice("@>+IIIIIIIIIIIIIIIII5?") for _ in s) if mode == "nasty" else "I" * len(s)
            if mode == "nasty" and n % 2 == 0:
                q = "@" + q[1:]
            if mode == "wrapped" and n % 3 == 0:
                cut = [0] + sorted(rng.sample(range(1, len(s)), 3)) + [len(s)]
                f.write(h + "\n" + "\n".join(s[a:b] for a, b in zip(cut[:-1], cut[1:])) + "\n+\n" + "\n".join(q[a:b] for a, b in zip(cut[:-1], cut[1:])) + "\n")
            else:
                f.write(h + "\n" + s + "\n+" + (h[1:] if n % 5 == 0 else "") + "\n" + q + "\n")


@pytest.mark.parametrize("mode", ["plain", "wrapped", "nasty", "fasta", "truncated"])
def test_cli_parallel_block_parser_equals_sequential_reader(tmp_path, mode):
    """query files are memory mapped and parsed in blocks by several threads; whatever the layout, the records must be those of the
    sequential kseq-like reader (blocks of 30 kB here, so that hundreds of block borders fall into records of every kind)"""
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=6, genome_len=60_000, seed=7)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=300, read_len=2500, seed=3)
    q = rd["path"]
    if mode in ("wrapped", "nasty", "fasta"):
        q = str(tmp_path / f"{mode}.fq"); _rewrite_fastq(rd["path"], q, mode)
    elif mode == "truncated":                                    # the file ends inside a quality string: kseq stops there (kseq.h:204)
        q = str(tmp_path / "trunc.fq")
        data = open(rd["path"], "rb").read()
        open(q, "wb").write(data[:len(data) * 2 // 3])
    outs = {}
    for tag, env in (("seq", {"MM_CLI_NO_MMAP": "1"}), ("one", {}), ("blocks", {"MM_CLI_BLOCK_BYTES": "30000", "MM_CLI_BATCH_READS": "17"})):
        pre = str(tmp_path / tag)
        p = subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", q, "-o", pre], capture_output=True, timeout=600, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        outs[tag] = (open(pre).read(), open(pre + ".meta").read(), open(pre + ".meta.unmappedReadsLengths").read())
    assert outs["seq"] == outs["one"] == outs["blocks"]
    assert len(outs["seq"][0]) > 10_000


def test_cli_config0_against_committed_golden(tmp_path):
    """BASELINE configs[0] through the GPU CLI against tests/golden/core_golden.json (hashes of the oracle CLI's files, committed):
    integer-only files byte for byte, mapping lines / WIMP with the float columns at 1e-5"""
    import hashlib, json
    from metamaps_amd import synth
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "core_golden.json")))["config0"]
    db = synth.make_db(str(tmp_path / "db"), n_genomes=10, genome_len=200_000, seed=7)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=1000, read_len=5000, seed=1)
    pre = str(tmp_path / "out")
    subprocess.run([CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", pre], check=True, capture_output=True, timeout=900)
    p = subprocess.run([CLI, "classify", "--DB", db.dir, "--mappings", pre, "--minreads", "3"], check=True, capture_output=True, timeout=900)
    sha = lambda suf: hashlib.sha256(open(pre + suf, "rb").read()).hexdigest()[:24]
    for suf in (".meta", ".meta.unmappedReadsLengths", ".EM.reads2Taxon"):
        assert sha(suf) == g["sha"][suf], suf
    assert sum(1 for _ in open(pre)) == g["n_lines"]
    for x, y in zip(open(pre).read().splitlines()[:12], g["first_lines"]):
        fx, fy = x.split(" "), y.split(" ")
        assert fx[:13] == fy[:13] and _close(fx[13], fy[13]), (x, y)
    par = dict(l.split(" ", 1) for l in open(pre + ".parameters").read().splitlines())
    for l in g["parameters"]:
        k_, v_ = l.split(" ", 1)
        assert par[k_] == v_, l
    wa, wb = open(pre + ".EM.WIMP").read().splitlines(), g["wimp"].splitlines()
    assert len(wa) == len(wb)
    for x, y in zip(wa, wb):
        fx, fy = x.split("\t"), y.split("\t")
        assert fx[:4] == fy[:4] and all(_close(u, v) for u, v in zip(fx[4:], fy[4:])), (x, y)
    import re
    lls = [float(x) for x in re.findall(r"Log likelihood: (\S+)", p.stdout.decode())]
    assert len(lls) == len(g["log_likelihood"]) and np.allclose(lls, g["log_likelihood"], rtol=1e-5)


def _two_line_fastq(path, recs, gz=False):
    import gzip
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for name, s in recs:
            f.write(b"@" + name.encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")


def test_cli_gz_query_and_gz_reference(oracle_lib, tmp_path):
    """.gz query and .gz reference through the GPU CLI (kseq over gzFile: winSketch.hpp:245-248, computeMap.hpp:121) == plain files == oracle;
    referenceSize is the size of the compressed file (commonFunc.hpp:211-231), so -w is given to keep the runs comparable"""
    import gzip, shutil
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=6, genome_len=50_000, seed=5)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=150, read_len=3000, seed=9)
    for src in (db.fasta, rd["path"]):
        with open(src, "rb") as f, gzip.open(src + ".gz", "wb") as g:
            shutil.copyfileobj(f, g)
    runs = {}
    for tag, exe, r, q in (("plain", CLI, db.fasta, rd["path"]), ("gzq", CLI, db.fasta, rd["path"] + ".gz"), ("gzrq", CLI, db.fasta + ".gz", rd["path"] + ".gz"),
                           ("cpu", orc.CLI, db.fasta + ".gz", rd["path"] + ".gz")):
        pre = str(tmp_path / tag)
        subprocess.run([exe, "mapDirectly", "--all", "-r", r, "-q", q, "-o", pre, "-w", "10"], check=True, capture_output=True, timeout=900)
        runs[tag] = pre
    assert sum(1 for _ in open(runs["plain"])) > 100
    for tag in ("gzq", "gzrq"):
        assert open(runs[tag]).read() == open(runs["plain"]).read()
        assert open(runs[tag] + ".meta").read() == open(runs["plain"] + ".meta").read()
    _cmp_table(runs["gzrq"], runs["cpu"], " ", {13})
    assert open(runs["gzrq"] + ".meta.unmappedReadsLengths").read() == open(runs["cpu"] + ".meta.unmappedReadsLengths").read()


def test_cli_duplicate_read_ids(oracle_lib, tmp_path):
    """mapWrap.h:71-75: a read ID seen before stops the run (exit 1) when the repeat carries mappings; a repeated ID on reads that
    do not map (or are too short) goes through.  GPU CLI and oracle CLI behave alike in both cases"""
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=4, genome_len=40_000, seed=3)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=30, read_len=2500, seed=2, frac_random=0.0, frac_short=0.0, with_oddities=False)
    recs = []
    with open(rd["path"], "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            recs.append((h[1:].split()[0].decode(), f.readline().strip())); f.readline(); f.readline()
    rng = np.random.default_rng(1)
    junk = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
    # (a) unmapped and too-short reads share IDs: fine
    ok = recs[:10] + [("dupU", junk(2000)), ("dupU", junk(2100)), ("dupS", junk(100)), ("dupS", junk(120))] + recs[10:20]
    _two_line_fastq(str(tmp_path / "ok.fq"), ok)
    # (b) a mapped read repeats the ID of an earlier mapped read: both programs stop with exit code 1
    bad = recs[:10] + [(recs[3][0], recs[12][1])] + recs[13:20]
    _two_line_fastq(str(tmp_path / "bad.fq"), bad)
    for exe, tag in ((CLI, "gpu"), (orc.CLI, "cpu")):
        p = subprocess.run([exe, "mapDirectly", "--all", "-r", db.fasta, "-q", str(tmp_path / "ok.fq"), "-o", str(tmp_path / (tag + "_ok")), "-w", "10"], capture_output=True, timeout=900)
        assert p.returncode == 0, (tag, p.stderr.decode()[-500:])
        p = subprocess.run([exe, "mapDirectly", "--all", "-r", db.fasta, "-q", str(tmp_path / "bad.fq"), "-o", str(tmp_path / (tag + "_bad")), "-w", "10"], capture_output=True, timeout=900)
        assert p.returncode == 1 and b"has already been processed" in p.stderr + p.stdout, (tag, p.returncode, p.stderr.decode()[-500:])
    _cmp_table(str(tmp_path / "gpu_ok"), str(tmp_path / "cpu_ok"), " ", {13})
    for suf in (".meta", ".meta.unmappedReadsLengths"):
        assert open(str(tmp_path / "gpu_ok") + suf).read() == open(str(tmp_path / "cpu_ok") + suf).read(), suf


@pytest.mark.parametrize("extra", [["--all"], ["--all", "--maxmemory-bytes", "1000000"], ["--all", "--maxmemory-bytes", "1000000", "--stream-chunks"]])
def test_cli_reference_streamed_in_groups(oracle_lib, tmp_path, monkeypatch, extra):
    """the reference reaches the device in groups of contigs (here ~100 kb each: a dozen groups, concatenated on the device), the index
    chunks are slices of the resident packed reference: same files as the oracle"""
    monkeypatch.setenv("MM_CLI_REF_GROUP_BASES", "100000")
    pa, g, c = _run_pair(tmp_path, [x for x in extra if x != "--stream-chunks"], gpu_only=[x for x in extra if x == "--stream-chunks"])
    assert g == c and (g >= 2 or len(extra) == 1)


@pytest.mark.parametrize("layout", ["wrapped", "hostile"])
def test_cli_reference_parsed_in_parallel_blocks(oracle_lib, tmp_path, monkeypatch, layout):
    """The reference FASTA is memory mapped and parsed in blocks by several threads (the chain-checked block parser of the query files:
    a block's records count once the block before it is seen to end exactly where this one starts), consumed in file order.  Same files as
    the sequential reader (MM_CLI_REF_SEQUENTIAL=1) and as the oracle (whose reader follows kseq.h character by character), for blocks of
    61 bytes, 5 kB and the default, on a reference rewritten with line widths from 1 to unwrapped, CRLF, blank lines, lower case,
    header comments that contain '>', '+' and '@'; "hostile": no line break at the end of the file."""
    import orc
    from metamaps_amd import synth
    db = synth.make_db(str(tmp_path / "db"), n_genomes=8, genome_len=50_000, seed=11)
    rd = synth.make_reads(db, str(tmp_path / "reads.fq"), n_reads=150, read_len=3000, seed=5)
    recs, name, seq = [], None, []
    for line in open(db.fasta, "rb"):
        if line.startswith(b">"):
            if name is not None: recs.append((name, b"".join(seq)))
            name, seq = line[1:].split()[0], []
        else:
            seq.append(line.strip())
    recs.append((name, b"".join(seq)))
    rng = np.random.default_rng(17)
    out = bytearray(b"\n\n")                                    # kseq skips everything before the first header character
    for i, (nm, sq) in enumerate(recs):
        width = [1, 7, 60, 61, 80, 1000, len(sq) + 1][i % 7]
        eol = b"\r\n" if i % 3 == 1 else b"\n"
        comment = [b"", b" plain comment", b" odd >comment +with @signs", b"\tTAB > + @"][i % 4]
        body = sq.lower() if i % 5 == 2 else sq
        out += b">" + nm + comment + eol
        for o in range(0, len(body), width):
            out += body[o:o + width] + eol
            if i % 4 == 3 and o == 0: out += eol                 # a blank line inside a record
    ref_path = str(tmp_path / "DB_rewritten.fa")
    if layout == "hostile":
        out = out.rstrip(b"\r\n")                                # no newline at the end of the file
    open(ref_path, "wb").write(bytes(out))
    import shutil
    for fn in os.listdir(os.path.dirname(db.fasta)):             # taxonomy files next to the reference
        if fn != os.path.basename(db.fasta) and os.path.isfile(os.path.join(os.path.dirname(db.fasta), fn)):
            shutil.copy(os.path.join(os.path.dirname(db.fasta), fn), str(tmp_path / fn))
    runs = {}
    for tag, env in (("seq", {"MM_CLI_REF_SEQUENTIAL": "1"}), ("b61", {"MM_CLI_REF_BLOCK_BYTES": "61"}), ("b5k", {"MM_CLI_REF_BLOCK_BYTES": "5000"}), ("default", {})):
        e = dict(os.environ); e.update(env)
        pre = str(tmp_path / ("gpu_" + tag))
        p = subprocess.run([CLI, "mapDirectly", "--all", "-r", ref_path, "-q", rd["path"], "-o", pre], capture_output=True, timeout=900, env=e)
        assert p.returncode == 0, (tag, p.stderr.decode()[-800:])
        runs[tag] = {suf: open(pre + suf, "rb").read() for suf in ("", ".meta", ".meta.unmappedReadsLengths")}
    for tag in ("b61", "b5k", "default"):
        assert runs[tag] == runs["seq"], tag
    assert len(runs["seq"][""]) > 10_000
    pb = str(tmp_path / "cpu")
    subprocess.run([orc.CLI, "mapDirectly", "--all", "-r", ref_path, "-q", rd["path"], "-o", pb], check=True, capture_output=True, timeout=900)
    _cmp_table(str(tmp_path / "gpu_default"), pb, " ", {13})
    assert open(str(tmp_path / "gpu_default") + ".meta").read() == open(pb + ".meta").read()


def test_cli_device_cap_decides_placement(tmp_path):
    """BASELINE config 5's decision — resident / spread over the devices / streamed — taken BY THE CLI, not by a flag: MM_DEVICE_BYTES_CAP (a test
    hook of the library's allocator: allocations beyond it fail, mm_ctx_device_info reports it) makes a 1 Gbp reference "larger than the
    device", so that `mapDirectly --maxmemory-bytes ...` must go to chunk streaming on one device and to sharding on three logical devices by
    itself — chunk rule on contig ranges, per-chunk thresholds from the accumulated histogram — with the allocator ENFORCING the cap while it
    does (a plan whose index builds or mapping buffers do not fit ends in MM_ERR_NOMEM).  Same files as the uncapped resident run.
    (1 Gbp and 8 GiB rather than something smaller: below that the fixed-size buffers of a context — K5 scratch slots, minimum table sizes —
    weigh more than the index, and the test would measure those.)"""
    from metamaps_amd import synth
    n_genomes = int(os.environ.get("MM_TEST_CAP_GENOMES", 100))
    db = synth.make_db(str(tmp_path / "db"), n_genomes=n_genomes, genome_len=10_000_000, seed=5)
    rd = synth.make_reads(db, str(tmp_path / "r.fq"), n_reads=3000, read_len=6000, seed=3)
    ref_bases = n_genomes * 10_000_000
    base = ["mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "--maxmemory-bytes", str(int(ref_bases * 3)), "--workers-per-gpu", "1"]
    # the CLI's size model (index_bytes, metamaps_main.cpp): 22 GB for the whole 1 Gbp reference (at this size nearly every hash is a list of one,
    # padded to a 64-byte sector), ~4 GB for each of the ~6 chunks of --maxmemory: 25 GB together do not fit 0.8 x 20 GiB, two per device do
    cap = int(os.environ.get("MM_TEST_DEVICE_CAP", 20 << 30))
    env = dict(os.environ, MM_DEVICE_BYTES_CAP=str(cap))
    env3 = dict(os.environ, MM_DEVICE_BYTES_CAP=str(3 * cap))     # three logical devices on the one physical device share its memory: the CLI gives each a third
    runs = {}
    # (the streamed runs with the index-scale threshold at 256 MiB, so that the chunk indexes' arrays and sort buffers go through the device's block
    # pool; once with the pool rescue — a refused request is served from pooled blocks — and once with the earlier give-everything-back path)
    env_s = dict(env, MM_INDEX_SCALE_MB="256", MM_ALLOC_TRACE="1")
    for tag, extra, e in (("resident", [], os.environ), ("auto_stream", [], env_s), ("auto_stream_no_rescue", [], dict(env_s, MM_NO_POOL_RESCUE="1")),
                          ("auto_shard", ["--devices", "0,0,0"], env3)):
        o = str(tmp_path / tag)
        p = subprocess.run([CLI] + base + ["-o", o] + extra, capture_output=True, timeout=900, env=dict(e))
        assert p.returncode == 0, (tag, p.stderr.decode()[-1500:], p.stdout.decode()[-1500:])
        runs[tag] = (open(o).read(), open(o + ".meta").read(), p.stdout.decode())
    assert "does not fit one device" not in runs["resident"][2]
    assert "chunk indexes are built and mapped one after the other" in runs["auto_stream"][2], runs["auto_stream"][2][-1500:]
    assert "the chunk indexes are spread over the devices" in runs["auto_shard"][2], runs["auto_shard"][2][-1500:]
    n_chunks = runs["resident"][2].count("INFO, index chunk ")
    assert n_chunks >= 4, runs["resident"][2][-1500:]
    assert len(runs["resident"][0]) > 100_000
    for tag in ("auto_stream", "auto_stream_no_rescue", "auto_shard"):
        assert runs[tag][0] == runs["resident"][0] and runs[tag][1] == runs["resident"][1], tag

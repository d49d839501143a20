"""Pins of A13-A15 (EM M step, cleanF, best mapping, taxonomy walk, every classify output file) by the reference's OWN output: the
example run in MetaMaps_example_output.zip (tests/golden/example/*, a real `metamaps classify` on miniSeq+H).  CPU only.

The example's `.EM` file carries, per mapping, the final posterior the reference computed (field 14) — not the mapping quality that
went into the EM (the reference overwrites that field, fEM.h:705), and the database (contigs without mappings, taxonomy dump) is not
part of the zip.  What can be rebuilt from the zip itself:
  * the contigs that carry mappings and their lengths (fields 6, 7) -> a taxonInfo.txt,
  * a taxonomy: every row of example.EM.WIMP names its level and taxon; the parent links are the NCBI lineage of the 16 species of the
    run (LINEAGE below; example.EM.evidenceUnknownSpecies confirms genome -> species -> genus for the 16 genomes with reads, the krona
    file the first non-x ancestor of the x-prefixed pseudo-taxa).  The per-level rows of the WIMP then check the links: a wrong parent
    gives a wrong sum.
With those,
  1. the oracle's OUTPUT WRITER (orc::write_classify_outputs = everything doEM does behind its loop) is fed with the example's
     posteriors (orc_finish_from_posteriors: f = one M step from them) and must reproduce the example's files: reads2Taxon, krona,
     lengthAndIdentities, and the WIMP — rows, order, names, every count column exactly; EMFrequency within 2e-3 (the reference stops
     its EM at a relative log-likelihood step of 1e-4, fEM.h:636, so its f is that close to the M step of its own posteriors and no
     closer: measured 1.33e-3); the 31 of 943 taxa that survive cleanF (fEM.h:1135-1163) exactly;
  2. the oracle's WHOLE classify (EM loop included) is run on mappings whose field 14 is w_i = p*_i / f^[t_i] * nLoc_i, f^ = M(p*): the
     E step only sees f[t] w_i / nLoc_i up to a per-read factor, so f^ is an exact fixed point of this problem with posteriors p*; the
     oracle's loop, STARTED AT f^, has to hold it — two iterations, unchanged log-likelihood, stop rule fires — and write the same
     files again.  (Started from uniform frequencies the problem does not retrace the reference's run: the zip lacks the frequencies of
     the 912 taxa cleanF removed, and w_i needs them to a relative precision one M step cannot give.)
Not pinned by any of this: nLoc itself (the zip has no contig without a mapping) and the mapping qualities that entered the EM."""
import collections
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EX = os.path.join(HERE, "golden", "example")

# NCBI lineage of the species of the example run: species -> (genus, family, order, phylum); all Bacteria (2)
LINEAGE = {
    "1063": ("1060", "31989", "204455", "1224"), "1280": ("1279", "90964", "1385", "1239"), "1282": ("1279", "90964", "1385", "1239"),
    "1299": ("1298", "183710", "118964", "1297"), "1311": ("1301", "1300", "186826", "1239"), "1351": ("1350", "81852", "186826", "1239"),
    "1396": ("1386", "186817", "1385", "1239"), "1520": ("1485", "31979", "186802", "1239"), "1596": ("1578", "33958", "186826", "1239"),
    "1639": ("1637", "186820", "1385", "1239"), "1747": ("1912216", "31957", "85009", "201174"), "210": ("209", "72293", "213849", "1224"),
    "287": ("286", "135621", "72274", "1224"), "487": ("482", "481", "206351", "1224"), "562": ("561", "543", "91347", "1224"),
    "821": ("816", "815", "171549", "976"),
}
# species of the genomes (definedGenomes rows) by the name the WIMP prints; the one renamed genus by hand
SPECIES_BY_NAME_PREFIX = {"Propionibacterium acnes": "1747"}
# first non-x ancestors the krona file shows for pseudo-taxa (a strain node between x1048 and its species)
EXTRA_NODES = {"243230": ("1299", "no rank", "Deinococcus radiodurans R1")}
X_PARENT = {"x1048": "243230"}


def _taxid(contig):
    return re.search(r"kraken:taxid\|(x?\d+)", contig).group(1)


def _example():
    em = [l.rstrip("\n").split(" ") for l in open(os.path.join(EX, "example.EM"))]
    wimp = [l.rstrip("\n").split("\t") for l in open(os.path.join(EX, "example.EM.WIMP"))]
    meta = dict(l.split() for l in open(os.path.join(EX, "example.meta")))
    return em, wimp, {k: int(v) for k, v in meta.items()}


def _build_db(dirname, em, wimp):
    """taxonInfo.txt + taxonomy/{names,nodes}.dmp from the example's own files"""
    os.makedirs(os.path.join(dirname, "taxonomy"), exist_ok=True)
    contigs = collections.OrderedDict()
    for f in em:
        contigs.setdefault(_taxid(f[5]), collections.OrderedDict())[f[5]] = int(f[6])
    with open(os.path.join(dirname, "taxonInfo.txt"), "w") as o:
        for t, cs in contigs.items():
            o.write(t + " " + ";".join(f"{c}={n}" for c, n in cs.items()) + "\n")
    name, rank = {"1": "root", "2": "Bacteria"}, {"1": "no rank", "2": "superkingdom"}
    parent = {"1": "1", "2": "1"}
    for r in wimp[1:]:
        if r[1] in ("0", "-3"):
            continue
        name[r[1]] = r[2]
        if r[0] != "definedGenomes":
            rank[r[1]] = r[0]
    sp_name = {name[s]: s for s in LINEAGE}
    for s, (g, fa, o_, ph) in LINEAGE.items():
        parent[s], parent[g], parent[fa], parent[o_], parent[ph] = g, fa, o_, ph, "2"
    for n, (p, rk, nm) in EXTRA_NODES.items():
        parent[n], rank[n], name[n] = p, rk, nm
    genomes = [r[1] for r in wimp[1:] if r[0] == "definedGenomes" and r[1] not in ("0", "-3")]
    for g in genomes:
        hit = [s for nm, s in sp_name.items() if name[g].startswith(nm)] + [s for pre, s in SPECIES_BY_NAME_PREFIX.items() if name[g].startswith(pre)]
        assert len(hit) == 1, (g, name[g], hit)
        parent[g] = X_PARENT.get(g, hit[0]); rank[g] = "no rank"
    for t in contigs:                                             # the 912 taxa that cleanF removes: known to the taxonomy, nothing more
        if t not in parent:
            parent[t], rank[t], name[t] = "2", "no rank", "taxon " + t
    with open(os.path.join(dirname, "taxonomy", "nodes.dmp"), "w") as o:
        for n in parent:
            o.write(f"{n}\t|\t{parent[n]}\t|\t{rank[n]}\t|\n")
    with open(os.path.join(dirname, "taxonomy", "names.dmp"), "w") as o:
        for n in parent:
            o.write(f"{n}\t|\t{name[n]}\t|\t\t|\tscientific name\t|\n")
    return contigs


def _check_against_example(prefix, wimp, f_tol):
    ex = lambda suf: open(os.path.join(EX, "example" + suf)).read()
    got = lambda suf: open(prefix + suf).read()
    assert got(".EM.reads2Taxon") == ex(".EM.reads2Taxon")
    assert got(".EM.lengthAndIdentitiesPerMappingUnit") == ex(".EM.lengthAndIdentitiesPerMappingUnit")
    ka, kb = [l.split("\t") for l in got(".EM.reads2Taxon.krona").splitlines()], [l.split("\t") for l in ex(".EM.reads2Taxon.krona").splitlines()]
    assert len(ka) == len(kb) == 78
    for a, b in zip(ka, kb):
        assert a[:2] == b[:2] and abs(float(a[2]) - float(b[2])) <= max(6e-7, 3 * f_tol * float(b[2])), (a, b)   # (first non-x taxon; posterior of the best mapping)
    mine = [l.rstrip("\n").split("\t") for l in open(prefix + ".EM.WIMP")]
    assert len(mine) == len(wimp) and mine[0] == wimp[0]
    worst = 0.0
    for a, b in zip(mine[1:], wimp[1:]):
        assert a[:4] == b[:4], (a, b)                             # level, taxon, name, Absolute — rows in the reference's order
        tol = f_tol if a[0] == "definedGenomes" else 2.5 * f_tol  # (a family / order / phylum row adds up the deviations of its genomes)
        for c in (4, 5):
            worst = max(worst, abs(float(a[c]) - float(b[c])))
            assert abs(float(a[c]) - float(b[c])) <= tol, (a, b)
    return worst


def test_output_writer_reproduces_the_example_files(oracle_lib, tmp_path):
    em, wimp, meta = _example()
    assert meta["ReadsMapped"] == 73 and len(em) == 2985
    db = str(tmp_path / "db")
    contigs = _build_db(db, em, wimp)
    assert len(contigs) == 943
    out = str(tmp_path / "out")
    rc = oracle_lib.L.orc_finish_from_posteriors(os.path.join(EX, "example.EM").encode(), os.path.join(EX, "example").encode(), db.encode(), out.encode())
    assert rc == 0
    worst = _check_against_example(out, wimp, 2e-3)
    assert worst > 1e-5                                           # (the reference's f is NOT the M step of its final posteriors to print precision: the tolerance is needed)
    # the recalibrated mappings come back as they went in (std::to_string of the parsed text)
    assert open(out + ".EM").read() == open(os.path.join(EX, "example.EM")).read()
    kept = [r[1] for r in wimp[1:] if r[0] == "definedGenomes" and r[1] not in ("0", "-3")]
    assert len(kept) == 31                                        # cleanF: 31 of 943 taxa, the same ones


def test_em_loop_holds_the_example_as_a_fixed_point(oracle_lib, tmp_path):
    """the oracle's E step / M step / stop rule on mappings whose field 14 is w_i = p*_i / f^[t_i] * nLoc_i (times a per-read constant),
    f^ = M(p*): the E step sees f[t] w_i / nLoc_i, so at f = f^ the posteriors are the example's p* and the M step returns f^ — the loop,
    started there, must stop after its second iteration (the stop rule needs two) with an unchanged log-likelihood and write the example's
    files.  nLoc here is this file's own restatement of fEM.h:325-348 over the synthesised taxonInfo: the oracle's has to agree with it for
    the fixed point to hold.  (Started from uniform frequencies the same problem does NOT retrace the reference's run: the zip does not
    carry the frequencies of the 912 taxa cleanF removed, and w_i needs them to a relative precision the one M step cannot give.)"""
    import ctypes as C
    em, wimp, meta = _example()
    db = str(tmp_path / "db")
    contigs = _build_db(db, em, wimp)
    s = collections.defaultdict(float)
    for f in em:
        s[_taxid(f[5])] += float(f[13])
    tot = sum(s.values())
    fhat = {t: v / tot for t, v in s.items()}
    reads = collections.OrderedDict()
    for f in em:
        reads.setdefault(f[0], []).append(f)
    pre = str(tmp_path / "run")
    with open(pre, "w") as o:
        for rid, lines in reads.items():
            rl = int(lines[0][1])
            hit = {f[5] for f in lines}
            nloc_of = lambda t: sum((n - rl + 1) if n >= rl else (1 if c in hit else 0) for c, n in contigs[t].items())
            scale = 1.0 / nloc_of(_taxid(lines[0][5]))            # (per read: log-likelihoods of the magnitude of a real run)
            for f in lines:
                t = _taxid(f[5])
                g = list(f)
                g[13] = repr(float(f[13]) / fhat[t] * nloc_of(t) * scale) if float(f[13]) > 0 else "0"   # (a posterior printed as 0.000000 carries no weight)
                o.write(" ".join(g) + "\n")
    for suf in (".meta", ".meta.unmappedReadsLengths"):
        open(pre + suf, "w").write(open(os.path.join(EX, "example" + suf)).read())
    f0 = str(tmp_path / "f0")
    with open(f0, "w") as o:
        for t, v in fhat.items():
            o.write(f"{t} {v!r}\n")
    ll = (C.c_double * 16)()
    n_it = oracle_lib.L.orc_classify_from(pre.encode(), db.encode(), f0.encode(), ll, 16)
    assert n_it == 2 and abs(ll[1] - ll[0]) < 1e-9 * abs(ll[0]) and -2000 < ll[0] < -500
    _check_against_example(pre, wimp, 2e-3)
    mine = [l.rstrip("\n").split(" ") for l in open(pre + ".EM")]
    assert len(mine) == len(em)
    assert all(a[:13] == b[:13] for a, b in zip(mine, em))
    assert max(abs(float(a[13]) - float(b[13])) for a, b in zip(mine, em)) <= 4e-6      # (%f text of posteriors whose printed values sum to 1 +- a few 1e-6 per read)

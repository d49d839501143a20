"""Deterministic synthetic MetaMaps inputs (small, host-side; numpy only).

Produces what the reference's `buildDB.pl` would have left in a database directory
(formats: SURVEY.md Appendix B; reference parsers fEM.h:1341-1358, taxonomy.h:137-246,
fEM.h:1421-1473) plus a FASTQ of error-laden long reads, so that `mapDirectly` and
`classify` can run end to end without any real data.  The large, miniSeq+H-scale
generator used by bench.py runs on the GPU (csrc/mm_synth.hip); this module is for the
parity tests and the reference-CPU-runnable configuration (BASELINE.json configs[0]).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = np.arange(256, dtype=np.uint8)
for a, b in (b"AT", b"TA", b"CG", b"GC"):
    _COMP[a] = b


def random_genome(rng: np.random.Generator, n: int) -> np.ndarray:
    return _ACGT[rng.integers(0, 4, n)]


def mutate(rng: np.random.Generator, g: np.ndarray, sub: float = 0.0, ins: float = 0.0, dele: float = 0.0) -> np.ndarray:
    """i.i.d. substitutions / insertions / deletions (ASCII uint8 in, ASCII uint8 out)."""
    n = g.size
    out = g.copy()
    if sub > 0:
        m = rng.random(n) < sub
        # substitute by a *different* base
        shift = rng.integers(1, 4, int(m.sum()))
        code = np.zeros(256, dtype=np.int64)
        code[_ACGT] = np.arange(4)
        out[m] = _ACGT[(code[out[m]] + shift) % 4]
    if dele > 0:
        keep = rng.random(n) >= dele
    else:
        keep = np.ones(n, dtype=bool)
    if ins > 0:
        nins = rng.random(n) < ins
        rep = 1 + nins.astype(np.int64)
        rep[~keep] = nins[~keep]  # deleted base may still be followed by an insertion
        pos = np.repeat(np.arange(n), rep)
        res = out[pos]
        # the second copy of a repeated position is the inserted random base
        first = np.ones(pos.size, dtype=bool)
        first[1:] = pos[1:] != pos[:-1]
        ins_mask = ~first | (~keep[pos] & first)
        res[ins_mask] = _ACGT[rng.integers(0, 4, int(ins_mask.sum()))]
        return res
    return out[keep]


def revcomp(s: np.ndarray) -> np.ndarray:
    return _COMP[s[::-1]]


@dataclass
class SynthDB:
    dir: str
    fasta: str
    contig_ids: list
    contig_taxon: list
    contig_seqs: list          # ASCII uint8 arrays, in DB.fa order
    genome_taxon: list
    genome_contigs: list       # per genome: indices into contig_* lists


def _wrap(seq: bytes, width: int = 80) -> bytes:
    return b"\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + b"\n"


def make_db(out_dir: str, n_genomes: int = 10, genome_len: int = 200_000, seed: int = 7,
            pair_divergence: float = 0.03, contigs_per_genome: int = 2, with_oddities: bool = True) -> SynthDB:
    """DB-mini of SURVEY.md §8 D1: genome 2i+1 = genome 2i with `pair_divergence` substitutions."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(out_dir, "taxonomy"), exist_ok=True)
    genomes = []
    for g in range(n_genomes):
        if g % 2 == 0:
            genomes.append(random_genome(rng, genome_len))
        else:
            genomes.append(mutate(rng, genomes[-1], sub=pair_divergence))
    if with_oddities and n_genomes >= 3:
        g = genomes[2]
        g[5000:5600] = ord("N")                       # an N run (hashed as-is, commonFunc.hpp:44-51)
        g[9000:9400] = np.frombuffer(g[9000:9400].tobytes().lower(), dtype=np.uint8)   # lower case, upper-cased by the mapper
        g[12000:12300] = np.frombuffer(b"ACGT" * 75, dtype=np.uint8)   # tandem repeat / palindromic k-mers
        g[15000:15200] = ord("A")                     # homopolymer: the same hash at every position
    # taxonomy: 1 root; 2 superkingdom; 100 phylum; 200 order; 300+ family; 500+ genus; 1000+ species; genomes below
    nodes = {"1": ("1", "no rank", "root"), "2": ("1", "superkingdom", "Bacteria"),
             "100": ("2", "phylum", "Synthphyla"), "200": ("100", "order", "Synthales")}
    genome_taxon = []
    for g in range(n_genomes):
        fam, gen, sp = str(300 + g // 8), str(500 + g // 4), str(1000 + g // 2)
        nodes.setdefault(fam, ("200", "family", f"Synthaceae{g // 8}"))
        nodes.setdefault(gen, (fam, "genus", f"Synthus{g // 4}"))
        nodes.setdefault(sp, (gen, "species", f"Synthus{g // 4} species{g // 2}"))
        tid = f"x{7000 + g}" if g % 2 == 1 else str(10000 + g)   # pseudo-IDs as buildDB.pl mints them
        nodes[tid] = (sp, "no rank", f"Synthus{g // 4} species{g // 2} strain{g}")
        genome_taxon.append(tid)
    with open(os.path.join(out_dir, "taxonomy", "nodes.dmp"), "w") as f:
        for tid, (par, rank, _) in nodes.items():
            f.write(f"{tid}\t|\t{par}\t|\t{rank}\t|\n")
    with open(os.path.join(out_dir, "taxonomy", "names.dmp"), "w") as f:
        for tid, (_, _, name) in nodes.items():
            f.write(f"{tid}\t|\t{name}\t|\t\t|\tscientific name\t|\n")
    open(os.path.join(out_dir, "taxonomy", "merged.dmp"), "w").close()
    # contigs
    contig_ids, contig_taxon, contig_seqs, genome_contigs = [], [], [], [[] for _ in range(n_genomes)]
    pieces = []
    for g, seq in enumerate(genomes):
        cuts = sorted(rng.choice(np.arange(1000, seq.size - 1000), contigs_per_genome - 1, replace=False).tolist()) \
            if contigs_per_genome > 1 else []
        bounds = [0] + cuts + [seq.size]
        for a, b in zip(bounds[:-1], bounds[1:]):
            pieces.append((g, seq[a:b]))
    if with_oddities:
        pieces.append((0, np.frombuffer(b"ACGTAC", dtype=np.uint8)))   # shorter than k: metadata only (winSketch.hpp:258)
    order = rng.permutation(len(pieces))
    for ci, pi in enumerate(order):
        g, s = pieces[pi]
        cid = f"C{ci}|kraken:taxid|{genome_taxon[g]}|ACC{pi:04d}.1"
        contig_ids.append(cid); contig_taxon.append(genome_taxon[g]); contig_seqs.append(s)
        genome_contigs[g].append(ci)
    fasta = os.path.join(out_dir, "DB.fa")
    with open(fasta, "wb") as f:
        for cid, s in zip(contig_ids, contig_seqs):
            f.write(b">" + cid.encode() + b" synthetic\n" + _wrap(s.tobytes()))
    with open(os.path.join(out_dir, "taxonInfo.txt"), "w") as f:
        for g in range(n_genomes):
            f.write(genome_taxon[g] + " " + ";".join(f"{contig_ids[c]}={contig_seqs[c].size}" for c in genome_contigs[g]) + "\n")
    with open(os.path.join(out_dir, "contigNstats_windowSize_1000.txt"), "w") as f:
        for cid, tid, s in zip(contig_ids, contig_taxon, contig_seqs):
            nwin = -(-s.size // 1000)
            ns = [int(((s[i * 1000:(i + 1) * 1000] == ord("N")) | (s[i * 1000:(i + 1) * 1000] == ord("n"))).sum()) for i in range(nwin)]
            f.write(f"{tid}\t{cid}\t" + ";".join(map(str, ns)) + "\n")
    return SynthDB(out_dir, fasta, contig_ids, contig_taxon, contig_seqs, genome_taxon, genome_contigs)


def make_reads(db: SynthDB, path: str, n_reads: int = 1000, read_len: int = 5000, seed: int = 1,
               sub: float = 0.04, ins: float = 0.03, dele: float = 0.05, frac_random: float = 0.05,
               frac_short: float = 0.02, len_jitter: float = 0.0, abundance_sigma: float = 1.5,
               with_oddities: bool = True) -> dict:
    """ONT-like reads (defaults ≈ 88 % identity, cf. simulate.pl:57) sampled from the DB contigs."""
    rng = np.random.default_rng(seed)
    ng = len(db.genome_taxon)
    ab = rng.lognormal(0.0, abundance_sigma, ng)
    ab /= ab.sum()
    truth = []
    with open(path, "wb") as f:
        for r in range(n_reads):
            L = read_len if len_jitter == 0 else int(read_len * np.exp(rng.normal(0, len_jitter)))
            u = rng.random()
            if u < frac_short:
                L = int(rng.integers(20, 900)); src = "short"
            if u >= frac_short and u < frac_short + frac_random:
                s = random_genome(rng, L); src = "random"
            else:
                g = int(rng.choice(ng, p=ab))
                cands = [c for c in db.genome_contigs[g] if db.contig_seqs[c].size > L + 10]
                c = cands[int(rng.integers(len(cands)))] if cands else max(db.genome_contigs[g], key=lambda c: db.contig_seqs[c].size)
                cs = db.contig_seqs[c]
                L0 = min(L, cs.size)
                st = int(rng.integers(0, cs.size - L0 + 1))
                s = np.frombuffer(cs[st:st + L0].tobytes().upper(), dtype=np.uint8).copy()
                if rng.random() < 0.5:
                    s = revcomp(s)
                s = mutate(rng, s, sub, ins, dele)
                src = db.genome_taxon[g] if u >= frac_short else "short"
            if with_oddities and r % 97 == 13 and s.size > 400:
                s = s.copy(); s[100:140] = ord("N")
            if with_oddities and r % 101 == 7 and s.size > 400:
                s = np.frombuffer(s.tobytes().lower(), dtype=np.uint8)
            name = f"read{r:06d}/{src}"
            truth.append((name, src, int(s.size)))
            f.write(b"@" + name.encode() + b" len=" + str(s.size).encode() + b"\n" + s.tobytes() + b"\n+\n" + b"I" * s.size + b"\n")
    return {"path": path, "truth": truth}


def write_db_dir(out_dir: str, contigs: list) -> dict:
    """A database directory (DB.fa, taxonInfo.txt, taxonomy/*.dmp, contigNstats_windowSize_1000.txt) around given sequences:
    contigs = [(genome id, ASCII sequence bytes)] in DB.fa order; genome g becomes taxon 1000000 + g under species 500000 + g // 4,
    genus 100000 + g // 16 (bench.py's CPU / CLI sample of the device-generated reference)."""
    os.makedirs(os.path.join(out_dir, "taxonomy"), exist_ok=True)
    nodes = {"1": ("1", "no rank", "root"), "2": ("1", "superkingdom", "Bacteria"), "100": ("2", "phylum", "Synthphyla"), "200": ("100", "order", "Synthales"),
             "300": ("200", "family", "Synthaceae")}
    per_taxon: dict = {}
    fasta = os.path.join(out_dir, "DB.fa")
    with open(fasta, "wb") as f, open(os.path.join(out_dir, "contigNstats_windowSize_1000.txt"), "w") as ns:
        for ci, (g, seq) in enumerate(contigs):
            tid, sp, ge = str(1000000 + g), str(500000 + g // 4), str(100000 + g // 16)
            nodes.setdefault(ge, ("300", "genus", f"Synthus{g // 16}"))
            nodes.setdefault(sp, (ge, "species", f"Synthus{g // 16} species{g // 4}"))
            nodes.setdefault(tid, (sp, "no rank", f"Synthus{g // 16} species{g // 4} strain{g}"))
            cid = f"C{ci}|kraken:taxid|{tid}|SYN{ci:05d}.1"
            f.write(b">" + cid.encode() + b"\n" + seq + b"\n")
            per_taxon.setdefault(tid, []).append(f"{cid}={len(seq)}")
            a = np.frombuffer(seq, dtype=np.uint8)
            isn = ((a == ord("N")) | (a == ord("n"))).astype(np.int64)
            nwin = -(-len(seq) // 1000)
            counts = np.add.reduceat(isn, np.arange(0, len(seq), 1000)) if len(seq) else np.zeros(0, dtype=np.int64)
            ns.write(f"{tid}\t{cid}\t" + ";".join(map(str, counts[:nwin].tolist())) + "\n")
    with open(os.path.join(out_dir, "taxonInfo.txt"), "w") as f:
        for tid, lst in per_taxon.items():
            f.write(tid + " " + ";".join(lst) + "\n")
    with open(os.path.join(out_dir, "taxonomy", "nodes.dmp"), "w") as f:
        for tid, (par, rank, _) in nodes.items():
            f.write(f"{tid}\t|\t{par}\t|\t{rank}\t|\n")
    with open(os.path.join(out_dir, "taxonomy", "names.dmp"), "w") as f:
        for tid, (_, _, name) in nodes.items():
            f.write(f"{tid}\t|\t{name}\t|\t\t|\tscientific name\t|\n")
    open(os.path.join(out_dir, "taxonomy", "merged.dmp"), "w").close()
    return {"dir": out_dir, "fasta": fasta}

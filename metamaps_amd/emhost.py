"""Host side of `classify` above the C ABI, in Python (used by bench.py and the tests; the C++ CLI in
csrc/host has the same logic).  Mirrors meta::doEM's data preparation and loop control:

  getTaxonIDsFromMappingsFile   fEM.h:1366   taxa seen in the mappings, std::set (lexicographic) order
  loadRelevantTaxonInfo         fEM.h:1320   taxon -> {contig -> length}
  getMappingLocations           fEM.h:234    per (read, taxon): number of possible mapping locations
  doEM loop                     fEM.h:501    f <- normalised sums; stop when ll gain <= 1 and relative < 1e-4

TEST-SIDE code: the product (libmetamaps_hip.so, the `metamaps` CLI) never imports this module; its `classify` is C++ (csrc/host/metamaps_main.cpp, ClassifyRun).
"""
from __future__ import annotations

import re
from dataclasses import dataclass

import numpy as np

_TAXON_RE = re.compile(r"kraken:taxid\|(x?\d+)")
_DBL_MIN = 2.2250738585072014e-308


def extract_taxon(contig: str) -> str:
    m = _TAXON_RE.search(contig)
    if not m:
        raise ValueError(f"Could not extract taxon ID from contig identifier '{contig}'")
    return m.group(1)


def load_taxon_info(db_dir: str) -> dict:
    info: dict = {}
    with open(f"{db_dir}/taxonInfo.txt") as f:
        for ln in f:
            ln = ln.rstrip("\n")
            if not ln:
                continue
            tid, rest = ln.split(" ")
            d = info.setdefault(tid, {})
            for c in rest.split(";"):
                name, length = c.split("=")
                d[name] = int(length)
    return info


@dataclass
class EMProblem:
    taxa: list            # taxon ids, index = taxon number on the device
    read_ids: list
    read_off: np.ndarray  # [n_reads+1]
    taxon: np.ndarray     # [n_entries] index into taxa
    mapq: np.ndarray      # [n_entries] as parsed from the 6-digit text
    inv_nloc: np.ndarray  # [n_entries] 1/(double)nLoc(read, taxon)
    lines: list           # the mapping lines, grouped like the device arrays


def nloc_for(read_len: int, contig_lengths: dict, contigs_hit: set) -> int:
    """fEM.h:325-348"""
    n = 0
    for cid, clen in contig_lengths.items():
        if clen >= read_len:
            n += clen - read_len + 1
        elif cid in contigs_hit:
            n += 1
    return n


def load_problem(mapped: str, db_dir: str) -> EMProblem:
    tinfo = load_taxon_info(db_dir)
    groups, cur, cur_id = [], [], None
    with open(mapped) as f:
        for ln in f:
            ln = ln.rstrip("\n")
            if not ln:
                continue
            rid = ln[:ln.index(" ")]
            if rid != cur_id:
                if cur:
                    groups.append(cur)
                cur, cur_id = [], rid
            cur.append(ln)
    if cur:
        groups.append(cur)
    taxa = sorted({extract_taxon(ln.split(" ")[5]) for g in groups for ln in g})
    tindex = {t: i for i, t in enumerate(taxa)}
    read_off = [0]
    taxon, mapq, inv = [], [], []
    for g in groups:
        flds = [ln.split(" ") for ln in g]
        rlen = int(flds[0][1])
        tx = [extract_taxon(f[5]) for f in flds]
        hit = {f[5] for f in flds}
        per_taxon = {t: nloc_for(rlen, tinfo[t], hit) for t in set(tx)}
        for f, t in zip(flds, tx):
            v = float(f[13])
            if 0 < v < _DBL_MIN:
                v = 0.0                   # std::stod out_of_range on a denormal → 0, fEM.h:269-275
            taxon.append(tindex[t]); mapq.append(v); inv.append(1.0 / float(per_taxon[t]))
        read_off.append(len(taxon))
    return EMProblem(taxa, [g[0][:g[0].index(" ")] for g in groups], np.array(read_off, dtype=np.int64),
                     np.array(taxon, dtype=np.int32), np.array(mapq, dtype=np.float64), np.array(inv, dtype=np.float64), groups)


def run_em(step, n_taxa: int, max_iter: int = 10_000):
    """step(f) -> (f_next already normalised over ALL ranks, log-likelihood).  fEM.h:491-661"""
    f = np.full(n_taxa, 1.0 / n_taxa, dtype=np.float64)
    lls = []
    ll_prev = 0.0
    for it in range(max_iter):
        f_next, ll = step(f)
        lls.append(ll)
        stop = False
        if it > 0:
            diff = ll - ll_prev
            rel = 1 - ll / ll_prev
            stop = diff <= 1 and rel < 0.0001
        f = f_next
        ll_prev = ll
        if stop:
            break
    return f, lls


def parse6(v: np.ndarray) -> np.ndarray:
    """The double that std::stod returns for the 6-significant-digit text of v (mapWrap.h:318 → fEM.h:265)."""
    out = np.zeros_like(v)
    nz = v > 0
    a = v[nz]
    e = np.floor(np.log10(a)).astype(np.int64)
    pe = np.power(10.0, e.astype(np.float64))
    e = np.where(a < pe, e - 1, np.where(a >= pe * 10, e + 1, e))
    t = 5 - e
    x = a * np.power(10.0, t.astype(np.float64))
    d = np.rint(x)
    bump = d >= 1e6
    d = np.where(bump, d / 10, d)
    t = np.where(bump, t - 1, t)
    r = d / np.power(10.0, t.astype(np.float64))
    r[r < _DBL_MIN] = 0.0
    out[nz] = r
    return out

#!/usr/bin/env python3
"""Regenerates tests/golden/core_golden.json: regression pins of the ORACLE's integer core, so that an edit of oracle/ cannot move
the checker and the checked side together unnoticed (the survey validated these pieces against
a scratch build of the reference that cannot be rebuilt here — Boost is absent —, so the values below are oracle output, pinned
from the state of the oracle that passed every cross-check of tests/test_oracle_crosschecks.py and the reader/murmur reference pins).

  a2          per adversarial case (tests/adversarial.py): number of minimizers, sha256 over the (hash, wpos, strand) int32/uint32 arrays
  min_hits    estimateMinimumHitsRelaxed(s, k = 16, pi = 80) for s = 1 .. 12000, as the list of s at which the value steps up
  accept_min  smallest shared count with nucIdentityUpperBound >= 80 (computeMap.hpp:415) for a ladder of sketch sizes
  config0     BASELINE configs[0]: 10-genome mini DB (synth.make_db, seed 7), 1000 reads of 5 kb (synth.make_reads, seed 1) through the
              oracle CLI: sha256 of PREFIX, .meta, .meta.unmappedReadsLengths, .EM, .EM.reads2Taxon, .EM.WIMP, the first lines of
              PREFIX and the whole WIMP as text, the EM log-likelihood of every round
Run from the repository root:  python tests/golden/make_core_golden.py"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def a2_digest(h, w, s):
    m = hashlib.sha256()
    m.update(np.ascontiguousarray(h, dtype="<u4").tobytes()); m.update(np.ascontiguousarray(w, dtype="<i4").tobytes()); m.update(np.ascontiguousarray(s, dtype="<i4").tobytes())
    return m.hexdigest()[:24]


def accept_min(O, s, k=16, pi=80.0):
    lo, hi = 0, s                                                # the upper bound of the identity is monotone in the shared count
    while lo < hi:
        mid = (lo + hi) // 2
        if O.identity(mid, s, k)[1] >= pi:
            hi = mid
        else:
            lo = mid + 1
    return lo


ACCEPT_LADDER = list(range(1, 200)) + list(range(200, 3000, 37)) + list(range(3000, 12001, 450))


def config0(workdir):
    import orc
    from metamaps_amd import synth
    db = synth.make_db(os.path.join(workdir, "db"), n_genomes=10, genome_len=200_000, seed=7)
    rd = synth.make_reads(db, os.path.join(workdir, "reads.fq"), n_reads=1000, read_len=5000, seed=1)
    pre = os.path.join(workdir, "out")
    subprocess.run([orc.CLI, "mapDirectly", "--all", "-r", db.fasta, "-q", rd["path"], "-o", pre, "-t", "8"], check=True, capture_output=True, timeout=3000)
    p = subprocess.run([orc.CLI, "classify", "--DB", db.dir, "--mappings", pre, "--minreads", "3", "-t", "8"], check=True, capture_output=True, timeout=3000)
    sha = lambda suf: hashlib.sha256(open(pre + suf, "rb").read()).hexdigest()[:24]
    lls = json.loads(p.stderr.decode().strip().splitlines()[-1])["ll"]   # the oracle reports the log-likelihood of every round on stderr
    return {"sha": {suf or "PREFIX": sha(suf) for suf in ("", ".meta", ".meta.unmappedReadsLengths", ".EM", ".EM.reads2Taxon", ".EM.WIMP")},
            "parameters": [l for l in open(pre + ".parameters").read().splitlines() if l.split(" ")[0] in ("kmerSize", "windowSize", "minReadLength", "referenceSize", "p_value")],
            "meta": open(pre + ".meta").read(), "first_lines": open(pre).read().splitlines()[:12], "n_lines": sum(1 for _ in open(pre)),
            "wimp": open(pre + ".EM.WIMP").read(), "log_likelihood": lls}


def generate():
    import orc
    from adversarial import adversarial_cases
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    O = orc.Oracle()
    g = {"a2": {}}
    for name, seq, k, w in adversarial_cases():
        h, wp, st = O.minimizers(seq, k, w)
        g["a2"][name] = [int(len(h)), a2_digest(h, wp, st)]
    steps, prev = [], 0
    for s in range(1, 12001):
        v = O.L.orc_min_hits_relaxed(s, 16, 80.0)
        while prev < v:                                          # (a step of more than one lists s several times)
            steps.append(s); prev += 1
        assert v == prev, "minimumHits is not monotone in s"
    g["min_hits"] = {"k": 16, "pi": 80, "s_max": 12000, "steps_up_at": steps}
    g["accept_min"] = {"k": 16, "pi": 80, "s": ACCEPT_LADDER, "min_shared": [accept_min(O, s) for s in ACCEPT_LADDER]}
    with tempfile.TemporaryDirectory() as d:
        g["config0"] = config0(d)
    return g


if __name__ == "__main__":
    out = os.path.join(HERE, "core_golden.json")
    json.dump(generate(), open(out, "w"), indent=0, sort_keys=True)
    print("wrote", out, os.path.getsize(out), "bytes")

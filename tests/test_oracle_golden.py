"""Pins the oracle (oracle/) against everything the reference itself offers for this path:
  * the real reference murmur header compiled in place (oracle/_ref, only where /root/reference exists),
  * the reference's example outputs (tests/golden/example/*, a real miniSeq+H run of the reference),
  * Boost.Math's own answers for the binomial calls (tests/golden/binom_golden.json, via scipy),
  * the known answers the survey recorded from the reference.
CPU only."""
import ctypes as C
import json
import math
import os
import random
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EX = os.path.join(HERE, "golden", "example")

MURMUR_KAT = {b"ACGTACGTACGTACGA": 1717151907, b"TCGTACGTACGTACGT": 3339777804, b"AAAAAAAAAAAAAAAA": 2987007239,
              b"TTTTTTTTTTTTTTTT": 1255594208, b"GATTACAGATTACAGA": 3069404967, b"NNNNNNNNNNNNNNNN": 3389057319}


def test_murmur_known_answers(oracle_lib):
    for s, v in MURMUR_KAT.items():
        assert oracle_lib.kmer_hash(s, 16) == v


def test_murmur_against_real_reference_header(oracle_lib):
    import orc
    if not os.path.exists(orc.REF_LIB):
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    ref = C.CDLL(orc.REF_LIB)
    ref.ref_kmer_hash.restype = C.c_uint32
    rnd = random.Random(9)
    for _ in range(20000):
        k = rnd.randint(1, 80)
        s = bytes(rnd.choice(b"ACGTNacgtRYKM") for _ in range(k))
        assert oracle_lib.kmer_hash(s, k) == ref.ref_kmer_hash(s, k)


def test_minimizer_known_answer(oracle_lib):
    seq = b"ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGGCTAGCTAGGCTAAGCTTTCGAGGATCGCGATATCGGCTAGGGATTCAGGCTAGCATCGACTAGCATCGGATC"
    h, w, s = oracle_lib.minimizers(seq, 16, 8)
    assert len(h) == 19
    assert list(zip(h[:5].tolist(), w[:5].tolist(), s[:5].tolist())) == [
        (72677884, 0, 1), (353807513, 7, -1), (439792174, 11, 1), (187768072, 12, 1), (285387513, 20, -1)]


def test_survey_known_answers(oracle_lib):
    assert [oracle_lib.L.orc_min_hits_relaxed(s, 16, 80.0) for s in (250, 500, 785, 1176, 2353)] == [3, 7, 11, 18, 39]


def _params():
    return dict(l.rstrip("\n").split(" ", 1) for l in open(os.path.join(EX, "example.parameters")) if " " in l)


def test_window_size_of_example_run(oracle_lib):
    p = _params()
    w = oracle_lib.L.orc_recommended_window(float(p["p_value"]), int(p["kmerSize"]), float(p["percentageIdentity"]),
                                            int(p["minReadLength"]), int(p["referenceSize"]))
    assert w == int(p["windowSize"]) == 16
    assert oracle_lib.L.orc_recommended_window(1e-3, 16, 80.0, 1000, int(p["referenceSize"])) == 8


def _em_lines():
    return [l.rstrip("\n").split(" ") for l in open(os.path.join(EX, "example.EM"))]


def test_identity_text_of_every_example_mapping(oracle_lib):
    """fields 10 and 13 of all 2985 reference mappings are pure functions of (shared, sketch, k)."""
    lines = _em_lines()
    assert len(lines) == 2985
    for f in lines:
        shared, sk = int(f[10]), int(f[11])
        ident, ub = oracle_lib.identity(shared, sk, 16)
        assert f"{ident:g}" == f[9], f
        assert ub >= 80.0                                   # it was reported, so it passed the filter (computeMap.hpp:415)
        corrected = np.float32(np.exp(-(1 - float(f[9]) / 100.0)))
        assert f"{float(corrected * np.float32(100)):g}" == f[12], f
        assert int(f[8]) == int(f[7]) + int(f[1]) - 1 and f[2] == "0" and int(f[3]) == int(f[1]) - 1


def test_accept_threshold_consistent_with_example(oracle_lib):
    """the smallest shared count the reference reported for a sketch size can never be below min_hits_relaxed."""
    best = {}
    for f in _em_lines():
        sk, sh = int(f[11]), int(f[10])
        best[sk] = min(best.get(sk, 10**9), sh)
    for sk, sh in best.items():
        assert sh >= oracle_lib.L.orc_min_hits_relaxed(sk, 16, 80.0), (sk, sh)


def test_reads2taxon_is_first_argmax_of_posterior():
    groups, order = {}, []
    for f in _em_lines():
        if f[0] not in groups:
            groups[f[0]] = []; order.append(f[0])
        groups[f[0]].append(f)
    r2t = [l.rstrip("\n").split("\t") for l in open(os.path.join(EX, "example.EM.reads2Taxon"))]
    unm = [l.rstrip("\n").split("\t")[1] for l in open(os.path.join(EX, "example.meta.unmappedReadsLengths"))]
    assert [x[0] for x in r2t] == order + unm
    n_checked = 0
    for rid, tax in r2t[:len(order)]:
        g = groups[rid]
        post = [float(x[13]) for x in g]
        assert abs(sum(post) - 1) < 1e-4
        top = max(post)
        cands = {re.search(r"kraken:taxid\|(x?\d+)", x[5]).group(1) for x, p in zip(g, post) if p == top}
        if len(cands) == 1:                                 # %f text can tie where the doubles did not
            assert tax in cands; n_checked += 1
    assert n_checked > 50
    assert all(t == "0" for _, t in r2t[len(order):])


def test_wimp_counts_and_pot_frequency():
    meta = dict(l.split() for l in open(os.path.join(EX, "example.meta")))
    total, short, mapped, notm = (int(meta[k]) for k in ("TotalReads", "ReadsTooShort", "ReadsMapped", "ReadsNotMapped"))
    assert total == short + mapped + notm
    r2t = [l.rstrip("\n").split("\t") for l in open(os.path.join(EX, "example.EM.reads2Taxon"))]
    cnt = {}
    for _, t in r2t:
        cnt[t] = cnt.get(t, 0) + 1
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(EX, "example.EM.WIMP"))][1:]
    dg = [r for r in rows if r[0] == "definedGenomes"]
    for r in dg:
        if r[1] not in ("0", "-3"):
            assert int(r[3]) == cnt.get(r[1], 0)
            assert abs(float(r[5]) - float(r[4]) * mapped / (total - short)) < 2e-6 * max(1.0, float(r[4]))   # fEM.h:171
    uncl = [r for r in dg if r[2] == "Unclassified"][0]
    assert int(uncl[3]) == notm and abs(float(uncl[5]) - notm / (total - short)) < 1e-6


def test_binomial_against_boost(oracle_lib):
    g = json.load(open(os.path.join(HERE, "golden", "binom_golden.json")))
    for n, p, q, x in g["quantile_upper"]:
        assert oracle_lib.L.orc_binom_quantile_upper(n, p, q) == x, (n, p)
    for n, p, k, v in g["pmf"]:
        got = oracle_lib.L.orc_binom_pmf(n, p, k)
        assert got == pytest.approx(v, rel=1e-10, abs=1e-300)
    for n, p, k, v in g["sf"]:
        got = oracle_lib.L.orc_binom_sf(n, p, k)
        assert got == pytest.approx(v, rel=1e-9, abs=1e-300)


def test_mapq_text_roundtrip_matches_reference_format(oracle_lib):
    """add_mapping_qualities appends two fields; qualities of a read sum to 1 (mapWrap.h:300-320)."""
    lines = ["r 6578 0 6577 + c1 100 1 6578 84.4464 34 785", "r 6578 0 6577 + c2 100 5 6582 81.2377 20 785",
             "r 6578 0 6577 - c3 100 9 6586 84.4464 34 785"]
    out = oracle_lib.add_mapq(16, lines)
    f = [l.split(" ") for l in out]
    assert all(len(x) == 14 for x in f)
    assert [x[12] for x in f] == ["85.5956", "82.8927", "85.5956"]        # the example file's field 13 for these identities
    q = [float(x[13]) for x in f]
    assert abs(sum(q) - 1) < 1e-5 and q[0] == q[2] and q[1] < q[0]


def test_unknown_species_coverage_columns_of_example(oracle_lib):
    """The coverage columns of the reference's own example.EM.evidenceUnknownSpecies (fEM.h:1070-1114): average reads per
    usable window, expected zero-coverage windows (Poisson) and the binomial tail — known answers from a real run, i.e.
    from the Boost.Math the reference was built with.  The file prints six decimals; the read count behind the printed
    average is recovered as the integer that reproduces it."""
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(EX, "example.EM.evidenceUnknownSpecies"))]
    assert rows[0][9:] == ["coverageWindows_usable_averageCoverage", "coverageWindows_usable_coverageIsZero",
                           "coverageWindows_usable_coverageIsZero_expected", "coverageWindows_usable_coverageIsZero_P"]
    checked = 0
    for r in rows[1:]:
        usable, avg_s, zero, exp_s, p_s = int(r[8]), r[9], int(r[10]), r[11], r[12]
        if avg_s == "NA":
            continue
        reads = round(float(avg_s) * usable)
        assert "%f" % (reads / usable) == avg_s
        if reads == 0:
            assert (exp_s, p_s) == (str(usable), "1")
            continue
        p0 = math.exp(-reads / usable)
        assert "%f" % (usable * p0) == exp_s
        pv = 1.0 if zero == 0 else 1 - oracle_lib.L.orc_binom_cdf_sum(usable, p0, zero - 1)
        assert "%f" % pv == p_s, (r[0], pv, p_s)
        checked += 1
    assert checked >= 10


def test_unknown_species_distribution_functions_vs_scipy(oracle_lib):
    """chi-square(1) cdf and binomial cdf against scipy (which embeds Boost.Math) over the reachable domain."""
    from scipy import stats
    rng = np.random.default_rng(5)
    for x in list(rng.uniform(0, 40, 200)) + [0.0, 1e-12, 3.841458820694124, 700.0]:
        assert oracle_lib.L.orc_chi2_1df_cdf(x) == pytest.approx(stats.chi2.cdf(x, 1), rel=1e-12, abs=1e-300)
    for _ in range(300):
        n = int(rng.integers(1, 20000)); lam = float(10 ** rng.uniform(-4, 1.5)); p = math.exp(-lam)
        k = int(min(n, max(0, rng.normal(n * p, 3 * math.sqrt(n * p * (1 - p)) + 1))))
        assert oracle_lib.L.orc_binom_cdf_sum(n, p, k) == pytest.approx(stats.binom.cdf(k, n, p), rel=1e-9, abs=1e-14)


# ---- regression pins of the oracle's integer core (tests/golden/core_golden.json, written by tests/golden/make_core_golden.py)
def _core_golden():
    return json.load(open(os.path.join(HERE, "golden", "core_golden.json")))


def test_core_golden_adversarial_minimizers(oracle_lib):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_core_golden import a2_digest
    from adversarial import adversarial_cases
    g = _core_golden()["a2"]
    cases = adversarial_cases()
    assert len(cases) == len(g) >= 290
    n_nonempty = 0
    for name, seq, k, w in cases:
        h, wp, st = oracle_lib.minimizers(seq, k, w)
        assert [int(len(h)), a2_digest(h, wp, st)] == g[name], name
        n_nonempty += len(h) > 0
    assert n_nonempty > 200


def test_core_golden_min_hits_and_accept_tables(oracle_lib):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_core_golden import accept_min
    g = _core_golden()
    steps = np.array(g["min_hits"]["steps_up_at"])
    for s in list(range(1, 400)) + list(range(400, 12001, 97)) + [785, 1176, 2353, 12000]:
        assert oracle_lib.L.orc_min_hits_relaxed(s, 16, 80.0) == int((steps <= s).sum()), s
    for s, v in list(zip(g["accept_min"]["s"], g["accept_min"]["min_shared"]))[::3]:
        assert accept_min(oracle_lib, s) == v, s


def test_core_golden_config0_through_the_oracle_cli(oracle_lib, tmp_path):
    """BASELINE configs[0] (1000 x 5 kb reads vs the 10-genome mini DB) through the oracle CLI reproduces the committed file hashes"""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_core_golden import config0
    g = _core_golden()["config0"]
    now = config0(str(tmp_path))
    assert now["sha"] == g["sha"] and now["meta"] == g["meta"] and now["first_lines"] == g["first_lines"]
    assert now["wimp"] == g["wimp"] and now["log_likelihood"] == g["log_likelihood"] and len(g["log_likelihood"]) >= 3

"""The adversarial sequence set for winnowing: deterministic, shared by the golden-vector
generator (tests/golden/make_core_golden.py), the oracle regression test and the GPU parity test."""
import random


def adversarial_cases(n_random: int = 260, seed: int = 20260928):
    """[(name, sequence bytes, k, w)]"""
    rng = random.Random(seed)
    rs = lambda n, a="ACGT": "".join(rng.choice(a) for _ in range(n))
    base = rs(400)
    fixed = [
        ("survey_107mer", "ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGGCTAGCTAGGCTAAGCTTTCGAGGATCGCGATATCGGCTAGGGATTCAGGCTAGCATCGACTAGCATCGGATC", 16, 8),
        ("homopolymer_A", "A" * 300, 16, 8), ("homopolymer_T", "T" * 300, 16, 5),
        ("palindrome_ACGT", "ACGT" * 80, 16, 8), ("palindrome_AT", "AT" * 150, 16, 8), ("palindrome_k8", "ACGTACGT" * 40, 8, 4),
        ("dinucleotide_CA", "CA" * 160, 16, 8), ("tandem_17", rs(17) * 30, 16, 8), ("tandem_k", rs(16) * 30, 16, 10),
        ("all_N", "N" * 200, 16, 8), ("N_run_inside", base[:150] + "N" * 60 + base[150:], 16, 8), ("N_every_20", "".join(c if i % 20 else "N" for i, c in enumerate(base)), 16, 8),
        ("lower_case", base.lower(), 16, 8), ("mixed_case", "".join(c.lower() if i % 3 else c for i, c in enumerate(base)), 16, 8),
        ("iupac", rs(300, "ACGTRYKMSWN"), 16, 8), ("len_eq_k", base[:16], 16, 8), ("len_eq_k_plus_w_minus_1", base[:23], 16, 8),
        ("len_eq_k_plus_w", base[:24], 16, 8), ("len_lt_k", base[:15], 16, 8), ("len_eq_w", base[:8], 5, 8), ("len_1", "A", 5, 3),
        ("w_1", base[:200], 16, 1), ("w_100", base, 16, 100), ("w_gt_len", base[:60], 16, 90), ("k_5", base, 5, 8), ("k_21", base, 21, 11), ("k_32", base, 32, 16),
        ("revcomp_symmetric_whole", base[:100] + "".join({"A": "T", "C": "G", "G": "C", "T": "A"}[c] for c in reversed(base[:100])), 16, 8),
        ("first_minimizer_repeats", (base[:24] + "G") * 12, 16, 8), ("two_equal_windows", base[:40] * 2, 16, 8),
    ]
    cases = [(n, s.encode(), k, w) for n, s, k, w in fixed]
    for i in range(n_random):
        k = rng.choice([5, 8, 11, 16, 16, 16, 21, 32])
        w = rng.choice([1, 2, 3, 5, 8, 8, 13, 16, 25, 50, 100])
        n = rng.choice([k, k + w - 1, k + w, 40, 150, 700, 3000])
        kind = i % 5
        if kind == 0:
            s = rs(n)
        elif kind == 1:
            s = rs(n, "ACGTN" if i % 2 else "ACGTacgtn")
        elif kind == 2:
            unit = rs(rng.randrange(1, 12))
            s = (unit * (n // len(unit) + 1))[:n]
        elif kind == 3:
            half = rs(n // 2 + 1)
            s = (half + "".join({"A": "T", "C": "G", "G": "C", "T": "A"}[c] for c in reversed(half)))[:max(n, 1)]
        else:
            s = rs(n, "AC")                                       # low complexity: many equal hashes in a window
        cases.append((f"rnd{i}_k{k}_w{w}", s.encode(), k, w))
    return cases

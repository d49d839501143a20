"""The parallel restatements the HIP kernels implement, checked against the oracle on the CPU."""
import random

from pure_forms import winnow_pure

COMP = {65: 84, 84: 65, 67: 71, 71: 67}


def _hashes(o, seq, k):
    n = len(seq) - k + 1
    rc = bytes(COMP.get(c, c) for c in reversed(seq))
    hf = [o.kmer_hash(seq[i:i + k], k) for i in range(n)]
    hb = [o.kmer_hash(rc[len(seq) - i - k:len(seq) - i], k) for i in range(n)]
    return hf, hb


def test_winnow_pure_form_equals_deque_loop(oracle_lib):
    rnd = random.Random(5)
    dup = 0
    for it in range(400):
        k = rnd.choice([4, 5, 8, 16, 16, 21]); w = rnd.randint(1, 20)
        L = rnd.randint(max(k, w), 300)
        mode = rnd.random()
        if mode < 0.3:
            seq = bytes(rnd.choice(b"ACGT") for _ in range(L))
        elif mode < 0.5:
            unit = bytes(rnd.choice(b"ACGT") for _ in range(rnd.randint(1, 6))); seq = (unit * (L // len(unit) + 1))[:L]
        elif mode < 0.7:
            seq = bytes(rnd.choice(b"ACGTNNN") for _ in range(L))
        elif mode < 0.85:
            h = bytes(rnd.choice(b"ACGT") for _ in range(L // 2)); seq = h + bytes(COMP[c] for c in reversed(h))
        else:
            seq = bytes(rnd.choice(b"AC") for _ in range(L))
        hf, hb = _hashes(oracle_lib, seq, k)
        H, W, S = oracle_lib.minimizers(seq, k, w)
        ref = list(zip(H.tolist(), W.tolist(), S.tolist()))
        assert winnow_pure(hf, hb, w) == ref, (k, w, seq)
        dup += sum(1 for a, b in zip(ref, ref[1:]) if a[0] == b[0])
    assert dup > 100      # the equal-hash re-emission rule (commonFunc.hpp:157) was exercised

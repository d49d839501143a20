"""The C-ABI library on a box without a GPU: it loads, exports every symbol the header declares, refuses to
create a context (no CPU fallback), and its host-side statistics agree with the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from metamaps_amd import capi


def test_exports_every_declared_symbol():
    L = capi.lib()
    names = capi.declared_symbols()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.mm_abi_version() == 6


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert capi.lib().mm_ctx_create(0, C.byref(h)) != 0
    with pytest.raises(capi.MMError):
        capi.Context(0)


def test_host_statistics_match_oracle(oracle_lib):
    L = capi.lib()
    for s in list(range(1, 60)) + [100, 250, 453, 500, 785, 1176, 2353, 4985, 11000]:
        for pi in (80.0, 85.0, 70.0, 95.0):
            assert L.mm_min_hits_relaxed(s, 16, pi) == oracle_lib.L.orc_min_hits_relaxed(s, 16, pi), (s, pi)
        for sh in sorted({0, 1, s // 20, s // 12, s // 8, s // 2, s}):
            a, b = C.c_float(), C.c_float()
            L.mm_identity(sh, s, 16, C.byref(a), C.byref(b))
            assert (a.value, b.value) == oracle_lib.identity(sh, s, 16), (s, sh)
    for L_, R in ((1000, 26762276280), (2000, 26762276280), (1000, 2025931), (1000, 160_000_000), (1000, 300_000_000_000)):
        assert L.mm_recommended_window(1e-3, 16, 80.0, L_, R) == oracle_lib.L.orc_recommended_window(1e-3, 16, 80.0, L_, R)
    assert L.mm_recommended_window(1e-3, 16, 80.0, 2000, 26762276280) == 16      # the reference's example run


def test_freq_threshold_rule():
    """winSketch.hpp:452-494 on a histogram: ignore = int64(float(unique)*0.001f/100)."""
    L = capi.lib()
    counts = np.array([1, 2, 3, 10, 50], dtype=np.int64)
    nh = np.array([10_000_000, 500_000, 1000, 60, 40], dtype=np.int64)
    uniq = int(nh.sum())
    # ignore = 105; from the top: 40 (<105 -> thr 50), 100 (<105 -> thr 10), 1100 (>105 -> stop)
    thr = L.mm_freq_threshold_from_hist(counts.ctypes.data_as(C.c_void_p), nh.ctypes.data_as(C.c_void_p), 5, uniq, 2**31 - 1)
    assert thr == 10
    nh2 = np.array([100, 5, 1, 1, 1], dtype=np.int64)   # tiny index: ignore = 0 -> first step already exceeds -> keep previous
    thr = L.mm_freq_threshold_from_hist(counts.ctypes.data_as(C.c_void_p), nh2.ctypes.data_as(C.c_void_p), 5, int(nh2.sum()), 2**31 - 1)
    assert thr == 2**31 - 1

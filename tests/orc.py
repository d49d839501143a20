"""ctypes access to the oracle (oracle/_build/liborc.so).  Tests only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "liborc.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_murmur.so")
CLI = os.path.join(ROOT, "oracle", "_build", "metamaps_oracle")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self):
        L = C.CDLL(LIB)
        L.orc_kmer_hash.restype = C.c_uint32
        L.orc_minimizers.restype = C.c_long
        L.orc_minimizers.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.orc_binom_pmf.restype = C.c_double
        L.orc_binom_pmf.argtypes = [C.c_int, C.c_double, C.c_int]
        L.orc_binom_sf.restype = C.c_double
        L.orc_binom_sf.argtypes = [C.c_int, C.c_double, C.c_int]
        L.orc_binom_cdf_sum.restype = C.c_double
        L.orc_binom_cdf_sum.argtypes = [C.c_uint64, C.c_double, C.c_uint64]
        L.orc_chi2_1df_cdf.restype = C.c_double
        L.orc_chi2_1df_cdf.argtypes = [C.c_double]
        L.orc_binom_quantile_upper.argtypes = [C.c_int, C.c_double, C.c_double]
        L.orc_min_hits_relaxed.argtypes = [C.c_int, C.c_int, C.c_float]
        L.orc_recommended_window.argtypes = [C.c_double, C.c_int, C.c_float, C.c_int, C.c_uint64]
        L.orc_identity.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_add_mapq.restype = C.c_long
        L.orc_add_mapq.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_long]
        L.orc_index_build.restype = C.c_void_p
        L.orc_index_build.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.orc_index_free.argtypes = [C.c_void_p]
        for n in ("orc_index_entries", "orc_index_contigs", "orc_index_unique_hashes"):
            getattr(L, n).restype = C.c_long
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_index_freq_threshold.argtypes = [C.c_void_p]
        L.orc_index_set_freq_threshold.argtypes = [C.c_void_p, C.c_int]
        L.orc_index_dump.argtypes = [C.c_void_p] * 5
        L.orc_index_contig_len.argtypes = [C.c_void_p, C.c_long]
        L.orc_map_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                   C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long]
        L.orc_map_directly.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_void_p]
        L.orc_classify.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
        L.orc_finish_from_posteriors.argtypes = [C.c_char_p] * 4
        L.orc_classify_from.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
        self.L = L

    def kmer_hash(self, s: bytes, k: int) -> int:
        return self.L.orc_kmer_hash(s, k)

    def minimizers(self, seq: bytes, k: int, w: int):
        cap = max(len(seq), 1)
        h = np.zeros(cap, dtype=np.uint32); wp = np.zeros(cap, dtype=np.int32); st = np.zeros(cap, dtype=np.int32)
        n = self.L.orc_minimizers(seq, len(seq), k, w, _p(h), _p(wp), _p(st), cap)
        return h[:n], wp[:n], st[:n]

    def identity(self, shared, s, k):
        a = C.c_float(); b = C.c_float()
        self.L.orc_identity(shared, s, k, C.byref(a), C.byref(b))
        return a.value, b.value

    def add_mapq(self, k: int, lines):
        out = C.create_string_buffer(1 << 20)
        n = self.L.orc_add_mapq(k, "\n".join(lines).encode(), out, 1 << 20)
        assert n >= 0
        return out.value.decode().split("\n")

    def index(self, fasta: str, k: int, w: int):
        return OracleIndex(self, fasta, k, w)


class OracleIndex:
    def __init__(self, orc: Oracle, fasta: str, k: int, w: int):
        self.o = orc
        self.h = orc.L.orc_index_build(fasta.encode(), k, w)
        assert self.h
        self.n = orc.L.orc_index_entries(self.h)
        self.n_contigs = orc.L.orc_index_contigs(self.h)
        self.n_unique = orc.L.orc_index_unique_hashes(self.h)
        self.freq_threshold = orc.L.orc_index_freq_threshold(self.h)

    def set_freq_threshold(self, thr: int):
        self.o.L.orc_index_set_freq_threshold(self.h, int(thr))
        self.freq_threshold = int(thr)

    def dump(self):
        h = np.zeros(self.n, dtype=np.uint32); s = np.zeros(self.n, dtype=np.int32)
        w = np.zeros(self.n, dtype=np.int32); st = np.zeros(self.n, dtype=np.int32)
        self.o.L.orc_index_dump(self.h, _p(h), _p(s), _p(w), _p(st))
        return h, s, w, st

    def map_read(self, seq: bytes, pi: float = 80.0):
        cap = len(seq) + 16
        n = np.zeros(5, dtype=np.int32)
        skh = np.zeros(cap, dtype=np.uint32); sks = np.zeros(cap, dtype=np.int32)
        hcap = 1 << 23
        hs = np.zeros(hcap, dtype=np.int32); hw = np.zeros(hcap, dtype=np.int32)
        ccap = 4096
        cand = np.zeros((ccap, 3), dtype=np.int32); l2 = np.zeros((ccap, 5), dtype=np.int64); mp = np.zeros((ccap, 6), dtype=np.int32)
        self.o.L.orc_map_read(self.h, seq, len(seq), pi, _p(n), _p(skh), _p(sks), cap, _p(hs), _p(hw), hcap, _p(cand), ccap, _p(l2), _p(mp), ccap)
        s, nh, mh, nc, nm = [int(x) for x in n]
        assert nh <= hcap and nc <= ccap
        return {"sketch_hash": skh[:s], "sketch_strand": sks[:s], "hit_contig": hs[:nh], "hit_wpos": hw[:nh], "min_hits": mh,
                "cand": cand[:nc], "l2": l2[:nc], "map": mp[:nm]}

    def close(self):
        self.o.L.orc_index_free(self.h)

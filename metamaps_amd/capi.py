"""ctypes binding of include/metamaps_hip.h (libmetamaps_hip.so).

Thin and literal: every wrapper is one C-ABI call (plus buffer allocation), so the parity tests exercise
exactly what a host program in another language would call.  There is no fallback: if the shared library
is missing, loading raises; if no gfx950 GPU is present, Context() raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmetamaps_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "metamaps_hip.h")

COMM_ID_BYTES = 128


class MMError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libmetamaps_hip status {status}: {msg}")
        self.status = status


SEED_STAGE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


class MapParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("w", C.c_int32), ("perc_identity", C.c_float), ("min_read_len", C.c_int32)]


class MapRecord(C.Structure):
    _fields_ = [("read", C.c_int32), ("ref_contig", C.c_int32), ("ref_start", C.c_int32), ("shared", C.c_int32),
                ("sketch", C.c_int32), ("strand", C.c_int32), ("mapq", C.c_double)]


RECORD_DTYPE = np.dtype([("read", "<i4"), ("ref_contig", "<i4"), ("ref_start", "<i4"), ("shared", "<i4"),
                         ("sketch", "<i4"), ("strand", "<i4"), ("mapq", "<f8")])
assert RECORD_DTYPE.itemsize == C.sizeof(MapRecord)


class MapStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_reads", "n_reads_long_enough", "n_reads_mapped", "n_mappings",
                                           "bases_long_enough", "sum_sketch", "sum_hits", "n_candidates",
                                           "sum_l2_stream_entries", "sum_l2_evals", "n_ambiguous_sketch_reads", "sum_hits_kept", "n_l2_rebuilds", "n_l2_wide_redo")] + \
               [(n, C.c_double) for n in ("ms_minimizer", "ms_sketch", "ms_probe_gather", "ms_sort_hits", "ms_l1_scan",
                                          "ms_l2", "ms_compact", "ms_total", "ms_hit_filter")] + \
               [("n_reads_giant", C.c_int64)]

    def as_dict(self):
        return {n: (int(getattr(self, n)) if t is C.c_int64 else float(getattr(self, n))) for n, t in self._fields_}


class IndexInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_contigs", "n_entries", "n_unique_hashes", "n_dup_flagged", "hbm_bytes")]


class SynthRefParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_species", C.c_int32), ("strains_per_species", C.c_int32),
                ("genome_len", C.c_int32), ("strain_divergence", C.c_float), ("genus_divergence", C.c_float)]


class SynthReadParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_reads", C.c_int64), ("read_len", C.c_int32), ("sub_rate", C.c_float),
                ("ins_rate", C.c_float), ("del_rate", C.c_float), ("frac_random", C.c_float), ("n_abundant", C.c_int32), ("read_len_min", C.c_int32)]


class SynthCommunityParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_genomes", C.c_int32), ("n_species", C.c_int32), ("n_genera", C.c_int32),
                ("median_len", C.c_double), ("sigma_len", C.c_double), ("min_len", C.c_int32), ("max_len", C.c_int32),
                ("strain_div_min", C.c_float), ("strain_div_max", C.c_float), ("genus_div_min", C.c_float), ("genus_div_max", C.c_float),
                ("strain_indel_events", C.c_int32), ("human_contigs", C.c_int32), ("human_bases", C.c_int64),
                ("repeat_fraction", C.c_float), ("n_fraction", C.c_float), ("n_repeat_families", C.c_int32), ("total_bases_target", C.c_int64)]


def declared_symbols(header: str = HEADER_PATH) -> list[str]:
    """Every function the public header declares (used by the CPU test that checks the exports)."""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("MM_LIB_PATH") or LIB_PATH            # (MM_LIB_PATH: an alternative build of the library, tools/ab.sh — A/B runs of two kernel variants on one box)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        vp, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
        P = C.POINTER
        sig = {
            "mm_abi_version": (C.c_int, []),
            "mm_ctx_release_cached": (C.c_int, [vp]),
            "mm_device_count": (C.c_int, []),
            "mm_ctx_create": (C.c_int, [C.c_int, P(vp)]),
            "mm_ctx_destroy": (None, [vp]),
            "mm_last_error": (C.c_char_p, [vp]),
            "mm_ctx_device_info": (C.c_int, [vp, C.c_char_p, C.c_size_t, P(C.c_int), P(u64), P(u64)]),
            "mm_ctx_synchronize": (C.c_int, [vp]),
            "mm_ctx_stream": (vp, [vp]),
            "mm_seqset_create": (C.c_int, [vp, P(vp)]),
            "mm_seqset_destroy": (None, [vp]),
            "mm_seqset_add": (C.c_int, [vp, C.c_char_p, i64]),
            "mm_seqset_add_view": (C.c_int, [vp, C.c_char_p, i64]),
            "mm_seqset_save": (C.c_int, [vp, C.c_char_p]),
            "mm_seqset_load": (C.c_int, [vp, C.c_char_p, P(vp)]),
            "mm_seqset_upload": (C.c_int, [vp]),
            "mm_seqset_slice": (C.c_int, [vp, vp, i64, i64, P(vp)]),
            "mm_seqset_concat": (C.c_int, [vp, P(vp), C.c_int, P(vp)]),
            "mm_seqset_count": (i64, [vp]),
            "mm_seqset_total_bases": (i64, [vp]),
            "mm_seqset_lengths": (C.c_int, [vp, vp]),
            "mm_seqset_fetch": (C.c_int, [vp, i64, C.c_char_p, i64]),
            "mm_seqset_fetch_range": (C.c_int, [vp, i64, i64, C.c_void_p, i64]),
            "mm_synth_reference": (C.c_int, [vp, P(SynthRefParams), P(vp)]),
            "mm_synth_reads": (C.c_int, [vp, vp, P(SynthReadParams), P(vp), vp]),
            "mm_synth_community": (C.c_int, [vp, P(SynthCommunityParams), P(vp), vp]),
            "mm_synth_community_species": (C.c_int, [P(SynthCommunityParams), vp]),
            "mm_minimizers": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, i64]),
            "mm_index_build": (C.c_int, [vp, vp, C.c_int, C.c_int, P(vp)]),
            "mm_index_save": (C.c_int, [vp, C.c_char_p]),
            "mm_index_load": (C.c_int, [vp, C.c_char_p, P(vp)]),
            "mm_index_destroy": (None, [vp]),
            "mm_index_get_info": (C.c_int, [vp, P(IndexInfo)]),
            "mm_index_freq_hist": (C.c_int, [vp, vp, vp, i64, P(i64)]),
            "mm_freq_threshold_from_hist": (C.c_int, [vp, vp, i64, i64, C.c_int]),
            "mm_index_set_freq_threshold": (C.c_int, [vp, C.c_int]),
            "mm_index_entries": (C.c_int, [vp, vp, vp, vp, vp, i64]),
            "mm_index_dup_neighbours": (C.c_int, [vp, vp, vp, i64]),
            "mm_recommended_window": (C.c_int, [f64, C.c_int, f32, C.c_int, u64]),
            "mm_estimate_pvalue": (f64, [C.c_int, C.c_int, f32, C.c_int, u64]),
            "mm_min_hits_relaxed": (C.c_int, [C.c_int, C.c_int, f32]),
            "mm_identity": (None, [C.c_int, C.c_int, C.c_int, P(f32), P(f32)]),
            "mm_map_batch": (C.c_int, [vp, vp, vp, P(MapParams), P(vp)]),
            "mm_map_batch_phased": (C.c_int, [vp, vp, vp, P(MapParams), SEED_STAGE_CB, vp, P(vp)]),
            "mm_map_batch_reusing": (C.c_int, [vp, vp, vp, P(MapParams), vp, P(vp)]),
            "mm_sketch_batch": (C.c_int, [vp, vp, P(MapParams), P(vp)]),
            "mm_mapping_destroy": (None, [vp]),
            "mm_mapping_get_stats": (C.c_int, [vp, P(MapStats)]),
            "mm_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "mm_mapping_gather": (C.c_int, [vp, C.c_int, i64, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
            "mm_mapping_release_intermediates": (C.c_int, [vp]),
            "mm_mapping_fetch": (C.c_int, [vp, vp, vp, i64]),
            "mm_mapping_add_qualities": (C.c_int, [vp, vp, vp, C.c_int]),
            "mm_mapping_concat": (C.c_int, [vp, P(vp), vp, C.c_int, P(vp)]),
            "mm_mapping_keep_best": (C.c_int, [vp, vp, C.c_int]),
            "mm_mapping_from_parts": (C.c_int, [vp, i64, vp, P(MapParams), C.c_int, P(vp), P(vp), vp, P(vp)]),
            "mm_index_plan_chunks": (C.c_int, [vp, vp, u64, vp, i32, P(i32)]),
            "mm_debug_sketch": (C.c_int, [vp, vp, vp, vp, i64]),
            "mm_debug_hits": (C.c_int, [vp, vp, vp, vp, i64]),
            "mm_debug_candidates": (C.c_int, [vp, vp, vp, i64]),
            "mm_debug_l2": (C.c_int, [vp, vp, i64]),
            "mm_debug_min_hits": (C.c_int, [vp, vp]),
            "mm_debug_probed_lists": (C.c_int, [vp, vp, vp, i32]),
            "mm_em_create": (C.c_int, [vp, i64, vp, vp, vp, vp, i32, P(vp)]),
            "mm_em_create_from_mapping": (C.c_int, [vp, vp, vp, vp, i32, i32, P(vp)]),
            "mm_em_taxon_counts": (C.c_int, [vp, vp]),
            "mm_em_sizes": (C.c_int, [vp, P(i64), P(i64), P(i32)]),
            "mm_em_destroy": (None, [vp]),
            "mm_em_iterate": (C.c_int, [vp, vp, vp, P(f64)]),
            "mm_em_iterate_allreduce": (C.c_int, [vp, vp, vp, P(f64)]),
            "mm_em_posteriors": (C.c_int, [vp, vp, vp, vp]),
            "mm_em_run": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, P(C.c_int)]),
            "mm_em_continue": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, P(C.c_int), P(C.c_int)]),
            "mm_comm_unique_id": (C.c_int, [C.c_char_p]),
            "mm_comm_init": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int]),
            "mm_comm_allreduce_f64": (C.c_int, [vp, vp, i64]),
            "mm_comm_share": (C.c_int, [vp, vp]),
            "mm_comm_destroy": (None, [vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    def __init__(self, device: int = 0):
        self.h = C.c_void_p()
        st = lib().mm_ctx_create(device, C.byref(self.h))
        if st != 0:
            raise MMError(st, "mm_ctx_create failed (no gfx950 GPU visible? this library has no CPU fallback)")

    def check(self, st: int):
        if st != 0:
            raise MMError(st, lib().mm_last_error(self.h).decode(errors="replace"))

    def close(self):
        if self.h:
            lib().mm_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus = C.c_int()
        tot, free = C.c_uint64(), C.c_uint64()
        self.check(lib().mm_ctx_device_info(self.h, name, 256, C.byref(cus), C.byref(tot), C.byref(free)))
        return {"name": name.value.decode(), "cus": cus.value, "hbm_total": tot.value, "hbm_free": free.value}

    def release_cached(self):
        """cached free device blocks of this context (and the device's recycled index-scale blocks) back to the driver"""
        self.check(lib().mm_ctx_release_cached(self.h))

    def synchronize(self):
        self.check(lib().mm_ctx_synchronize(self.h))

    @property
    def stream(self) -> int:
        return lib().mm_ctx_stream(self.h) or 0

    # ---- sequences
    def seqset(self, seqs) -> "SeqSet":
        h = C.c_void_p()
        self.check(lib().mm_seqset_create(self.h, C.byref(h)))
        s = SeqSet(self, h)
        for q in seqs:
            if isinstance(q, str):
                q = q.encode()
            q = bytes(q)
            self.check(lib().mm_seqset_add(h, q, len(q)))
        self.check(lib().mm_seqset_upload(h))
        return s

    def load_seqset(self, path: str) -> "SeqSet":
        h = C.c_void_p()
        self.check(lib().mm_seqset_load(self.h, path.encode(), C.byref(h)))
        return SeqSet(self, h)

    def synth_reference(self, **kw) -> "SeqSet":
        p = SynthRefParams(**kw)
        h = C.c_void_p()
        self.check(lib().mm_synth_reference(self.h, C.byref(p), C.byref(h)))
        return SeqSet(self, h)

    def synth_community(self, **kw):
        """(reference, contig -> genome) of the SURVEY D1 community; genome n_genomes is the human-like one"""
        p = SynthCommunityParams(**kw)
        h = C.c_void_p()
        genome = np.zeros(p.n_genomes + p.human_contigs, dtype=np.int32)
        self.check(lib().mm_synth_community(self.h, C.byref(p), C.byref(h), _ptr(genome)))
        return SeqSet(self, h), genome

    @staticmethod
    def synth_community_species(**kw) -> np.ndarray:
        """genome -> species of the community synth_community(**kw) generates"""
        p = SynthCommunityParams(**kw)
        sp = np.zeros(p.n_genomes, dtype=np.int32)
        st = lib().mm_synth_community_species(C.byref(p), _ptr(sp))
        if st != 0:
            raise MMError(st, "mm_synth_community_species: bad parameters")
        return sp

    def synth_reads(self, ref: "SeqSet", **kw):
        p = SynthReadParams(**kw)
        h = C.c_void_p()
        truth = np.zeros(p.n_reads, dtype=np.int32)
        self.check(lib().mm_synth_reads(self.h, ref.h, C.byref(p), C.byref(h), _ptr(truth)))
        return SeqSet(self, h), truth

    def minimizers(self, s: "SeqSet", k: int, w: int):
        n = s.count
        off = np.zeros(n + 1, dtype=np.int64)
        self.check(lib().mm_minimizers(self.h, s.h, k, w, _ptr(off), None, None, None, 0))
        tot = int(off[-1])
        hsh = np.zeros(tot, dtype=np.uint32)
        wp = np.zeros(tot, dtype=np.int32)
        st = np.zeros(tot, dtype=np.int32)
        self.check(lib().mm_minimizers(self.h, s.h, k, w, _ptr(off), _ptr(hsh), _ptr(wp), _ptr(st), tot))
        return off, hsh, wp, st

    def index(self, contigs: "SeqSet", k: int, w: int, auto_threshold: bool = True) -> "Index":
        h = C.c_void_p()
        self.check(lib().mm_index_build(self.h, contigs.h, k, w, C.byref(h)))
        idx = Index(self, h)
        if auto_threshold:
            counts, nh = idx.freq_hist()
            thr = lib().mm_freq_threshold_from_hist(_ptr(counts), _ptr(nh), len(counts), idx.info()["n_unique_hashes"], 2**31 - 1)
            idx.set_freq_threshold(thr)
        return idx

    def load_index(self, path: str) -> "Index":
        """the persistent device index Index.save wrote (mm_index_load): no kernel runs, the stored freqThreshold is in force"""
        h = C.c_void_p()
        self.check(lib().mm_index_load(self.h, path.encode(), C.byref(h)))
        return Index(self, h)

    def map_batch(self, idx: "Index", reads: "SeqSet", k: int, w: int, pi: float = 80.0, min_read_len: int = 1000, at_seed_stage=None, at_last_kernel=None,
                  sketch_of: "Mapping | None" = None) -> "Mapping":
        """at_seed_stage / at_last_kernel: callables run between the sketch stage and the seed stage / once K5 is enqueued (mm_map_batch_phased);
        sketch_of: a mapping of the same reads (or their sketch_batch) whose minimizers and sketches are taken instead of recomputed (mm_map_batch_reusing)"""
        p = MapParams(k, w, pi, min_read_len)
        h = C.c_void_p()
        if sketch_of is not None:
            self.check(lib().mm_map_batch_reusing(self.h, idx.h, reads.h, C.byref(p), sketch_of.h, C.byref(h)))
            return Mapping(self, h, reads.count)
        if at_seed_stage is None and at_last_kernel is None:
            self.check(lib().mm_map_batch(self.h, idx.h, reads.h, C.byref(p), C.byref(h)))
        else:
            def _stage(_user, stage):
                f = at_seed_stage if stage == 1 else at_last_kernel
                if f is not None:
                    f()
            cb = SEED_STAGE_CB(_stage)
            self.check(lib().mm_map_batch_phased(self.h, idx.h, reads.h, C.byref(p), cb, None, C.byref(h)))
        return Mapping(self, h, reads.count)

    def sketch_batch(self, reads: "SeqSet", k: int, w: int, pi: float = 80.0, min_read_len: int = 1000) -> "Mapping":
        """minimizers and sketches of the reads alone (mm_sketch_batch): a `sketch_of` for map_batch against any index"""
        p = MapParams(k, w, pi, min_read_len)
        h = C.c_void_p()
        self.check(lib().mm_sketch_batch(self.h, reads.h, C.byref(p), C.byref(h)))
        return Mapping(self, h, reads.count)

    def em(self, read_off, taxon, mapq, inv_nloc, n_taxa: int) -> "EM":
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        taxon = np.ascontiguousarray(taxon, dtype=np.int32)
        mapq = np.ascontiguousarray(mapq, dtype=np.float64)
        inv_nloc = np.ascontiguousarray(inv_nloc, dtype=np.float64)
        h = C.c_void_p()
        self.check(lib().mm_em_create(self.h, len(read_off) - 1, _ptr(read_off), _ptr(taxon), _ptr(mapq), _ptr(inv_nloc), n_taxa, C.byref(h)))
        return EM(self, h, n_taxa, len(read_off) - 1, len(taxon))

    def em_from_mapping(self, mapping: "Mapping", contig_taxon, contig_len, n_taxa: int) -> "EM":
        """the EM problem of `classify` built on the device from the mapping's records (after add_qualities)"""
        contig_taxon = np.ascontiguousarray(contig_taxon, dtype=np.int32)
        contig_len = np.ascontiguousarray(contig_len, dtype=np.int32)
        h = C.c_void_p()
        self.check(lib().mm_em_create_from_mapping(self.h, mapping.h, _ptr(contig_taxon), _ptr(contig_len), len(contig_taxon), n_taxa, C.byref(h)))
        nr, ne, nt = C.c_int64(), C.c_int64(), C.c_int32()
        self.check(lib().mm_em_sizes(h, C.byref(nr), C.byref(ne), C.byref(nt)))
        return EM(self, h, n_taxa, nr.value, ne.value)

    # ---- communicator
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        st = lib().mm_comm_unique_id(buf)
        if st != 0:
            raise MMError(st, "mm_comm_unique_id failed")
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, nranks: int):
        self.check(lib().mm_comm_init(self.h, uid, rank, nranks))

    def comm_share(self, owner: "Context"):
        """use the communicator of another context of the same device (collectives then go out in the caller's order)"""
        self.check(lib().mm_comm_share(self.h, owner.h))

    def comm_info(self):
        """(ranks, this rank) as RCCL reports them for the context's communicator"""
        n, r = C.c_int(), C.c_int()
        self.check(lib().mm_comm_info(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def comm_allreduce(self, arr: np.ndarray):
        assert arr.dtype == np.float64 and arr.flags.c_contiguous
        self.check(lib().mm_comm_allreduce_f64(self.h, _ptr(arr), arr.size))


class SeqSet:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h

    @property
    def count(self) -> int:
        return int(lib().mm_seqset_count(self.h))

    @property
    def total_bases(self) -> int:
        return int(lib().mm_seqset_total_bases(self.h))

    def lengths(self) -> np.ndarray:
        a = np.zeros(self.count, dtype=np.int32)
        self.ctx.check(lib().mm_seqset_lengths(self.h, _ptr(a)))
        return a

    def slice(self, first: int, count: int) -> "SeqSet":
        """sequences [first, first + count) as a set of their own (device-side copy)"""
        h = C.c_void_p()
        self.ctx.check(lib().mm_seqset_slice(self.ctx.h, self.h, first, count, C.byref(h)))
        return SeqSet(self.ctx, h)

    @staticmethod
    def concat(ctx: "Context", parts: list) -> "SeqSet":
        arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
        h = C.c_void_p()
        ctx.check(lib().mm_seqset_concat(ctx.h, arr, len(parts), C.byref(h)))
        return SeqSet(ctx, h)

    def save(self, path: str):
        self.ctx.check(lib().mm_seqset_save(self.h, path.encode()))

    def fetch(self, i: int, length: int) -> bytes:
        buf = C.create_string_buffer(length + 1)
        self.ctx.check(lib().mm_seqset_fetch(self.h, i, buf, length))
        return buf.raw[:length]

    def fetch_range(self, first: int, count: int):
        """(ASCII of sequences [first, first + count) back to back as a uint8 array, their lengths)"""
        ln = self.lengths()[first:first + count]
        buf = np.empty(int(ln.sum()), dtype=np.uint8)
        self.ctx.check(lib().mm_seqset_fetch_range(self.h, first, count, buf.ctypes.data, len(buf)))
        return buf, ln

    def close(self):
        if self.h:
            lib().mm_seqset_destroy(self.h)
            self.h = None


class Index:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h
        self.freq_threshold = 2**31 - 1

    def info(self) -> dict:
        i = IndexInfo()
        self.ctx.check(lib().mm_index_get_info(self.h, C.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in i._fields_}

    def save(self, path: str):
        self.ctx.check(lib().mm_index_save(self.h, path.encode()))

    def freq_hist(self):
        n = C.c_int64()
        self.ctx.check(lib().mm_index_freq_hist(self.h, None, None, 0, C.byref(n)))
        counts = np.zeros(n.value, dtype=np.int64)
        nh = np.zeros(n.value, dtype=np.int64)
        self.ctx.check(lib().mm_index_freq_hist(self.h, _ptr(counts), _ptr(nh), n.value, C.byref(n)))
        return counts, nh

    def set_freq_threshold(self, thr: int):
        self.ctx.check(lib().mm_index_set_freq_threshold(self.h, int(thr)))
        self.freq_threshold = int(thr)

    def plan_chunks(self, max_memory_bytes: int) -> list[int]:
        """first contig of every --maxmemory chunk (this index must cover the whole reference)"""
        n = C.c_int32()
        self.ctx.check(lib().mm_index_plan_chunks(self.ctx.h, self.h, int(max_memory_bytes), None, 0, C.byref(n)))
        fc = np.zeros(n.value, dtype=np.int32)
        self.ctx.check(lib().mm_index_plan_chunks(self.ctx.h, self.h, int(max_memory_bytes), _ptr(fc), n.value, C.byref(n)))
        return [int(x) for x in fc]

    def entries(self):
        n = self.info()["n_entries"]
        hsh = np.zeros(n, dtype=np.uint32)
        ct = np.zeros(n, dtype=np.int32)
        wp = np.zeros(n, dtype=np.int32)
        st = np.zeros(n, dtype=np.int32)
        self.ctx.check(lib().mm_index_entries(self.h, _ptr(hsh), _ptr(ct), _ptr(wp), _ptr(st), n))
        return hsh, ct, wp, st

    def dup_neighbours(self):
        """per entry: distance to the previous / next entry of its contig with the same hash (0 = none), as K5 reads them"""
        n = self.info()["n_entries"]
        pd = np.zeros(n, dtype=np.int32)
        nd = np.zeros(n, dtype=np.int32)
        self.ctx.check(lib().mm_index_dup_neighbours(self.h, _ptr(pd), _ptr(nd), n))
        return pd, nd

    def close(self):
        if self.h:
            lib().mm_index_destroy(self.h)
            self.h = None


class Mapping:
    def __init__(self, ctx: Context, h, n_reads: int):
        self.ctx, self.h, self.n_reads = ctx, h, n_reads

    def stats(self) -> dict:
        s = MapStats()
        self.ctx.check(lib().mm_mapping_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def add_qualities(self, k: int):
        self.ctx.check(lib().mm_mapping_add_qualities(self.ctx.h, self.h, None, k))

    def release_intermediates(self):
        """frees everything but the records (minimizers, sketches, hits, candidates): what a chunk mapping keeps while the other chunks are mapped"""
        self.ctx.check(lib().mm_mapping_release_intermediates(self.h))

    def keep_best(self, k: int):
        """default (non --all) reporting: per read keep identity >= best - 1.0"""
        self.ctx.check(lib().mm_mapping_keep_best(self.ctx.h, self.h, k))

    @staticmethod
    def concat(ctx: "Context", parts: list["Mapping"], contig_base: list[int]) -> "Mapping":
        arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
        base = np.asarray(contig_base, dtype=np.int32)
        out = C.c_void_p()
        ctx.check(lib().mm_mapping_concat(ctx.h, arr, _ptr(base), len(parts), C.byref(out)))
        return Mapping(ctx, out, parts[0].n_reads)

    @staticmethod
    def from_parts(ctx: "Context", read_len, parts: list, contig_base: list[int], k: int, w: int, pi: float = 80.0, min_read_len: int = 1000) -> "Mapping":
        """parts = [(offsets, records)] as returned by fetch() for index chunks mapped on other GPUs"""
        rl = np.ascontiguousarray(read_len, dtype=np.int32)
        offs = [np.ascontiguousarray(o, dtype=np.int64) for o, _ in parts]
        recs = [np.ascontiguousarray(r, dtype=RECORD_DTYPE) for _, r in parts]
        oarr = (C.c_void_p * len(parts))(*[o.ctypes.data for o in offs])
        rarr = (C.c_void_p * len(parts))(*[(r.ctypes.data if len(r) else None) for r in recs])
        base = np.asarray(contig_base, dtype=np.int32)
        mp = MapParams(k, w, pi, min_read_len)
        out = C.c_void_p()
        ctx.check(lib().mm_mapping_from_parts(ctx.h, len(rl), _ptr(rl), C.byref(mp), len(parts), oarr, rarr, _ptr(base), C.byref(out)))
        return Mapping(ctx, out, len(rl))

    @staticmethod
    def gather(ctx: "Context", owner: int, read_len, parts: list["Mapping"], chunk_id: list[int], chunk_rank: list[int], contig_base: list[int],
               k: int, w: int, pi: float = 80.0, min_read_len: int = 1000):
        """mm_mapping_gather: this rank's chunk mappings of one batch -> the merged mapping on `owner` (None on the other ranks); collective"""
        rl = np.ascontiguousarray(read_len, dtype=np.int32)
        arr = (C.c_void_p * max(len(parts), 1))(*[p.h for p in parts])
        cid = np.asarray(chunk_id, dtype=np.int32); crk = np.asarray(chunk_rank, dtype=np.int32); base = np.asarray(contig_base, dtype=np.int32)
        mp = MapParams(k, w, pi, min_read_len)
        out = C.c_void_p()
        ctx.check(lib().mm_mapping_gather(ctx.h, owner, len(rl), _ptr(rl), C.byref(mp), arr, _ptr(cid), len(parts), len(crk), _ptr(crk), _ptr(base), C.byref(out)))
        return Mapping(ctx, out, len(rl)) if out.value else None

    def fetch(self, rec_buf: np.ndarray | None = None):
        """offsets [n_reads+1] and the mm_map_record array.  `rec_buf` (RECORD_DTYPE, any capacity) lets a caller
        that maps batch after batch reuse one host buffer instead of faulting in fresh pages every time."""
        off = np.empty(self.n_reads + 1, dtype=np.int64)
        self.ctx.check(lib().mm_mapping_fetch(self.h, _ptr(off), None, 0))
        n = int(off[-1])
        rec = rec_buf[:n] if rec_buf is not None and len(rec_buf) >= n else np.empty(n, dtype=RECORD_DTYPE)
        self.ctx.check(lib().mm_mapping_fetch(self.h, _ptr(off), _ptr(rec), n))
        return off, rec

    def debug_sketch(self):
        off = np.zeros(self.n_reads + 1, dtype=np.int64)
        self.ctx.check(lib().mm_debug_sketch(self.h, _ptr(off), None, None, 0))
        h = np.zeros(int(off[-1]), dtype=np.uint32)
        s = np.zeros(int(off[-1]), dtype=np.int32)
        self.ctx.check(lib().mm_debug_sketch(self.h, _ptr(off), _ptr(h), _ptr(s), len(h)))
        return off, h, s

    def debug_hits(self):
        off = np.zeros(self.n_reads + 1, dtype=np.int64)
        self.ctx.check(lib().mm_debug_hits(self.h, _ptr(off), None, None, 0))
        c = np.zeros(int(off[-1]), dtype=np.int32)
        w = np.zeros(int(off[-1]), dtype=np.int32)
        self.ctx.check(lib().mm_debug_hits(self.h, _ptr(off), _ptr(c), _ptr(w), len(c)))
        return off, c, w

    def debug_candidates(self):
        off = np.zeros(self.n_reads + 1, dtype=np.int64)
        self.ctx.check(lib().mm_debug_candidates(self.h, _ptr(off), None, 0))
        t = np.zeros((int(off[-1]), 3), dtype=np.int32)
        self.ctx.check(lib().mm_debug_candidates(self.h, _ptr(off), _ptr(t), len(t)))
        return off, t

    def debug_l2(self, n_cand: int):
        a = np.zeros((n_cand, 6), dtype=np.int64)
        if n_cand:
            self.ctx.check(lib().mm_debug_l2(self.h, _ptr(a), n_cand))
        return a

    def debug_min_hits(self):
        a = np.zeros(self.n_reads, dtype=np.int32)
        self.ctx.check(lib().mm_debug_min_hits(self.h, _ptr(a)))
        return a

    def debug_probed_lists(self, idx: "Index", n_bins: int = 66) -> np.ndarray:
        a = np.zeros(n_bins, dtype=np.int64)
        self.ctx.check(lib().mm_debug_probed_lists(self.h, idx.h, _ptr(a), n_bins))
        return a

    def close(self):
        if self.h:
            lib().mm_mapping_destroy(self.h)
            self.h = None


class EM:
    def __init__(self, ctx: Context, h, n_taxa: int, n_reads: int, n_entries: int):
        self.ctx, self.h, self.n_taxa, self.n_reads, self.n_entries = ctx, h, n_taxa, n_reads, n_entries

    def iterate(self, f: np.ndarray):
        f = np.ascontiguousarray(f, dtype=np.float64)
        part = np.zeros(self.n_taxa, dtype=np.float64)
        ll = C.c_double()
        self.ctx.check(lib().mm_em_iterate(self.h, _ptr(f), _ptr(part), C.byref(ll)))
        return part, ll.value

    def iterate_allreduce(self, f: np.ndarray):
        f = np.ascontiguousarray(f, dtype=np.float64)
        nxt = np.zeros(self.n_taxa, dtype=np.float64)
        ll = C.c_double()
        self.ctx.check(lib().mm_em_iterate_allreduce(self.h, _ptr(f), _ptr(nxt), C.byref(ll)))
        return nxt, ll.value

    def run(self, f0: np.ndarray, max_iter: int = 1000):
        """the whole EM loop on the device: (final f, log-likelihood of every iteration)"""
        f0 = np.ascontiguousarray(f0, dtype=np.float64)
        f = np.zeros(self.n_taxa, dtype=np.float64)
        ll = np.zeros(1024, dtype=np.float64)
        n = C.c_int()
        self.ctx.check(lib().mm_em_run(self.h, _ptr(f0), max_iter, _ptr(f), _ptr(ll), len(ll), C.byref(n)))
        return f, ll[:min(n.value, len(ll))]

    def continue_run(self, max_iter: int = 1000):
        """up to max_iter more iterations from where run() / continue_run() stopped: (f, their log-likelihoods, stop rule fired)"""
        f = np.zeros(self.n_taxa, dtype=np.float64)
        ll = np.zeros(1024, dtype=np.float64)
        n, stopped = C.c_int(), C.c_int()
        self.ctx.check(lib().mm_em_continue(self.h, max_iter, _ptr(f), _ptr(ll), len(ll), C.byref(n), C.byref(stopped)))
        return f, ll[:min(n.value, len(ll))], bool(stopped.value)

    def taxon_counts(self) -> np.ndarray:
        c = np.zeros(self.n_taxa, dtype=np.int64)
        self.ctx.check(lib().mm_em_taxon_counts(self.h, _ptr(c)))
        return c

    def posteriors(self, f: np.ndarray):
        f = np.ascontiguousarray(f, dtype=np.float64)
        post = np.zeros(self.n_entries, dtype=np.float64)
        best = np.zeros(self.n_reads, dtype=np.int64)
        self.ctx.check(lib().mm_em_posteriors(self.h, _ptr(f), _ptr(post), _ptr(best)))
        return post, best

    def close(self):
        if self.h:
            lib().mm_em_destroy(self.h)
            self.h = None

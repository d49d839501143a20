"""Build time of an index against store / load time of its persistent form (mm_index_save / mm_index_load).
    python tools/index_persist.py [--scale 0.1] [--dir /tmp]
One JSON line: bases, entries, file bytes, build / save / load seconds (load twice: the second read comes from the page cache),
and whether the loaded index maps a read batch to the same records as the built one."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamaps_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.1, help="fraction of the bench's community reference (1.0 = 26.8 Gbp, 149 GB index)")
    ap.add_argument("--dir", default="/tmp", help="where the file goes (/dev/shm: memory-backed, what a 149 GB index needs on a box with a small disk)")
    ap.add_argument("--reads", type=int, default=20000)
    a = ap.parse_args()
    s = a.scale
    ctx = capi.Context(0)
    ng, sp, ge = max(4, int(12000 * s)), max(2, int(3000 * s)), max(1, int(600 * s))
    human = max(1, int(round(24 * min(s, 1.0)))) if s >= 0.04 else 0
    ref, _ = ctx.synth_community(seed=20260928, n_genomes=ng, n_species=sp, n_genera=ge, median_len=2.0e6, sigma_len=0.6, min_len=5_000, max_len=12_000_000,
                                 strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
                                 human_contigs=human, human_bases=int(3.1e9 * min(s, 1.0)), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000,
                                 total_bases_target=int(26_762_276_280 * s))
    out = {"reference_bases": ref.total_bases, "dir": a.dir}
    builds = []
    for _ in range(2):                                             # (the first build also pays the device's first big allocations)
        t = time.time(); idx = ctx.index(ref, 16, 8); builds.append(round(time.time() - t, 3))
        if _ == 0:
            idx.close()
    out["build_s"] = builds
    info = idx.info()
    out.update(entries=info["n_entries"], unique_hashes=info["n_unique_hashes"], hbm_bytes=info["hbm_bytes"])
    path = os.path.join(a.dir, "persist_test.mmidx")
    try:
        t = time.time(); idx.save(path); out["save_s"] = round(time.time() - t, 3)
    except Exception:
        if os.path.exists(path):
            os.remove(path)                                        # (a partial file of 100+ GB fills the disk for everything that follows)
        raise
    out["file_bytes"] = os.path.getsize(path)
    reads = ctx.synth_reads(ref, seed=5, n_reads=a.reads, read_len=10000, sub_rate=0.03, ins_rate=0.03, del_rate=0.03, frac_random=0.02, n_abundant=200, read_len_min=0)[0]
    Mb = ctx.map_batch(idx, reads, 16, 8)
    ob, rb = Mb.fetch()
    Mb.close(); idx.close()
    loads = []
    for _ in range(2):
        t = time.time(); L = ctx.load_index(path); loads.append(round(time.time() - t, 3))
        if _ == 0:
            L.close()
    out["load_s"] = loads
    out["load_GBps"] = [round(out["file_bytes"] / x / 1e9, 2) for x in loads]
    L.set_freq_threshold(idx.freq_threshold)
    Ml = ctx.map_batch(L, reads, 16, 8)
    ol, rl = Ml.fetch()
    out["same_records"] = bool((ob == ol).all() and rb.tobytes() == rl.tobytes())
    out["n_records"] = int(len(rb))
    os.remove(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

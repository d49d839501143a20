"""The zone kernel against l2_kernel (MM_L2_V1=1) on bench batches at full size (10^5 x 10 kb reads vs the 26.8 Gbp community): records must be byte-identical.
   python tools/zone_vs_v1_bench_scale.py [BATCH ...]   (defaults 0 10)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metamaps_amd import capi

which = [int(x) for x in sys.argv[1:]] or [0, 10]
sys.argv = ["bench.py"]
args = bench.parse_args()
ctx = capi.Context(0)
ref, contig_taxon, n_taxa, desc = bench.build_reference(ctx, args, "community")
idx = ctx.index(ref, 16, 8)
for b in which:
    rd, _ = ctx.synth_reads(ref, seed=1000 + 97 * b, n_reads=100000, read_len=10000, read_len_min=0, frac_random=0.05, n_abundant=100, sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
    out = {}
    for mode, env in (("zone", {}), ("own_ranges", {"MM_L2_NO_RANGES": "1"}), ("v1", {"MM_L2_V1": "1"})):
        for k_, v_ in env.items(): os.environ[k_] = v_
        M = ctx.map_batch(idx, rd, 16, 8, pi=80.0, min_read_len=1000)
        off, rec = M.fetch(); st = M.stats(); M.close()
        out[mode] = (off.copy(), rec.tobytes(), st["ms_l2"])
        for k_ in env: os.environ.pop(k_)
    same = all(np.array_equal(out["v1"][0], out[m][0]) and out["v1"][1] == out[m][1] for m in ("zone", "own_ranges"))
    print(f"batch {b}: {len(out['v1'][0]) - 1} reads, {out['v1'][0][-1]} records, identical {same}; K5 ms zone {out['zone'][2]:.1f} own ranges {out['own_ranges'][2]:.1f} l2_kernel {out['v1'][2]:.1f}", flush=True)
    assert same
    rd.close()

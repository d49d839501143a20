#!/usr/bin/env python3
"""bench.py — Gbp of long reads mapped + classified per second (BASELINE.json metric).

One "step" = one pass of the whole hot path over one batch of synthetic reads that is already resident
in HBM (packed 2-bit), against a reference index that is already resident in HBM:

    K1 minimizers → K2 sketch → K3 probe/gather → K4 sort + L1 → K5/K6 L2 + strand → K8 mapping qualities
    → records to the host → EM iterations (K9, RCCL all-reduce of the per-taxon sums when N>1) → posteriors

Index construction (which the reference redoes on every run, mapWrap.h:432) happens once, untimed, in the
setup, and is reported separately in `config.index_build_s`.

Multi-GPU: one process per GPU (torch.distributed launch), index replicated, every rank maps its own
`--reads` reads (weak scaling: per-GPU work fixed), EM sufficient statistics all-reduced over RCCL.
Timing: barrier + synchronize on both sides of exactly K steps, MAX over ranks; rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    # workload: BASELINE configs[1] shape; scale knobs exist so that smaller boxes / quick checks can run
    ap.add_argument("--reads", type=int, default=int(os.environ.get("MM_BENCH_READS", 100_000)), help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--read-len-min", type=int, default=0, help="mixed lengths, log-uniform in [read-len-min, read-len] (BASELINE config 3 shape); 0 = fixed")
    ap.add_argument("--species", type=int, default=int(os.environ.get("MM_BENCH_SPECIES", 3000)))
    ap.add_argument("--strains", type=int, default=int(os.environ.get("MM_BENCH_STRAINS", 4)))
    ap.add_argument("--genome-len", type=int, default=int(os.environ.get("MM_BENCH_GENOME_LEN", 2_200_000)))
    ap.add_argument("--window", type=int, default=8, help="w the CLI derives for a 26.76 GB DB.fa at default flags")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=20000)
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads for cpu_baseline (0 = all host cores)")
    ap.add_argument("--cpu-sample-genomes", type=int, default=24)
    return ap.parse_args()


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
    r[r < 2.2250738585072014e-308] = 0.0
    out[nz] = r
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from metamaps_amd import capi, emhost
    ctx = capi.Context(local)
    # the EM all-reduce always goes through the RCCL communicator, also with one rank (the same code path at every N)
    uid = [capi.Context.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    ctx.comm_init(uid[0], rank, world)
    import ctypes
    ctypes.CDLL(None).fflush(None)                                # RCCL prints a version banner through C stdio: out before the JSON line

    k, w = 16, args.window
    G = args.species * args.strains
    # ---------------- setup (untimed): reference + index + reads, all generated on the device ----------------
    t0 = time.time()
    ref = ctx.synth_reference(seed=20260928, n_species=args.species, strains_per_species=args.strains,
                              genome_len=args.genome_len, strain_divergence=0.02, genus_divergence=0.2)
    ctx.synchronize()
    t_ref = time.time() - t0
    t0 = time.time()
    idx = ctx.index(ref, k, w)
    ctx.synchronize()
    t_index = time.time() - t0
    info = idx.info()
    reads, truth = ctx.synth_reads(ref, seed=1000 + rank, n_reads=args.reads, read_len=args.read_len, read_len_min=args.read_len_min,
                                   sub_rate=0.04, ins_rate=0.03, del_rate=0.05, frac_random=0.05, n_abundant=100)
    read_len = reads.lengths().astype(np.int64)
    ctx.synchronize()

    contig_taxon = np.arange(G, dtype=np.int32)                  # one contig per genome = one taxon per contig
    contig_len = ref.lengths().astype(np.int32)
    agg = {"ms_l2": 0.0, "ms_hf": 0.0, "l2_launches": 0, "l2_stream": 0, "hf_hits": 0, "stats": None, "em_iters": 0}
    rec_buf = np.empty(max(64 * args.reads, 1 << 16), dtype=capi.RECORD_DTYPE)   # host result buffer reused by every step

    def step():
        tt = [time.perf_counter()]
        M = ctx.map_batch(idx, reads, k, w, pi=80.0, min_read_len=1000)
        tt.append(time.perf_counter())
        M.add_qualities(k)
        off, rec = M.fetch(rec_buf)
        st = M.stats()
        tt.append(time.perf_counter())
        # ---- classify: the EM problem built on the device from the records (fEM.h:234-373), then device iterations
        em = ctx.em_from_mapping(M, contig_taxon, contig_len, G)
        M.close()
        seen = (em.taxon_counts() > 0).astype(np.float64)
        ctx.comm_allreduce(seen)
        present = seen > 0
        n_seen = int(present.sum())
        f0 = np.where(present, 1.0 / max(n_seen, 1), 0.0)

        def em_step(f):
            return em.iterate_allreduce(f)

        f = f0
        lls, ll_prev = [], 0.0
        for it in range(1000):
            f_next, ll = em_step(f)
            lls.append(ll)
            stop = it > 0 and (ll - ll_prev) <= 1 and (1 - ll / ll_prev) < 1e-4
            f, ll_prev = f_next, ll
            if stop:
                break
        tt.append(time.perf_counter())
        post, best = em.posteriors(f)
        em.close()
        tt.append(time.perf_counter())
        agg["host_ms"] = {"map_batch": (tt[1] - tt[0]) * 1e3, "mapq_fetch": (tt[2] - tt[1]) * 1e3, "em_prepare_iterate": (tt[3] - tt[2]) * 1e3,
                          "posteriors": (tt[4] - tt[3]) * 1e3}
        agg["ms_l2"] += st["ms_l2"]; agg["l2_launches"] += 1; agg["l2_stream"] += st["sum_l2_stream_entries"]
        agg["ms_hf"] += st["ms_hit_filter"]; agg["hf_hits"] += st["sum_hits"]
        agg["stats"] = st; agg["em_iters"] = len(lls)
        return st, f, best

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        step()
    agg.update({"ms_l2": 0.0, "ms_hf": 0.0, "l2_launches": 0, "l2_stream": 0, "hf_hits": 0})
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, f, best = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        bb = torch.tensor([float(st["bases_long_enough"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(bb, op=dist.ReduceOp.SUM)
        bases_all = float(bb.item())
    else:
        bases_all = float(st["bases_long_enough"])

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = bases_all * args.steps / dt / 1e9
        # roofline of the dominant kernel — whichever of the two big kernels took longer per launch (hipEvents on the ctx
        # stream around each).  Algorithmic bytes per launch, SURVEY.md §8 D3:
        #   K5/K6  l2_kernel                  8 B per streamed index entry   (8·Σ_c M_{r,c})
        #   K3c    hit_filter_kernel<false>   8 B per seed hit               (8·H_r)
        nl = max(agg["l2_launches"], 1)
        cands = [("l2_kernel", 8.0 * agg["l2_stream"] / nl, agg["ms_l2"] / nl),
                 ("hit_filter_kernel<false>", 8.0 * agg["hf_hits"] / nl, agg["ms_hf"] / nl)]
        dom_name, dom_bytes, dom_ms = max(cands, key=lambda c: c[2])
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        len_txt = f"{args.read_len}" if not args.read_len_min else f"{args.read_len_min}-{args.read_len}"
        out_workload = (f"{args.reads} synthetic {len_txt} bp ONT-error reads per GPU vs synthetic miniSeq+H-shaped index "
                        f"({args.species} species x {args.strains} strains x {args.genome_len} bp = {G * args.genome_len / 1e9:.2f} Gbp), k=16 w={w}, --all")
        out = {
            "metric": "Gbp long reads mapped+classified per sec (whole node), miniSeq+H DB",
            "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": out_workload,
                "reads_per_gpu": args.reads, "read_len": args.read_len, "reference_bp": G * args.genome_len,
                "index_entries": info["n_entries"], "index_unique_hashes": info["n_unique_hashes"], "index_hbm_bytes": info["hbm_bytes"],
                "freq_threshold": idx.freq_threshold, "reference_synth_s": round(t_ref, 3), "index_build_s": round(t_index, 3),
                "parallelism": f"reads sharded x{world}, index replicated, RCCL all-reduce of EM sums",
                "em_iterations": agg["em_iters"],
                "per_step": {kk: st[kk] for kk in ("n_reads_long_enough", "n_reads_mapped", "n_mappings", "sum_sketch", "sum_hits",
                                                   "n_candidates", "sum_l2_stream_entries", "sum_l2_evals", "n_ambiguous_sketch_reads",
                                                   "sum_hits_kept", "n_l2_rebuilds", "n_l2_wide_redo")},
                "stage_ms": {kk: round(st[kk], 3) for kk in st if kk.startswith("ms_")},
                "host_wall_ms": {kk: round(v, 3) for kk, v in agg.get("host_ms", {}).items()},
            },
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(out_workload, dom_name),
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": dom_ms,
                         "other_kernels": {n: {"ms_per_launch": m, "algorithmic_bytes_per_launch": b, "achieved": (b / (m * 1e-3) / 1e9 if m > 0 else 0.0)}
                                           for n, b, m in cands if n != dom_name}},
        }
        if not args.no_cpu_baseline and world == 1:              # (the contract asks for it at N=1 only)
            try:
                out["cpu_baseline"] = cpu_baseline(args, ctx, ref, reads, truth, k, w)
            except Exception as e:  # the baseline is a reported side number; never let it kill the bench line
                out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def measured_traffic(workload: str, kernel: str = "l2_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    collected and corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes; profiles/r01_pmc_hbm_traffic.txt).
    bench.py cannot run the profiler itself, so the number is reported only for the workload it was measured on."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if t.get("workload") != workload:
            return None
        for name, v in t.get("by_kernel", {}).items():
            if kernel.split("<")[0] in name and ("<false>" in name) == ("<false>" in kernel):
                return v
        return t["traffic_bytes_per_launch"] if kernel == t.get("kernel") else None
    except Exception:
        return None


def cpu_baseline(args, ctx, ref, reads, truth, k, w):
    """Time the oracle (CPU restatement of the reference, -t <all host cores>) on a bounded sample of the same
    workload: the genomes the sampled reads come from plus fillers, and `cpu_sample_reads` reads.  The full
    index is far beyond a CPU budget of seconds (the reference indexes ~2 Mbp/s), so the sample DB is small;
    mapping time per read on it is a LOWER bound of what the full DB would cost the CPU."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    exe = os.path.join(ROOT, "oracle", "_build", "metamaps_oracle")
    rl = reads.lengths()
    # reads of the first `cpu_sample_genomes` source genomes met in read order, so that the sample DB stays small
    # (the oracle indexes ~3 Mbp/s single-threaded) while the read sample is large enough to keep every core busy
    allowed, pick_reads = [], []
    for r in range(len(rl)):
        t = int(truth[r])
        if t < 0:
            continue
        if t not in allowed:
            if len(allowed) >= args.cpu_sample_genomes:
                continue
            allowed.append(t)
        pick_reads.append(r)
        if len(pick_reads) >= args.cpu_sample_reads:
            break
    genomes = sorted(allowed)
    g = 0
    while len(genomes) < args.cpu_sample_genomes:
        if g not in genomes:
            genomes.append(g)
        g += 1
    glen = int(ref.lengths()[0])
    with tempfile.TemporaryDirectory() as d:
        fa, fq = os.path.join(d, "DB.fa"), os.path.join(d, "reads.fq")
        with open(fa, "wb") as f:
            for gi in genomes:
                f.write(f">C{gi}|kraken:taxid|{gi + 1}|SYN{gi}\n".encode() + ref.fetch(gi, glen) + b"\n")
        nb = 0
        with open(fq, "wb") as f:
            for r in pick_reads:
                s = reads.fetch(r, int(rl[r])); nb += len(s)
                f.write(f"@r{r}\n".encode() + s + b"\n+\n" + b"I" * len(s) + b"\n")
        # the GPU boxes show 256 cores but the oracle stops scaling at ~16 threads there (tools/cpu_scaling.sh: 0.072 Gbp/s at 16,
        # 0.077 at 64, 0.061 at 256), so more than 64 threads only adds scheduling noise
        cores = args.cpu_threads or min(len(os.sched_getaffinity(0)), 64)
        p = subprocess.run([exe, "mapDirectly", "--all", "-r", fa, "-q", fq, "-o", os.path.join(d, "out"), "-w", str(w), "-t", str(cores)],
                           capture_output=True, check=True, timeout=1200)
        js = json.loads(p.stderr.decode().strip().splitlines()[-1])
    return {"value": js["bases"] / js["map_seconds"] / 1e9, "unit": "Gbp/s", "cores": cores, "kind": "port",
            "sample": f"{len(pick_reads)} of the bench reads ({js['bases']} bp) vs a {len(genomes)}-genome slice of the bench reference "
                      f"({len(genomes) * glen / 1e6:.1f} Mbp), oracle mapping phase only, -t {cores} ({js['map_seconds']:.2f} s; single-threaded index build "
                      f"{js['seconds'] - js['map_seconds']:.2f} s excluded), classify excluded",
            "mappings": js["mappings"], "map_seconds": js["map_seconds"]}


if __name__ == "__main__":
    main()

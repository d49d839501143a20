#!/usr/bin/env python3
"""bench.py — Gbp of long reads mapped + classified per second (BASELINE.json metric).

One "step" = one pass of the whole hot path over one batch of synthetic reads that is already resident
in HBM (packed 2-bit), against a reference index that is already resident in HBM:

    K1 minimizers → K2 sketch → K3 probe + seed-hit filter → K4 sort + L1 → K5/K6 L2 + strand → K8 mapping qualities
    → records to the host → EM iterations (K9, RCCL all-reduce of the per-taxon sums) → posteriors

Index construction (which the reference redoes on every run, mapWrap.h:432) happens once, untimed, in the
setup, and is reported separately in `config.index_build_s`.

Workload (BASELINE configs[1]): 100 000 × 10 kb ONT-error reads against a miniSeq+H-shaped reference.  Two shapes of that
reference are generated on the device (csrc/mm_synth.hip):
  community  (default, `value` is measured on it)  SURVEY.md §8 D1: 12 000 microbial genomes of lognormal length in 3 000
             species of 1–12 strains (0.1–5 % substitutions + block indels), 600 genera, plus 24 human-like contigs (3.1 Gbp,
             45 % interspersed library repeats, 1 % N runs), contig order shuffled, 26.76 Gbp in all (the size at which the CLI
             derives w = 8)
  uniform    round 1's shape: 3 000 species × 4 strains × 2.2 Mbp, substitutions only (reported beside it in config.other_shape)

Multi-GPU: one process per GPU (torch.distributed launch), index replicated, every rank maps its own
`--reads` reads (weak scaling: per-GPU work fixed), EM sufficient statistics all-reduced over RCCL (also at N = 1: the same
code path at every N).  Timing: barrier + synchronize on both sides of exactly K steps, MAX over ranks; rank 0 prints one JSON line.
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
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r02_traffic.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    # workload: BASELINE configs[1] shape; scale knobs exist so that smaller boxes / quick checks can run
    ap.add_argument("--shape", choices=("community", "uniform"), default=os.environ.get("MM_BENCH_SHAPE", "community"))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("MM_BENCH_READS", 100_000)), help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--read-len-min", type=int, default=0, help="mixed lengths, log-uniform in [read-len-min, read-len] (BASELINE config 3 shape); 0 = fixed")
    ap.add_argument("--pacbio", action="store_true", help="PacBio-like errors (2/8/2 percent del/ins/sub) instead of ONT-like (5/3/4)")
    ap.add_argument("--scale", type=float, default=float(os.environ.get("MM_BENCH_SCALE", 1.0)), help="scales the number of genomes of the reference (quick checks)")
    ap.add_argument("--window", type=int, default=8, help="w the CLI derives for a 26.76 GB DB.fa at default flags")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("MM_BENCH_WORKERS", 3)),
                    help="host threads / contexts per GPU that take the steps in turn: while one runs the EM iterations, result download and host "
                         "bookkeeping of its step, the next step's mapping kernels run (the mapping sections themselves are serialised, so that "
                         "kernel durations — the roofline — are those of kernels that own the GPU)")
    ap.add_argument("--staged-map", action="store_true", help="two locks instead of one around the mapping section (K1 + K2 | K3 ... K6, swapped at mm_map_batch_phased's "
                    "callback): the next step's minimizer stage runs under this step's seed stage; ~2 percent more throughput, but the seed filter's "
                    "duration then includes the time it shares the CUs (K1 issues VALU instructions in 99 percent of its cycles: the two do not complement each other)")
    ap.add_argument("--hold-lock-to-the-end", action="store_true", help="release the mapping lock when mm_map_batch returns instead of when its last big kernel is enqueued")
    ap.add_argument("--free-overlap", action="store_true", help="do not serialise the mapping sections of the workers (higher throughput, kernel durations inflated)")
    ap.add_argument("--measure-free-overlap", action="store_true", help="after the timed region, six more steps with nothing serialised, reported in config.free_overlap")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-shape", action="store_true", help="skip the second reference shape (config.other_shape)")
    ap.add_argument("--cpu-sample-reads", type=int, default=20000)
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads for cpu_baseline (0 = all host cores, at most 64)")
    ap.add_argument("--cpu-sample-genomes", type=int, default=100)
    return ap.parse_args()


def build_reference(ctx, args, shape):
    """(reference seqset, contig -> taxon, number of taxa, description) of one reference shape, generated on the device"""
    if shape == "uniform":
        species, strains, glen = max(1, int(3000 * args.scale)), 4, 2_200_000
        ref = ctx.synth_reference(seed=20260928, n_species=species, strains_per_species=strains, genome_len=glen, strain_divergence=0.02, genus_divergence=0.2)
        G = species * strains
        return ref, np.arange(G, dtype=np.int32), G, f"uniform: {species} species x {strains} strains x {glen} bp, substitutions only"
    ng, sp, ge = max(4, int(12000 * args.scale)), max(2, int(3000 * args.scale)), max(1, int(600 * args.scale))
    human = max(1, int(round(24 * min(args.scale, 1.0)))) if args.scale >= 0.04 else 0
    ref, genome = ctx.synth_community(seed=20260928, n_genomes=ng, n_species=sp, n_genera=ge, median_len=2.0e6, sigma_len=0.6, min_len=5_000, max_len=12_000_000,
                                      strain_div_min=0.001, strain_div_max=0.05, genus_div_min=0.15, genus_div_max=0.25, strain_indel_events=8,
                                      human_contigs=human, human_bases=int(3.1e9 * min(args.scale, 1.0)), repeat_fraction=0.45, n_fraction=0.01, n_repeat_families=1000,
                                      total_bases_target=int(26_762_276_280 * args.scale))
    desc = (f"community (SURVEY D1): {ng} microbial genomes in {sp} species / {ge} genera, lognormal lengths, 0.1-5 % strain divergence + block indels, "
            f"{human} human-like contigs ({int(3.1e9 * min(args.scale, 1.0))} bp, 45 % library repeats, 1 % N), contigs shuffled")
    return ref, genome.astype(np.int32), ng + 1, desc


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

    from metamaps_amd import capi
    import threading
    sys.setswitchinterval(1e-4)                                   # worker threads hand the GPU over at lock releases: do not let one sit on the interpreter
    W = max(1, args.workers)
    ctxs = [capi.Context(local) for _ in range(W)]               # one context (stream + allocator + communicator) per worker thread
    ctx = ctxs[0]
    # the EM all-reduce always goes through the RCCL communicator, also with one rank (the same code path at every N).  The worker
    # contexts of a rank share ONE communicator and run their classify sections in step order, so that every rank issues the same
    # sequence of collectives
    uid = [capi.Context.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    ctx.comm_init(uid[0], rank, world)
    for c in ctxs[1:]:
        c.comm_share(ctx)
    import ctypes
    ctypes.CDLL(None).fflush(None)                                # RCCL prints a version banner through C stdio: out before the JSON line

    k, w = 16, args.window
    err = dict(sub_rate=0.02, ins_rate=0.08, del_rate=0.02) if args.pacbio else dict(sub_rate=0.04, ins_rate=0.03, del_rate=0.05)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for c in ctxs:
            c.synchronize()

    def run_shape(shape, steps, warmup):
        """setup (untimed) + warm-up + `steps` timed steps on one reference shape; returns everything the report needs"""
        t0 = time.time()
        ref, contig_taxon, n_taxa, desc = build_reference(ctx, args, shape)
        ctx.synchronize()
        t_ref = time.time() - t0
        t0 = time.time()
        idx = ctx.index(ref, k, w)
        ctx.synchronize()
        t_index = time.time() - t0
        info = idx.info()
        reads_w, truth = [], None
        for c in ctxs:                                           # every worker context holds the batch (the same reads)
            rd, truth = c.synth_reads(ref, seed=1000 + rank, n_reads=args.reads, read_len=args.read_len, read_len_min=args.read_len_min,
                                      frac_random=0.05, n_abundant=100, **err)
            reads_w.append(rd)
        reads = reads_w[0]
        ctx.synchronize()
        contig_len = ref.lengths().astype(np.int32)
        agg = {"ms_l2": 0.0, "ms_hf": 0.0, "launches": 0, "l2_stream": 0, "hf_units": 0, "stats": None, "em_iters": 0}
        rec_bufs = [np.empty(max(64 * args.reads, 1 << 16), dtype=capi.RECORD_DTYPE) for _ in ctxs]   # host result buffers reused by every step
        map_lock, agg_lock = threading.Lock(), threading.Lock()
        front_lock, back_lock = threading.Lock(), threading.Lock()
        hold_lock = [bool(args.hold_lock_to_the_end)]
        em_turn = {"next": 0, "cv": threading.Condition()}

        def step(wi, serialise=True, ticket=0):
            """serialise: True (default) — one lock around the mapping section of a step, released when its last big kernel (K5) is
            enqueued (mm_map_batch_phased, stage 2), so that the next step's minimizer stage fills the CUs that kernel leaves as it drains;
            "staged" (--staged-map) — one lock for K1 + K2, one for K3 ... K6, swapped at stage 1; False: nothing serialised"""
            c = ctxs[wi]
            tt = [time.perf_counter()]
            staged = serialise == "staged"
            swapped = [False]
            released = [False]
            def swap():
                back_lock.acquire(); front_lock.release(); swapped[0] = True
            def release():                                          # the step's last big kernel is enqueued: the next step may start its minimizer stage,
                if released[0]: return                              # which takes the CUs that kernel leaves as it drains
                released[0] = True
                if staged: back_lock.release()
                elif serialise: map_lock.release()
            if staged:
                front_lock.acquire()
            elif serialise:
                map_lock.acquire()
            try:
                try:
                    M = c.map_batch(idx, reads_w[wi], k, w, pi=80.0, min_read_len=1000, at_seed_stage=swap if staged else None,
                                    at_last_kernel=release if (serialise and not hold_lock[0]) else None)
                finally:
                    if staged and not swapped[0]:
                        swap()
                tt.append(time.perf_counter())
            finally:
                release()
            M.add_qualities(k)                                      # (one 60 us kernel on this worker's stream: under the next step's minimizer stage)
            off, rec = M.fetch(rec_bufs[wi])
            st = M.stats()
            tt.append(time.perf_counter())
            # ---- classify: the EM problem built on the device from the records (fEM.h:234-373), then device iterations; the classify
            # sections of the workers run in step order (one communicator, the same order of collectives on every rank)
            em = c.em_from_mapping(M, contig_taxon, contig_len, n_taxa)
            M.close()
            with em_turn["cv"]:
                em_turn["cv"].wait_for(lambda: em_turn["next"] == ticket)
            try:
                seen = (em.taxon_counts() > 0).astype(np.float64)
                c.comm_allreduce(seen)
                present = seen > 0
                f = np.where(present, 1.0 / max(int(present.sum()), 1), 0.0)
                f, lls = em.run(f)                                  # the EM loop, device resident (fEM.h:501-661)
            finally:
                with em_turn["cv"]:
                    em_turn["next"] = ticket + 1
                    em_turn["cv"].notify_all()
            tt.append(time.perf_counter())
            post, best = em.posteriors(f)
            em.close()
            tt.append(time.perf_counter())
            with agg_lock:
                agg["host_ms"] = {"map_batch": (tt[1] - tt[0]) * 1e3, "mapq_fetch": (tt[2] - tt[1]) * 1e3, "em_prepare_iterate": (tt[3] - tt[2]) * 1e3,
                                  "posteriors": (tt[4] - tt[3]) * 1e3}
                agg["ms_l2"] += st["ms_l2"]; agg["launches"] += 1; agg["l2_stream"] += st["sum_l2_stream_entries"]
                agg["ms_hf"] += st["ms_hit_filter"]; agg["hf_units"] += st["sum_hits"] + st["sum_sketch"]
                agg["stats"] = st; agg["em_iters"] = len(lls)
            return st

        def run_steps(n, serialise=True):
            """n steps, taken in turn by the worker threads (every rank runs the same schedule, so the collectives of communicator i match)"""
            em_turn["next"] = 0
            def work(wi):
                for s_i in range(wi, n, W):
                    step(wi, serialise, s_i)
            th = [threading.Thread(target=work, args=(wi,)) for wi in range(1, W)]
            for t in th:
                t.start()
            work(0)
            for t in th:
                t.join()

        for wi in range(1, W):                                    # setup: every further worker context runs once (its scratch buffers get allocated)
            step(wi, True, 0); em_turn["next"] = 0
        sched = False if args.free_overlap else ("staged" if (args.staged_map and W > 1) else True)
        run_steps(max(warmup, 0), sched)
        agg.update({"ms_l2": 0.0, "ms_hf": 0.0, "launches": 0, "l2_stream": 0, "hf_units": 0})
        barrier()
        t0 = time.perf_counter()
        run_steps(steps, sched)
        barrier()
        dt = time.perf_counter() - t0
        st = agg["stats"]
        # stage times of a step whose kernels all own the GPU (the timed region lets the next step's K1 queue behind K5): two more steps, untimed
        st_clean = st
        if W > 1 and sched is True and not hold_lock[0]:
            keep = dict(agg)
            hold_lock[0] = True; run_steps(2, sched); hold_lock[0] = False
            st_clean = agg["stats"]
            agg.clear(); agg.update(keep)
        free = None
        if args.measure_free_overlap and W > 1 and not args.free_overlap and world == 1 and shape == args.shape:   # beside the headline: the same steps with nothing serialised
            keep = dict(agg)
            barrier(); t1 = time.perf_counter(); run_steps(6, False); barrier()
            d1 = time.perf_counter() - t1
            free = {"ms_per_step": d1 / 6 * 1e3, "value": float(st["bases_long_enough"]) * 6 / d1 / 1e9, "steps": 6}
            agg.clear(); agg.update(keep)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            bb = torch.tensor([float(st["bases_long_enough"])], dtype=torch.float64, device="cuda")
            dist.all_reduce(bb, op=dist.ReduceOp.SUM)
            bases_all = float(bb.item())
        else:
            bases_all = float(st["bases_long_enough"])
        return dict(ref=ref, idx=idx, reads=reads, reads_w=reads_w, truth=truth, contig_taxon=contig_taxon, info=info, desc=desc, t_ref=t_ref, t_index=t_index, free=free,
                    agg=agg, st=st, st_clean=st_clean, dt=dt, steps=steps, bases_all=bases_all, value=bases_all * steps / dt / 1e9, ms_step=dt / steps * 1e3,
                    freq_threshold=idx.freq_threshold, reference_bp=int(ref.total_bases))

    R = run_shape(args.shape, args.steps, args.warmup)

    out = None
    if rank == 0:
        agg, st, info = R["agg"], R["st"], R["info"]
        # roofline of the dominant kernel — whichever of the two big kernels took longer per launch (hipEvents on the ctx
        # stream around each).  Algorithmic bytes per launch, SURVEY.md §8 D3:
        #   K5/K6  l2_kernel            8 B per streamed index entry                          (8·Σ_c M_{r,c})
        #   K3     seed_filter_kernel   8 B per sketch hash probed + 8 B per seed hit         (8·s_r + 8·H_r)
        nl = max(agg["launches"], 1)
        cands = [("l2_kernel (launches of one step: <true,u8,4,2> + <true,u8,2,2>)", 8.0 * agg["l2_stream"] / nl, agg["ms_l2"] / nl),
                 ("seed_filter_kernel", 8.0 * agg["hf_units"] / nl, agg["ms_hf"] / nl)]
        dom_name, dom_bytes, dom_ms = max(cands, key=lambda c: c[2])
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        len_txt = f"{args.read_len}" if not args.read_len_min else f"{args.read_len_min}-{args.read_len}"
        out_workload = (f"{args.reads} synthetic {len_txt} bp {'PacBio' if args.pacbio else 'ONT'}-error reads per GPU vs synthetic miniSeq+H-shaped index "
                        f"({R['desc']}; {R['reference_bp'] / 1e9:.2f} Gbp), k=16 w={w}, --all")
        out = {
            "metric": "Gbp long reads mapped+classified per sec (whole node), miniSeq+H DB",
            "value": R["value"], "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": R["ms_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": out_workload,
                "reads_per_gpu": args.reads, "read_len": args.read_len, "reference_bp": R["reference_bp"], "reference_contigs": info["n_contigs"],
                "index_entries": info["n_entries"], "index_unique_hashes": info["n_unique_hashes"], "index_hbm_bytes": info["hbm_bytes"],
                "freq_threshold": R["freq_threshold"], "reference_synth_s": round(R["t_ref"], 3), "index_build_s": round(R["t_index"], 3),
                "parallelism": f"reads sharded x{world}, index replicated, RCCL all-reduce of EM sums; {W} worker contexts per GPU take the steps in turn"
                               + ("" if W == 1 or args.free_overlap else ((" (mapping sections serialised" + ("" if args.hold_lock_to_the_end else "; the lock passes on when a step's last big kernel, K5, is enqueued: the next step's minimizer "
                                   "kernel waits in its queue and takes the CUs K5 leaves as it drains — its stage time, ms_minimizer, then includes that wait") + ")") if not args.staged_map else
                                  " (the minimizer + sketch stage of step i+1 runs under the seed stage of step i; everything from the hit sort on owns the GPU)")),
                "workers_per_gpu": W, "free_overlap": R["free"],
                "em_iterations": agg["em_iters"],
                "per_step": {kk: st[kk] for kk in ("n_reads_long_enough", "n_reads_mapped", "n_mappings", "sum_sketch", "sum_hits",
                                                   "n_candidates", "sum_l2_stream_entries", "sum_l2_evals", "n_ambiguous_sketch_reads",
                                                   "sum_hits_kept", "n_l2_rebuilds", "n_l2_wide_redo")},
                "stage_ms": {kk: round(R["st_clean"][kk], 3) for kk in st if kk.startswith("ms_")},
                "stage_ms_note": "stage times of a step run after the timed region with the mapping lock held to the end of the step; in the timed region the next step's K1 "
                                 "queues behind K5 (config.parallelism), which shows up in that step's ms_minimizer / ms_total: stage_ms_timed_region",
                "stage_ms_timed_region": {kk: round(st[kk], 3) for kk in st if kk.startswith("ms_")},
                "host_wall_ms": {kk: round(v, 3) for kk, v in agg.get("host_ms", {}).items()},
            },
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args, dom_name),
                         "note": "l2_kernel uses 87 percent of its VALU issue slots (profiles/r02_sq_counters.txt): an integer-ALU-bound kernel, the HBM "
                                 "fraction says how far it sits from a bound it does not touch; seed_filter_kernel 40 percent, minimizer_kernel 99 percent",
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": dom_ms,
                         "other_kernels": {n: {"ms_per_launch": m, "algorithmic_bytes_per_launch": b, "achieved": (b / (m * 1e-3) / 1e9 if m > 0 else 0.0)}
                                           for n, b, m in cands if n != dom_name}},
        }
        if not args.no_cpu_baseline and world == 1:              # (the contract asks for it at N=1 only)
            try:
                out["cpu_baseline"], out["e2e_cli"] = cpu_baseline_and_cli(args, R, k, w)
            except Exception as e:  # the baseline is a reported side number; never let it kill the bench line
                out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    # the other reference shape, beside the headline (one GPU only: it costs a second index build)
    if world == 1 and not args.no_other_shape:
        for rd in R["reads_w"]:
            rd.close()
        for kk in ("idx", "ref"):
            R[kk].close()
        other = "uniform" if args.shape == "community" else "community"
        try:
            R2 = run_shape(other, max(3, min(args.steps, 5)), 1)
            out["config"]["other_shape"] = {"shape": R2["desc"], "value": R2["value"], "unit": "Gbp/s", "ms_per_step": R2["ms_step"], "steps": R2["steps"],
                                            "reference_bp": R2["reference_bp"], "freq_threshold": R2["freq_threshold"],
                                            "stage_ms": {kk: round(R2["st_clean"][kk], 3) for kk in R2["st"] if kk.startswith("ms_")},
                                            "per_step": {kk: R2["st"][kk] for kk in ("n_reads_mapped", "n_mappings", "sum_sketch", "sum_hits", "sum_hits_kept", "n_candidates", "sum_l2_stream_entries")}}
        except Exception as e:
            out["config"]["other_shape"] = {"failed": str(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


def measured_traffic(args, kernel: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    collected and corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes; profiles/r02_pmc_hbm_traffic.txt).
    bench.py cannot run the profiler itself, so the number is reported only for the workload it was measured on."""
    try:
        t = json.load(open(TRAFFIC_FILE))
        if t.get("shape") != args.shape or t.get("reads") != args.reads or t.get("read_len") != args.read_len or args.read_len_min or args.scale != 1.0:
            return None
        hits = [v for name, v in t.get("by_kernel", {}).items() if kernel.split(" ")[0] in name]   # (K5 runs as two launches per step: the
        return sum(hits) if hits else None                                                          #  4-wave and the 2-wave workgroup shape)
    except Exception:
        return None


def cpu_baseline_and_cli(args, R, k, w):
    """The oracle (CPU restatement of the reference, `-t` = host cores) and the drop-in CLI (FASTQ in, files out) on the same
    bounded sample of the bench workload, written to disk: the contigs the first `cpu_sample_reads` bench reads come from (up to
    `cpu_sample_genomes`, filled up with further contigs) as DB.fa + DBDIR, those reads as FASTQ.  The full 26.8 Gbp index is far
    beyond a CPU budget of seconds (the reference indexes ~2 Mbp/s on one thread), so the sample reference is ~1/100 of it;
    per-read CPU cost grows with the seed hits the reference draws, i.e. the CPU figure is an UPPER bound of what the full
    reference would give."""
    from metamaps_amd import synth
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    exe = os.path.join(ROOT, "oracle", "_build", "metamaps_oracle")
    cli = os.path.join(ROOT, "metamaps_amd", "csrc", "metamaps")
    ref, reads, truth, contig_taxon = R["ref"], R["reads"], R["truth"], R["contig_taxon"]
    rl, cl = reads.lengths(), ref.lengths()
    allowed, pick_reads = [], []
    for r in range(len(rl)):
        t = int(truth[r])
        if t >= 0 and t not in allowed:
            if len(allowed) >= args.cpu_sample_genomes or cl[t] > 20_000_000:   # (a human-like contig would be most of the sample)
                continue
            allowed.append(t)
        if t < 0 or t in allowed:
            pick_reads.append(r)
        if len(pick_reads) >= args.cpu_sample_reads:
            break
    contigs = sorted(allowed)
    c = 0
    while len(contigs) < args.cpu_sample_genomes and c < len(cl):
        if c not in contigs and cl[c] <= 20_000_000:
            contigs.append(c)
        c += 1
    nproc = len(os.sched_getaffinity(0))
    cores = args.cpu_threads or min(nproc, 64)
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        db = synth.write_db_dir(os.path.join(d, "db"), [(int(contig_taxon[ci]), ref.fetch(ci, int(cl[ci]))) for ci in contigs])
        # two FASTQ files: the CPU sample, and every bench read that stems from the slice (or from nowhere) for the CLI's throughput
        fq, fq_all, n_all, bases_all = os.path.join(d, "reads.fq"), os.path.join(d, "reads_all.fq"), 0, 0
        in_slice = np.zeros(len(cl) + 1, dtype=bool); in_slice[contigs] = True; in_slice[-1] = True   # (truth -1: random reads)
        small = set(pick_reads)
        with open(fq, "wb") as f, open(fq_all, "wb") as fa:
            for r in range(len(rl)):
                if not in_slice[int(truth[r])]:
                    continue
                s = reads.fetch(r, int(rl[r]))
                rec = f"@r{r}\n".encode() + s + b"\n+\n" + b"I" * len(s) + b"\n"
                fa.write(rec); n_all += 1; bases_all += len(s) if len(s) >= 1000 else 0
                if r in small:
                    f.write(rec)
        t_files = time.time() - t0
        ref_bp = int(sum(int(cl[ci]) for ci in contigs))
        # ---- oracle: mapDirectly (index build single-threaded apart from the winnowing, excluded) + classify
        p = subprocess.run([exe, "mapDirectly", "--all", "-r", db["fasta"], "-q", fq, "-o", os.path.join(d, "cpu"), "-w", str(w), "-t", str(cores)],
                           capture_output=True, check=True, timeout=1500)
        js = json.loads(p.stderr.decode().strip().splitlines()[-1])
        t0 = time.time()
        subprocess.run([exe, "classify", "--DB", db["dir"], "--mappings", os.path.join(d, "cpu"), "-t", str(cores)], capture_output=True, check=True, timeout=1500)
        t_cls = time.time() - t0
        bases = js["bases"]
        cpu = {"value": bases / (js["map_seconds"] + t_cls) / 1e9, "unit": "Gbp/s", "cores": cores, "nproc": nproc, "kind": "port",
               "sample": f"{len(pick_reads)} of the bench reads ({bases} bp long enough) vs a {len(contigs)}-contig slice of the bench reference ({ref_bp / 1e6:.1f} Mbp, "
                         f"{100.0 * ref_bp / R['reference_bp']:.2f} % of it), oracle -t {cores}: mapping {js['map_seconds']:.2f} s + classify {t_cls:.2f} s "
                         f"(index build {js['seconds'] - js['map_seconds']:.2f} s excluded, as for the GPU)",
               "mapping_only_value": bases / js["map_seconds"] / 1e9, "map_seconds": js["map_seconds"], "classify_seconds": t_cls, "mappings": js["mappings"]}
        # ---- the drop-in CLI on the same files: FASTQ in -> mapping file + .meta, classify -> WIMP etc. (index build timed apart)
        env = dict(os.environ, MM_CLI_TIMING="1")
        t0 = time.time()
        p = subprocess.run([cli, "mapDirectly", "--all", "-r", db["fasta"], "-q", fq, "-o", os.path.join(d, "gpu"), "-w", str(w)], capture_output=True, check=True, timeout=900, env=env)
        t_map_all = time.time() - t0
        laps = {}
        for ln in p.stderr.decode().splitlines():
            if ln.startswith("INFO, lap "):
                laps[ln.split(" at +")[0][len("INFO, lap "):]] = float(ln.split(" at +")[1].split()[0])
        t_setup = laps.get("3 index build", 0.0)                # context + reference parse + pack + index build
        t0 = time.time()
        subprocess.run([cli, "classify", "--DB", db["dir"], "--mappings", os.path.join(d, "gpu")], capture_output=True, check=True, timeout=900)
        t_cli_cls = time.time() - t0
        same = open(os.path.join(d, "gpu.EM.reads2Taxon")).read() == open(os.path.join(d, "cpu.EM.reads2Taxon")).read()
        small_run = {"map_seconds": t_map_all - t_setup, "classify_seconds": t_cli_cls, "reads": len(pick_reads)}
        # ... and on every bench read of the slice (fixed costs of two process starts weigh less)
        t0 = time.time()
        p = subprocess.run([cli, "mapDirectly", "--all", "-r", db["fasta"], "-q", fq_all, "-o", os.path.join(d, "gpuall"), "-w", str(w)], capture_output=True, check=True, timeout=900, env=env)
        t_map_all = time.time() - t0
        map_phases = {}
        for ln in p.stderr.decode().splitlines():
            if ln.startswith("INFO, lap 3 index build"):
                t_setup = float(ln.split(" at +")[1].split()[0])
            if ln.startswith("INFO, time "):
                map_phases[" ".join(ln.split()[2:-2])] = float(ln.split()[-2])
        map_phases["process wall"] = t_map_all
        t0 = time.time()
        pc = subprocess.run([cli, "classify", "--DB", db["dir"], "--mappings", os.path.join(d, "gpuall")], capture_output=True, check=True, timeout=900, env=env)
        t_cli_cls = time.time() - t0
        cls_phases = {ln.split()[2] + " " + " ".join(ln.split()[3:-2]): float(ln.split()[-2]) for ln in pc.stderr.decode().splitlines() if ln.startswith("INFO, time c")}
        cls_phases.update({"main: " + ln.split(" at +")[0][12:]: float(ln.split(" at +")[1].split()[0]) for ln in pc.stderr.decode().splitlines() if ln.startswith("INFO, main: ")})
        e2e = {"value": bases_all / max(t_map_all - t_setup + t_cli_cls, 1e-9) / 1e9, "unit": "Gbp/s",
               "what": "metamaps mapDirectly (reads FASTQ -> PREFIX, .meta) + metamaps classify (-> .EM.*), wall clock of the two processes minus "
                       "context + reference parse + index build, on every bench read that stems from the cpu_baseline's reference slice",
               "reads": n_all, "bases": bases_all, "map_seconds": t_map_all - t_setup, "setup_seconds": t_setup, "classify_seconds": t_cli_cls,
               "mapping_only_value": bases_all / max(t_map_all - t_setup, 1e-9) / 1e9,
               "map_phases_s": map_phases, "classify_phases_s": cls_phases, "on_the_cpu_sample": small_run, "reads2taxon_identical_to_oracle": same, "sample_files_written_s": round(t_files, 2)}
    return cpu, e2e


if __name__ == "__main__":
    main()

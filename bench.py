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
`--reads` reads (weak scaling: per-GPU work fixed; the default) or — `--scaling strong`, BASELINE configs[2] — its contiguous shard of
ONE batch of `--reads` reads that every rank generates; EM sufficient statistics all-reduced over RCCL (also at N = 1: the same
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
HBM_PEAK_MEASURED_GBS = 6300.0   # what a streaming read of K5's access pattern (46 KB runs from ~6 000 waves) reaches on this part: tools/ubench/stream46k, docs/history.md section 6
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json")]   # the newest committed PMC summary that exists


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 36, --config 3: 12; --config 4: 4 passes, --config 5: 2 passes of ~20 s)")   # (three worker contexts take the steps in turn: a timed region of three steps would be one round of them, no steady state)
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 3; --config 4: 1, --config 5: 0)")
    # workload: BASELINE configs[1] shape; scale knobs exist so that smaller boxes / quick checks can run
    ap.add_argument("--shape", choices=("community", "uniform"), default=os.environ.get("MM_BENCH_SHAPE", "community"))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("MM_BENCH_READS", 100_000)), help="reads per GPU")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("MM_BENCH_SCALING", "weak"),
                    help="weak (default): every rank maps its own --reads reads per step.  strong: --reads is the TOTAL per step (BASELINE configs[2]: the 100k x 10 kb "
                         "batch of configs[1] sharded 8x); every rank generates the same batches (seed independent of the rank) and maps "
                         "shard_range(total, rank, world) of each (metamaps_amd/dist.py), the EM sums are all-reduced as at every N")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--read-len-min", type=int, default=0, help="mixed lengths, log-uniform in [read-len-min, read-len] (BASELINE config 3 shape); 0 = fixed")
    ap.add_argument("--pacbio", action="store_true", help="PacBio-like errors (2/8/2 percent del/ins/sub) instead of ONT-like (5/3/4)")
    ap.add_argument("--scale", type=float, default=float(os.environ.get("MM_BENCH_SCALE", 1.0)), help="scales the number of genomes of the reference (quick checks)")
    ap.add_argument("--window", type=int, default=8, help="w the CLI derives for a 26.76 GB DB.fa at default flags")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("MM_BENCH_WORKERS", 3)),
                    help="host threads / contexts per GPU that take the steps in turn, each on its own stream: the kernels of two steps share the GPU "
                         "(what one leaves idle — launch gaps, draining kernels, host sections — the other fills).  Measured on the distinct-batch "
                         "workload (round 3): 2 workers 47.3 ms per step, 3 workers 47.6, 3 workers with the mapping sections serialised "
                         "(--serialise-map, the default of rounds 2-3) 50.6.  Round 5 (tools: in turns on one box): at 10^5 reads per step 2, 3 and 4 workers are "
                         "the same within the noise (45.1-46.2 ms); at configs[2]'s per-GPU share of 12 500 reads a step is a third launch gaps and host sections, "
                         "and a third worker fills them: 6.86 -> 6.29 ms per step (median 7.0 -> 5.8): the default is 3")
    ap.add_argument("--serialise-map", action="store_true", help="one lock around the mapping section of a step, passed on when its last big kernel (K5) is enqueued: "
                    "kernel durations are then those of kernels that (nearly) own the GPU; 7 percent less throughput on distinct batches")
    ap.add_argument("--staged-map", action="store_true", help="two locks instead of one around the mapping section (K1 + K2 | K3 ... K6, swapped at mm_map_batch_phased's "
                    "callback): the next step's minimizer stage runs under this step's seed stage; ~2 percent more throughput, but the seed filter's "
                    "duration then includes the time it shares the CUs (K1 issues VALU instructions in 99 percent of its cycles: the two do not complement each other)")
    ap.add_argument("--hold-lock-to-the-end", action="store_true", help="release the mapping lock when mm_map_batch returns instead of when its last big kernel is enqueued")
    ap.add_argument("--free-overlap", action="store_true", help="(the default since round 3; kept for old command lines) do not serialise the mapping sections of the workers")
    ap.add_argument("--measure-free-overlap", action="store_true", help="after the timed region, six more steps with nothing serialised, reported in config.free_overlap")
    ap.add_argument("--config", type=int, choices=(1, 3, 4, 5), default=1, help="BASELINE.json configs[N] as far as one GPU carries it: 1 (default, the configuration `value` is "
                    "quoted on) 100k x 10 kb ONT reads vs the resident index; 3: mixed 1-50 kb PacBio reads vs the index split by the --maxmemory chunk rule into resident "
                    "chunk indexes; 4: 10 kb reads vs chunk indexes that are built, mapped and dropped in turn (the multi-pass streaming of an index larger than HBM)")
    ap.add_argument("--distinct-batches", type=int, default=24, help="read batches generated up front (seeds 1000 + rank + 97 i); step s maps batch s mod this")
    ap.add_argument("--chunk-gib", type=float, default=0.0, help="--maxmemory of configs 3 / 4 in GiB (default 70 -> 4 chunks for config 3, 25 -> 11 chunks for config 4)")
    ap.add_argument("--batches-per-pass", type=int, default=4, help="config 4: read batches mapped against every chunk index of a pass (the index builds of a pass are inside the timed region)")
    ap.add_argument("--no-e2e-full", action="store_true", help="skip e2e_cli_full (the drop-in CLI on the whole 26.8 GB DB.fa written to local disk: ~2 minutes)")
    ap.add_argument("--no-e2e-stream", action="store_true", help="skip e2e_cli_stream (the CLI on --e2e-stream-batches batches in one FASTQ; inside e2e_cli_full)")
    ap.add_argument("--e2e-stream-batches", type=int, default=10)
    ap.add_argument("--e2e-dir", default=os.environ.get("MM_BENCH_E2E_DIR", ""), help="directory for the files of e2e_cli_full (default: a temporary directory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-shape", action="store_true", help="skip the second reference shape (config.other_shape)")
    ap.add_argument("--cpu-sample-reads", type=int, default=20000)
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads for cpu_baseline (0 = all host cores)")
    ap.add_argument("--cpu-sample-genomes", type=int, default=100, help="at least this many contigs in the CPU sample reference ...")
    ap.add_argument("--cpu-sample-bases", type=float, default=1.0e9, help="... and at least this many bases (filled up with further contigs of the bench reference)")
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


def default_steps(args):
    """K / W when the caller names none: enough steps for a steady state of three workers, few passes where a pass is seconds"""
    if args.steps is None:
        # 36 steps of ~37 ms: the three worker contexts start the timed region together and finish it together — ~17 ms of fill and drain, which twelve steps carried as
        # 1.4 ms each (tools/alloc_probe.sh, MM_BENCH_STEP_LOG: 39.7 ms per step over 12 steps, 37.5-37.9 over 36, 36.9 over 60 on one box)
        args.steps = {3: 12, 4: 4, 5: 2}.get(args.config, 36)
    if args.warmup is None:
        args.warmup = {4: 1, 5: 0}.get(args.config, 3)


def sched_free(args):
    return not ((args.serialise_map or args.staged_map) and not args.free_overlap)


def main():
    args = parse_args()
    default_steps(args)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and "MM_CPU_BUDGET" not in os.environ:
        # one process per GPU: every rank would size its pools (and choose between spinning and sleeping on its streams) from the whole
        # container's CPU quota; a rank gets its share (read by the library at its first call, cpu_budget.hpp)
        os.environ["MM_CPU_BUDGET"] = str(max(2, min(os.cpu_count() or 1, cpu_quota() or (os.cpu_count() or 1)) // world))
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
        # the build's sort buffers (~100 GB, kept pooled by the library for the next build or for the contexts' own buffers) go back to the
        # driver here: this process builds one index per shape, and the CLI legs further down are other processes on the same device
        ctx.release_cached()
        # distinct read batches, generated up front (packed 2-bit, resident): step s maps batch s mod B, so that no step re-maps what the
        # step before it left in the caches.  The worker contexts read them in turn (read-only; any context of the device may).
        B = max(1, min(args.distinct_batches, steps + max(warmup, 0)))
        batches, truth = [], None
        strong = args.scaling == "strong"
        from metamaps_amd.dist import shard_range
        lo, hi = shard_range(args.reads, rank, world) if strong else (0, args.reads)
        for b in range(B):
            rd, tr = ctx.synth_reads(ref, seed=1000 + (0 if strong else rank) + 97 * b, n_reads=args.reads, read_len=args.read_len, read_len_min=args.read_len_min,
                                     frac_random=0.05, n_abundant=100, **err)
            if strong and world > 1:                                # this rank's contiguous shard of the batch every rank generated
                whole = rd
                rd = whole.slice(lo, hi - lo); whole.close()
                tr = tr[lo:hi]
            batches.append(rd)
            truth = tr if b == 0 else truth
        reads = batches[0]
        ctx.synchronize()
        contig_len = ref.lengths().astype(np.int32)
        agg = {"ms_l2": 0.0, "ms_hf": 0.0, "ms_mz": 0.0, "launches": 0, "l2_stream": 0, "hf_units": 0, "stats": None, "em_iters": 0, "bases": 0, "done_t": []}
        rec_bufs = [np.empty(max(64 * (hi - lo), 1 << 16), dtype=capi.RECORD_DTYPE) for _ in ctxs]   # host result buffers reused by every step
        map_lock, agg_lock = threading.Lock(), threading.Lock()
        front_lock, back_lock = threading.Lock(), threading.Lock()
        hold_lock = [bool(args.hold_lock_to_the_end)]
        em_turn = {"next": 0, "cv": threading.Condition()}
        em_wall = [0.0, 0]                                            # seconds inside em.run, iterations

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
                    M = c.map_batch(idx, batches[ticket % B], k, w, pi=80.0, min_read_len=1000, at_seed_stage=swap if staged else None,
                                    at_last_kernel=release if (serialise and not hold_lock[0]) else None)
                finally:
                    if staged and not swapped[0]:
                        swap()
                tt.append(time.perf_counter())
            finally:
                release()
            M.add_qualities(k)                                      # K8: three small launches, thread per mapping (round 3: one thread per read, 3.2 ms)
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
                t_em = time.perf_counter()
                f, lls = em.run(f)                                  # the EM loop, device resident (fEM.h:501-661)
                em_wall[0] += time.perf_counter() - t_em; em_wall[1] += len(lls)
            finally:
                with em_turn["cv"]:
                    em_turn["next"] = ticket + 1
                    em_turn["cv"].notify_all()
            tt.append(time.perf_counter())
            post, best = em.posteriors(f)
            em.close()
            tt.append(time.perf_counter())
            if os.environ.get("MM_BENCH_STEP_LOG"):                  # per step: worker, batch, host sections and the device's stage times (which step of the timed region is long, and where)
                print(f"STEPLOG ticket {ticket} worker {wi} batch {ticket % B}: start {tt[0] * 1e3:.1f} map_batch {(tt[1] - tt[0]) * 1e3:.1f} mapq+fetch {(tt[2] - tt[1]) * 1e3:.1f} "
                      f"em {(tt[3] - tt[2]) * 1e3:.1f} post {(tt[4] - tt[3]) * 1e3:.1f} end {tt[4] * 1e3:.1f} | device ms: K1 {st['ms_minimizer']:.1f} K2 {st['ms_sketch']:.1f} K3 {st['ms_probe_gather']:.1f} "
                      f"sort {st['ms_sort_hits']:.1f} L1 {st['ms_l1_scan']:.1f} K5 {st['ms_l2']:.1f} total {st['ms_total']:.1f} | cands {st['n_candidates']} em_iters {len(lls)}", file=sys.stderr, flush=True)
            with agg_lock:
                agg["host_ms"] = {"map_batch": (tt[1] - tt[0]) * 1e3, "mapq_fetch": (tt[2] - tt[1]) * 1e3, "em_prepare_iterate": (tt[3] - tt[2]) * 1e3,
                                  "posteriors": (tt[4] - tt[3]) * 1e3}
                agg["ms_l2"] += st["ms_l2"]; agg["launches"] += 1; agg["l2_stream"] += st["sum_l2_stream_entries"]
                agg["ms_hf"] += st["ms_hit_filter"]; agg["hf_units"] += st["sum_hits"] + st["sum_sketch"]; agg["ms_mz"] += st["ms_minimizer"]
                agg["stats"] = st; agg["em_iters"] = len(lls); agg["bases"] += st["bases_long_enough"]; agg["done_t"].append(tt[4])
            return st

        step_base = [0]

        def run_steps(n, serialise=True):
            """n steps, taken in turn by the worker threads (every rank runs the same schedule, so the collectives of communicator i match)"""
            em_turn["next"] = step_base[0]
            base = step_base[0]
            step_base[0] += n
            def work(wi):
                for s_i in range(wi, n, W):
                    step(wi, serialise, base + s_i)
            th = [threading.Thread(target=work, args=(wi,)) for wi in range(1, W)]
            for t in th:
                t.start()
            work(0)
            for t in th:
                t.join()

        for wi in range(1, W):                                    # setup: every further worker context runs once (its scratch buffers get allocated)
            em_turn["next"] = 0; step(wi, True, 0)
        sched = ("staged" if (args.staged_map and W > 1) else True) if (args.serialise_map or args.staged_map) and not args.free_overlap else False
        run_steps(max(warmup, 0), sched)
        agg.update({"ms_l2": 0.0, "ms_hf": 0.0, "ms_mz": 0.0, "launches": 0, "l2_stream": 0, "hf_units": 0, "bases": 0, "done_t": []})
        barrier()
        t0 = time.perf_counter()
        if os.environ.get("MM_ALLOC_TRACE"): print(f"MM_ALLOC_TRACE (bench) timed region starts at {time.time() * 1e3:.1f} ms", file=sys.stderr, flush=True)
        run_steps(steps, sched)
        barrier()
        dt = time.perf_counter() - t0
        if os.environ.get("MM_ALLOC_TRACE"): print(f"MM_ALLOC_TRACE (bench) timed region ends at {time.time() * 1e3:.1f} ms", file=sys.stderr, flush=True)
        if os.environ.get("MM_BENCH_RANK_LOG"):                       # tools/scale_check.sh: what every rank saw
            nr, rk = ctx.comm_info()
            print(f"RANKLOG rank {rank}/{world}: RCCL communicator of {nr} ranks (this one {rk}), {steps} steps in {dt * 1e3:.1f} ms, "
                  f"{em_wall[1]} EM iterations at {em_wall[0] / max(em_wall[1], 1) * 1e6:.1f} us each (warm-up included)", file=sys.stderr, flush=True)
        st = agg["stats"]
        bases_timed = float(agg["bases"])
        # step times inside the timed region: a step's time = from the previous step's completion (or the start) to its own — with W worker
        # contexts the steps overlap, so this is the steady-state distance between results, the quantity `value` averages
        done = sorted(agg["done_t"])
        gaps = np.diff(np.array([t0] + done)) * 1e3 if done else np.array([dt * 1e3])
        step_ms = {"min": float(gaps.min()), "median": float(np.median(gaps)), "max": float(gaps.max()), "first": float(gaps[0])}
        if len(gaps) <= 64:
            step_ms["all"] = [round(float(g), 1) for g in gaps]
        # stage times of a step whose kernels all own the GPU (the timed region lets the next step's K1 queue behind K5): two more steps, untimed
        st_clean = st
        if W > 1 and not hold_lock[0]:                               # (one lock, held to the end of the mapping section)
            keep = dict(agg); keep["done_t"] = list(agg["done_t"])
            hold_lock[0] = True; run_steps(2, True); hold_lock[0] = False
            st_clean = agg["stats"]
            agg.clear(); agg.update(keep)
        # beside the headline: the same number of steps ... six steps that map ONE batch again and again (what rounds 1 and 2 timed): the
        # distance between the two is what the batch-to-batch variation of the workload costs (other genomes, other candidate counts)
        same = None
        if world == 1 and B > 1 and shape == args.shape:
            keep = dict(agg); keep["done_t"] = list(agg["done_t"]); agg["bases"] = 0
            saved = list(batches)
            batches[:] = [saved[0]] * len(saved)
            run_steps(2, sched)
            agg["bases"] = 0
            barrier(); t1 = time.perf_counter(); run_steps(6, sched); barrier()
            d1 = time.perf_counter() - t1
            same = {"ms_per_step": d1 / 6 * 1e3, "value": float(agg["bases"]) / d1 / 1e9, "steps": 6}
            batches[:] = saved
            agg.clear(); agg.update(keep)
        free = None
        if args.measure_free_overlap and W > 1 and sched is not False and world == 1 and shape == args.shape:   # beside the headline: the same steps with nothing serialised
            keep = dict(agg); agg["bases"] = 0
            barrier(); t1 = time.perf_counter(); run_steps(6, False); barrier()
            d1 = time.perf_counter() - t1
            free = {"ms_per_step": d1 / 6 * 1e3, "value": float(agg["bases"]) / d1 / 1e9, "steps": 6}
            agg.clear(); agg.update(keep)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            bb = torch.tensor([bases_timed], dtype=torch.float64, device="cuda")
            dist.all_reduce(bb, op=dist.ReduceOp.SUM)
            bases_all = float(bb.item())
        else:
            bases_all = bases_timed
        return dict(ref=ref, idx=idx, reads=reads, reads_w=batches, truth=truth, contig_taxon=contig_taxon, info=info, desc=desc, t_ref=t_ref, t_index=t_index, free=free,
                    agg=agg, st=st, st_clean=st_clean, same_batch=same, dt=dt, steps=steps, bases_all=bases_all, value=bases_all / dt / 1e9, ms_step=dt / steps * 1e3, step_ms=step_ms,
                    n_batches=B, freq_threshold=idx.freq_threshold, reference_bp=int(ref.total_bases))

    if args.config == 5 and args.scale == 1.0:
        args.scale = 11.2                                          # SURVEY D1, DB-refseq-scale: 140 000 genomes, ~300 Gbp
    if args.config == 5 and args.window == 8:
        args.window = w = 6                                        # what the CLI derives for a ~300 GB DB.fa
    if args.config in (3, 4, 5):
        def allreduce_max_sum(dt, bases):
            if world == 1:
                return dt, bases
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            bb = torch.tensor([bases], dtype=torch.float64, device="cuda"); dist.all_reduce(bb, op=dist.ReduceOp.SUM)
            return float(tt.item()), float(bb.item())
        out = run_chunked(args, ctxs, k, w, rank, world, barrier, allreduce_max_sum)
        for c in ctxs:
            c.close()
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    R = run_shape(args.shape, args.steps, args.warmup)

    out = None
    if rank == 0:
        agg, st, info = R["agg"], R["st"], R["info"]
        roof = roofline_block(args, R)
        roof["peak_measured"] = HBM_PEAK_MEASURED_GBS
        roof["peak_measured_source"] = "tools/ubench/stream46k (round 3): 6.3 TB/s for this access pattern; frac stays algorithmic bytes / the nominal 8 TB/s"
        roof["frac_of_measured_peak"] = roof["achieved"] / HBM_PEAK_MEASURED_GBS if roof.get("achieved") else None
        len_txt = f"{args.read_len}" if not args.read_len_min else f"{args.read_len_min}-{args.read_len}"
        out_workload = (f"{args.reads} synthetic {len_txt} bp {'PacBio' if args.pacbio else 'ONT'}-error reads {'per GPU' if args.scaling == 'weak' else f'in all, sharded over {world} GPU(s)'} vs synthetic miniSeq+H-shaped index "
                        f"({R['desc']}; {R['reference_bp'] / 1e9:.2f} Gbp), k=16 w={w}, --all")
        out = {
            "metric": "Gbp long reads mapped+classified per sec (whole node), miniSeq+H DB",
            "value": R["value"], "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": R["ms_step"], "step_ms": {kk: (round(v, 3) if not isinstance(v, list) else v) for kk, v in R["step_ms"].items()}, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": out_workload,
                "baseline_config": "configs[1]" if args.scaling == "weak" else f"configs[2] (the configs[1] batch of {args.reads} reads sharded x{world}: strong scaling)",
                "distinct_read_batches": R["n_batches"], "one_batch_repeated": R["same_batch"],
                "reads_per_gpu": args.reads if args.scaling == "weak" else [list(__import__("metamaps_amd.dist", fromlist=["shard_range"]).shard_range(args.reads, r, world)) for r in range(world)], "read_len": args.read_len, "reference_bp": R["reference_bp"], "reference_contigs": info["n_contigs"],
                "index_entries": info["n_entries"], "index_unique_hashes": info["n_unique_hashes"], "index_hbm_bytes": info["hbm_bytes"],
                "freq_threshold": R["freq_threshold"], "reference_synth_s": round(R["t_ref"], 3), "index_build_s": round(R["t_index"], 3),
                "parallelism": f"reads sharded x{world} ({args.scaling} scaling), index replicated, RCCL all-reduce of EM sums; {W} worker contexts per GPU take the steps in turn"
                               + ("" if W == 1 else " (nothing serialised: the kernels of the steps in flight share the GPU)" if sched_free(args) else ((" (mapping sections serialised" + ("" if args.hold_lock_to_the_end else "; the lock passes on when a step's last big kernel, K5, is enqueued: the next step's minimizer "
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
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:              # (the contract asks for it at N=1 only)
            # the CLI of the e2e_cli leg is another process on this device: what this one holds in reserve goes back first (the contexts'
            # cached blocks, which since round 4 are pieces of the ~100 GB of pooled index-build buffers: with those kept the device had
            # 0.6 GiB left and the CLI's index build failed for lack of memory)
            for c in ctxs:
                c.release_cached()
            try:
                out["cpu_baseline"], out["e2e_cli"] = cpu_baseline_and_cli(args, R, k, w)
            except Exception as e:  # the baseline is a reported side number; never let it kill the bench line
                tail = getattr(e, "stderr", None)
                tail = (" | stderr: " + tail.decode(errors="replace")[-400:]) if isinstance(tail, (bytes, bytearray)) and tail else ""
                out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": 1, "kind": "port", "sample": f"failed: {e}{tail}"}
    # the other reference shape, beside the headline (one GPU only: it costs a second index build)
    if world == 1 and not args.no_other_shape:
        for rd in R["reads_w"]:
            rd.close()
        for kk in ("idx", "ref"):
            R[kk].close()
        other = "uniform" if args.shape == "community" else "community"
        try:
            R2 = run_shape(other, max(3, min(args.steps, 8)), max(1, args.warmup))   # (the headline's warm-up: the steps right after an index build can meet
            # the runtime's clean-up of the memory the previous index gave back — a 1-2 s stall of one step, docs/history.md section 6)
            out["config"]["other_shape"] = {"shape": R2["desc"], "value": R2["value"], "unit": "Gbp/s", "ms_per_step": R2["ms_step"], "steps": R2["steps"], "warmup": max(1, args.warmup),
                                            "step_ms": {kk: (round(v, 3) if not isinstance(v, list) else v) for kk, v in R2["step_ms"].items()},
                                            "reference_bp": R2["reference_bp"], "freq_threshold": R2["freq_threshold"],
                                            "stage_ms": {kk: round(R2["st_clean"][kk], 3) for kk in R2["st"] if kk.startswith("ms_")},
                                            "per_step": {kk: R2["st"][kk] for kk in ("n_reads_mapped", "n_mappings", "sum_sketch", "sum_hits", "sum_hits_kept", "n_candidates", "sum_l2_stream_entries")}}
            for rd in R2["reads_w"]:
                rd.close()
            R2["idx"].close(); R2["ref"].close()
        except Exception as e:
            out["config"]["other_shape"] = {"failed": str(e)}
    for rd in R.get("reads_w", []):                              # everything off the device before the contexts go (and the CLI of e2e_cli_full comes)
        rd.close()
    R["idx"].close(); R["ref"].close()
    for c in ctxs:
        c.close()
    if rank == 0 and world == 1 and not args.no_e2e_full and (args.scale == 1.0 or os.environ.get("MM_BENCH_E2E_ANY_SCALE")) and args.shape == "community" and not args.read_len_min:
        try:
            out["e2e_cli_full"] = e2e_cli_full(args, k, w, 1000 + rank)
            out["e2e_cli_stream"] = out["e2e_cli_full"].pop("e2e_cli_stream", None)
            es = out["e2e_cli_stream"] or {}
            # the boundary numbers beside `value` (which is the resident-data pipeline): the drop-in CLI in steady state, FASTQ in -> every classify file out
            out["cli_value"] = es.get("value")                    # one process, `mapDirectly --then-classify`
            out["cli_two_process_value"] = (es.get("two_processes") or {}).get("value_all_in")   # the reference's form: `mapDirectly`, then `classify`
            out["cli_value_note"] = "Gbp/s of e2e_cli_stream (index built -> last classify file written; see that object); `value` has reads generated on the device and no text"
        except Exception as e:  # a reported side number; never let it kill the bench line
            out["e2e_cli_full"] = {"failed": str(e)[:600]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_chunked(args, ctxs, k, w, rank, world, barrier, allreduce_max_sum):
    """BASELINE configs[3] / configs[4] as far as one GPU carries them (the same JSON contract as the default mode):
      3  mixed-length PacBio-error reads (1-50 kb, log-uniform) against the reference split by the --maxmemory chunk rule
         (winSketch.hpp:274-329) into chunk indexes that all stay resident; a step maps one batch against every chunk, merges read-wise in
         chunk order (unifyFiles, mapWrap.h:128-145), mapping qualities over the union, EM
      4  the multi-pass streaming of an index that does not fit HBM: a step is one PASS — every chunk index is built, `--batches-per-pass`
         resident read batches are mapped against it, the records go to the host, the index is dropped; then merge, mapping qualities, EM
         per batch.  The index builds are inside the timed region (they recur with every pass; the reference rebuilds them with every run)."""
    from metamaps_amd import capi
    import threading
    mode = args.config
    ctx = ctxs[0]
    W = len(ctxs) if mode == 3 else 1                               # config 3: the steps are taken in turn by the worker contexts, as in the default mode
    if mode == 3 and args.read_len == 10_000 and not args.read_len_min:      # config 3's read shape unless the caller chose one
        args.read_len, args.read_len_min, args.pacbio, args.reads = 50_000, 1_000, True, min(args.reads, 60_000)
    err = dict(sub_rate=0.02, ins_rate=0.08, del_rate=0.02) if args.pacbio else dict(sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
    gib = args.chunk_gib or (70.0 if mode == 3 else 25.0 if mode == 4 else 150.0)
    t0 = time.time()
    ref, contig_taxon, n_taxa, desc = build_reference(ctx, args, args.shape)
    contig_len = ref.lengths().astype(np.int32)
    if mode == 5:
        # configs[4] at its size: the index of the whole reference does not fit the device, so the chunk rule is evaluated on the indexes of contig
        # ranges, as the CLI does (metamaps_main.cpp: every cut inside a range is final, the range's last, open chunk starts the next range)
        from metamaps_amd.chunkplan import plan_chunks_by_ranges
        plan, info = plan_chunks_by_ranges(ctx, ref, contig_len, k, w, int(gib * (1 << 30)), int(float(os.environ.get("MM_BENCH_RANGE_GBP", 8)) * 1e9))
    else:
        whole = ctx.index(ref, k, w)
        info = whole.info()
        plan = whole.plan_chunks(int(gib * args.scale * (1 << 30)))
        whole.close()
    bounds = [(a, (plan[i + 1] if i + 1 < len(plan) else ref.count) - a) for i, a in enumerate(plan)]
    base = [a for a, _ in bounds]
    # per-chunk freqThreshold from the histogram accumulated over the chunks (never cleared, winSketch.hpp:452-494)
    from metamaps_amd.chunkplan import AccumulatedThreshold
    acc_thr, thrs, chunk_idx, t_build = AccumulatedThreshold(), [], [], []
    for a, n in bounds:
        tb = time.time()
        sl = ref.slice(a, n); ix = ctx.index(sl, k, w, auto_threshold=False); sl.close()
        ctx.synchronize(); t_build.append(time.time() - tb)
        thrs.append(acc_thr.next(ix))
        if mode == 3:
            chunk_idx.append(ix)
        else:
            ix.close()
    t_setup = time.time() - t0
    n_batches = max(1, min(args.distinct_batches, (args.steps + max(args.warmup, 0)) * (args.batches_per_pass if mode in (4, 5) else 1)))
    batches = [ctx.synth_reads(ref, seed=1000 + rank + 97 * b, n_reads=args.reads, read_len=args.read_len, read_len_min=args.read_len_min, frac_random=0.05, n_abundant=100, **err)[0]
               for b in range(n_batches)]
    lens = [b.lengths() for b in batches]
    agg = {"ms_l2": 0.0, "ms_hf": 0.0, "l2_stream": 0, "hf_units": 0, "bases": 0, "launch_sets": 0, "em_iters": 0, "ms_index": 0.0, "stage": {}, "st": None, "done_t": []}

    agg_lock = threading.Lock()
    em_turn = {"next": 0, "cv": threading.Condition()}

    def note(st):
        agg["ms_l2"] += st["ms_l2"]; agg["ms_hf"] += st["ms_hit_filter"]; agg["l2_stream"] += st["sum_l2_stream_entries"]; agg["hf_units"] += st["sum_hits"] + st["sum_sketch"]
        for kk in st:
            if kk.startswith("ms_"):
                agg["stage"][kk] = agg["stage"].get(kk, 0.0) + st[kk]

    def classify(c, M, ticket=None):
        """ticket: the classify sections of overlapping steps run in step order (one communicator shared by the worker contexts: the same
        sequence of collectives on every rank)"""
        em = c.em_from_mapping(M, contig_taxon, contig_len, n_taxa)
        if ticket is not None:
            with em_turn["cv"]:
                em_turn["cv"].wait_for(lambda: em_turn["next"] == ticket)
        try:
            seen = (em.taxon_counts() > 0).astype(np.float64)
            c.comm_allreduce(seen)
            present = seen > 0
            f, lls = em.run(np.where(present, 1.0 / max(int(present.sum()), 1), 0.0))
        finally:
            if ticket is not None:
                with em_turn["cv"]:
                    em_turn["next"] = ticket + 1
                    em_turn["cv"].notify_all()
        em.posteriors(f)
        em.close()
        agg["em_iters"] = len(lls)

    def step(s_i, c):
        if mode == 3:
            rd = batches[s_i % n_batches]
            parts = []
            for ix in chunk_idx:                                    # (minimizers and sketches once per batch: mm_map_batch_reusing, as the CLI does)
                parts.append(c.map_batch(ix, rd, k, w, pi=80.0, min_read_len=1000, sketch_of=parts[0] if parts else None))
            U = capi.Mapping.concat(c, parts, base)
            for p_ in parts:
                p_.close()
            U.add_qualities(k); U.fetch()
            st = U.stats()
            with agg_lock:
                note(st); agg["st"] = st; agg["bases"] += st["bases_long_enough"]; agg["launch_sets"] += 1
            classify(c, U, s_i); U.close()
        else:
            bs = [(s_i * args.batches_per_pass + j) % n_batches for j in range(args.batches_per_pass)]
            host = [[] for _ in bs]
            sk = [c.sketch_batch(batches[b], k, w, pi=80.0, min_read_len=1000) for b in bs]   # K1 + K2 once per batch and pass, not once per chunk
            for ci, (a, n) in enumerate(bounds):
                tb = time.perf_counter()
                sl = ref.slice(a, n); ix = c.index(sl, k, w, auto_threshold=False); sl.close(); ix.set_freq_threshold(thrs[ci])
                c.synchronize(); agg["ms_index"] += (time.perf_counter() - tb) * 1e3
                for j, b in enumerate(bs):
                    M = c.map_batch(ix, batches[b], k, w, pi=80.0, min_read_len=1000, sketch_of=sk[j])
                    M.release_intermediates(); host[j].append(M)    # the records stay on the device (rounds 1-3: to the host and back, mm_mapping_from_parts)
                    st = M.stats(); note(st); agg["st"] = st
                ix.close()
            for m_ in sk:
                m_.close()
            for j, b in enumerate(bs):
                V = capi.Mapping.concat(c, host[j], base)
                for p_ in host[j]:
                    p_.close()
                V.add_qualities(k); V.fetch()
                agg["bases"] += V.stats()["bases_long_enough"]; agg["launch_sets"] += 1
                classify(c, V); V.close()
        with agg_lock:
            agg["done_t"].append(time.perf_counter())

    def run_steps(first, n):
        em_turn["next"] = first
        def work(wi):
            for s_i in range(wi, n, W):
                step(first + s_i, ctxs[wi])
        th = [threading.Thread(target=work, args=(wi,)) for wi in range(1, W)]
        for t in th:
            t.start()
        work(0)
        for t in th:
            t.join()

    # how many worker contexts the device has room for beside the resident chunk indexes (four of them take 227 of 288 GiB at full
    # scale): the first context's step shows what one needs (hit lists, sort buffers: 16 GiB), every further one must fit 1.3 times over.
    # At full scale: one (two were tried, MM_BENCH_C3_WORKERS=2: out of memory with every cache given back, tools/mem_config3.py)
    if W > 1:
        import torch
        ctx.release_cached()                                       # (the index builds' pooled temporaries: room for a second worker context)
        torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info()[0]
        em_turn["next"] = 0; step(0, ctxs[0])
        torch.cuda.synchronize(); f1 = torch.cuda.mem_get_info()[0]
        need = max(f0 - f1, 1 << 30)
        W = max(1, min(W, 1 + int(f1 / (1.3 * need))))
        if os.environ.get("MM_BENCH_C3_WORKERS"): W = max(1, min(len(ctxs), int(os.environ["MM_BENCH_C3_WORKERS"])))   # (measurement aid)
        print(f"config 3: {f0 / 2**30:.1f} GiB free beside the chunk indexes, {need / 2**30:.1f} GiB per worker context -> {W} worker context(s)", file=sys.stderr)
    for wi in range(1, W):                                          # setup: every further worker context runs once (its scratch buffers get allocated)
        em_turn["next"] = 0; step(0, ctxs[wi])
    run_steps(0, max(args.warmup, 0))
    for kk in ("ms_l2", "ms_hf", "ms_index"):
        agg[kk] = 0.0
    agg.update({"l2_stream": 0, "hf_units": 0, "bases": 0, "launch_sets": 0, "stage": {}, "done_t": []})
    barrier()
    t0 = time.perf_counter()
    run_steps(max(args.warmup, 0), args.steps)
    barrier()
    dt = time.perf_counter() - t0
    dt, bases_all = allreduce_max_sum(dt, float(agg["bases"]))
    gaps = np.diff(np.array([t0] + sorted(agg["done_t"]))) * 1e3
    cands = [("l2_kernel (all launches of the timed region)", 8.0 * agg["l2_stream"], agg["ms_l2"]), ("seed_filter_kernel / hit_filter_kernel (all launches of the timed region)", 8.0 * agg["hf_units"], agg["ms_hf"])]
    dom_name, dom_bytes, dom_ms = max(cands, key=lambda c_: c_[2])
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    len_txt = f"{args.read_len}" if not args.read_len_min else f"{args.read_len_min}-{args.read_len}"
    what = ("the index split by the --maxmemory chunk rule into resident chunk indexes, every batch mapped against each, merged in chunk order" if mode == 3 else
            f"chunk indexes built, mapped ({args.batches_per_pass} resident read batches per pass) and dropped in turn: the index builds of every pass are inside the timed region")
    out = {
        "metric": "Gbp long reads mapped+classified per sec (whole node), miniSeq+H DB",
        "value": bases_all / dt / 1e9, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "step_ms": {"min": round(float(gaps.min()), 3), "median": round(float(np.median(gaps)), 3), "max": round(float(gaps.max()), 3)},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[{min(mode, 4)}] on one GPU per rank: {args.reads} synthetic {len_txt} bp {'PacBio' if args.pacbio else 'ONT'}-error reads per batch vs synthetic miniSeq+H-shaped "
                        f"index ({desc}; {ref.total_bases / 1e9:.2f} Gbp), k=16 w={w}, --all, --maxmemory {gib:g} GiB -> {len(bounds)} index chunks; {what}",
            "baseline_config": f"configs[{min(mode, 4)}]" + (" at its size: the whole reference's index does not fit the device" if mode == 5 else ""), "reads_per_batch": args.reads, "read_len": args.read_len, "read_len_min": args.read_len_min,
            "batches_per_step": args.batches_per_pass if mode in (4, 5) else 1, "distinct_read_batches": n_batches,
            "reference_bp": int(ref.total_bases), "reference_contigs": info["n_contigs"], "index_entries": info["n_entries"],
            "chunks": len(bounds), "chunk_first_contig": base, "chunk_freq_thresholds": thrs, "chunk_index_build_s": [round(x, 3) for x in t_build],
            "setup_s": round(t_setup, 2), "em_iterations": agg["em_iters"],
            "per_timed_region": {"bases": int(agg["bases"]), "batches_classified": agg["launch_sets"], "sum_l2_stream_entries": int(agg["l2_stream"]), "probes_plus_hits": int(agg["hf_units"]),
                                 "ms_index_builds": round(agg["ms_index"], 1), "stage_ms_sum": {kk: round(v, 2) for kk, v in agg["stage"].items()}},
            "mapping_only_value": (bases_all / max(dt - agg["ms_index"] * 1e-3, 1e-9) / 1e9) if mode in (4, 5) else None,
            "parallelism": f"reads sharded x{world}, every rank holds / streams every chunk index, RCCL all-reduce of EM sums; " + (f"{W} worker contexts per GPU take the steps in turn (mapping sections not serialised)" if W > 1 else "one context per GPU, steps one after the other"),
        },
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes": dom_bytes, "ms": dom_ms,
                     "other_kernels": {n_: {"ms": m_, "algorithmic_bytes": b_, "achieved": (b_ / (m_ * 1e-3) / 1e9 if m_ > 0 else 0.0)} for n_, b_, m_ in cands if n_ != dom_name}},
    }
    for b in batches:
        b.close()
    for ix in chunk_idx:
        ix.close()
    ref.close()
    return out


def roofline_block(args, R):
    """The `roofline` object of the bench line.  One regime per field:
      *_alone         the kernel's launches of ONE step with the GPU to itself — HIP events on the worker's stream around the launches, in the two steps
                      right behind the timed region whose mapping sections hold the lock to their end (the durations `rocprofv3 --kernel-trace --stats`
                      shows for this command with --serialise-map: profiles/r06_kernel_stats.txt); algorithmic bytes from THAT step's counters
      *_timed_region  the same events inside the timed region, averaged over its steps: two worker contexts share the GPU there, so a launch also
                      waits for and runs beside the other step's kernels (profiles/r06_kernel_stats_default_cmd.txt); bytes averaged over the same steps
    Algorithmic bytes per launch follow SURVEY.md section 8 D3 (B_map = ceil(L/4) + 8 s + 8 H + 8 sum M + 32 O per read):
      K1 minimizer stage   ceil(L/4)            the packed bases it reads (what it writes, 8 B per minimizer, is not part of D3; VALU-bound)
      K3 seed filter       8 (s + H)            one probe per sketch hash + every seed hit of the kept lists
      K5 L2 kernel pair    8 sum M              every streamed index entry
    frac = bytes / ms_alone / 8 TB/s.  The dominant kernel — `kernel`, `achieved`, `frac`, `ms_per_launch` at the top level — is the one with the largest
    ms_alone.  traffic = HBM bytes per launch from the committed PMC passes (a file constant, not measured by this run): traffic_source."""
    agg, st_t, st_a = R["agg"], R["st"], R["st_clean"]
    nl = max(agg["launches"], 1)
    def entry(kernels, ms_a, b_a, ms_t, b_t, prof_names, bound_note):
        tr, src = measured_traffic(args, prof_names)
        e = {"kernels": kernels, "ms_alone": ms_a, "algorithmic_bytes": b_a, "achieved": (b_a / (ms_a * 1e-3) / 1e9 if ms_a > 0 else 0.0),
             "ms_timed_region": ms_t, "algorithmic_bytes_timed_region": b_t, "achieved_timed_region": (b_t / (ms_t * 1e-3) / 1e9 if ms_t > 0 else 0.0),
             "traffic": tr, "traffic_source": src, "bound": bound_note}
        e["frac"] = e["achieved"] / HBM_PEAK_GBS; e["frac_timed_region"] = e["achieved_timed_region"] / HBM_PEAK_GBS
        e["traffic_over_algorithmic"] = (tr / b_a) if tr and b_a else None
        return e
    kern = {
        "K1": entry("minimizer_kernel<2> + jstar_kernel + compact_tiles_kernel (the stage: mm_map_stats.ms_minimizer)", st_a["ms_minimizer"], st_a["bases_long_enough"] / 4.0,
                    agg["ms_mz"] / nl, agg["bases"] / 4.0 / nl, ["mm::minimizer_kernel<2>"],
                    "VALU: two MurmurHash3 x64-128 per position = sixteen 64-bit multiplies (profiles/r06_sq_counters.txt: 97 percent of the issue slots); the HBM fraction is not its limit.  "
                    "D3 counts only the packed bases it reads: the 8-byte records it writes (2.1 GB per step, twice: staged, then compacted) are what traffic_over_algorithmic shows"),
        "K3": entry("seed_filter_stream_kernel<false> (mm_map_stats.ms_hit_filter)", st_a["ms_hit_filter"], 8.0 * (st_a["sum_hits"] + st_a["sum_sketch"]),
                    agg["ms_hf"] / nl, 8.0 * agg["hf_units"] / nl, ["mm::seed_filter_stream_kernel<false>"],
                    "two things (DESIGN.md section 7): what a CU executes per read (the time follows the CUs at work: profiles/r05_sf_grid_sweep.txt; VALU 37 percent, LDS atomics, 13 barriers) and the memory side's rate "
                    "of random 64-byte requests - table sector, list pieces, survivors: 5.3e8 per launch (FETCH_SIZE) against 47e9 requests/s over a 90 GB footprint, which 64 CUs reach alone (profiles/r05_randread_cus.txt)"),
        "K5": entry("l2z_kernel<4,2,true> + l2z_kernel<2,2,false> (the zone kernel's launch pair of a step: mm_map_stats.ms_l2)", st_a["ms_l2"], 8.0 * st_a["sum_l2_stream_entries"],
                    agg["ms_l2"] / nl, 8.0 * agg["l2_stream"] / nl, ["mm::l2z_kernel<4, 2, true>", "mm::l2z_kernel<2, 2, false>"],
                    "instruction issue: 1.8 VALU + 1.5 scalar wave-instructions per streamed entry (l2_kernel, round 5: 2.8 + 1.3), VALU 62 percent busy (profiles/r06_sq_counters.txt); the stream is read once "
                    "(the band's threshold masks come out of pass A); phase shares: profiles/r06_l2_phases.txt"),
    }
    dom = max(kern, key=lambda kk: kern[kk]["ms_alone"])
    d = kern[dom]
    d3 = {"reads_2bit": st_a["bases_long_enough"] / 4.0, "probes_8s": 8.0 * st_a["sum_sketch"], "seed_hits_8H": 8.0 * st_a["sum_hits"],
          "l2_stream_8M": 8.0 * st_a["sum_l2_stream_entries"], "records_32O": 32.0 * st_a["n_mappings"]}
    d3_sum = sum(d3.values())
    return {"bound": "hbm", "kernel": f"{dom}: {d['kernels']}", "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"], "traffic": d["traffic"],
            "ms_per_launch": d["ms_alone"], "algorithmic_bytes_per_launch": d["algorithmic_bytes"], "regime": "alone (see kernels.*.ms_alone); the same kernel inside the timed region: kernels." + dom + ".*_timed_region",
            "kernels": kern,
            "whole_step": {"d3_bytes": d3, "algorithmic_bytes": d3_sum, "ms_per_step": R["ms_step"], "achieved": d3_sum / (R["ms_step"] * 1e-3) / 1e9,
                           "frac": d3_sum / (R["ms_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "note": "all D3 bytes of one step (counters of the step measured alone) over the timed region's ms_per_step: what the pipeline as a whole moves per second"},
            "how_to_recompute": "frac = kernels.K.algorithmic_bytes / (kernels.K.ms_alone * 1e-3) / 1e9 / peak; ms_alone agrees with the avg_ms column of profiles/r06_kernel_stats.txt "
                                "(K5: the sum of its two launches — that profile runs them one behind the other, MM_L2_ONE_STREAM=1; by default they run side by side on two streams and ms_alone is the time of the pair, "
                                "within 0.1 ms of the sum at this batch size; K1: minimizer_kernel<2> + compact_tiles_kernel + jstar_kernel), ms_timed_region with profiles/r06_kernel_stats_default_cmd.txt; "
                                "the bytes follow from config.per_step (counters of the last timed step; the step measured alone maps another batch: kernels.K.algorithmic_bytes is its own)"}


def measured_traffic(args, prof_names):
    """(HBM bytes per launch, where they come from) for the kernels named as the rocprofv3 summaries name them, from the newest committed PMC summary
    (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, collected and corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
    tools/collect_profiles.sh -> profiles/rNN_pmc_hbm_traffic.txt -> rNN_traffic.json).  bench.py cannot run the profiler itself: the number is a FILE
    CONSTANT, reported only for the workload it was measured on, and the source string says so."""
    for path in TRAFFIC_FILES:
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if t.get("shape") != args.shape or t.get("reads") != args.reads or t.get("read_len") != args.read_len or args.read_len_min or args.scale != 1.0:
            return None, "no PMC pass for this workload"
        vals = [t.get("by_kernel", {}).get(n) for n in prof_names]
        if any(v is None for v in vals):
            continue
        return float(sum(vals)), f"file constant: {os.path.relpath(path, ROOT)} ({' + '.join(prof_names)}; FETCH_SIZE x corr + WRITE_SIZE of one launch on batch 0, not measured by this run)"
    return None, "no committed PMC summary names this kernel"


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max or v1 cfs quota / period), None when unlimited or unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(round(float(q) / float(per))))
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, int(round(q / per)))
    except Exception:
        return None


def cpu_baseline_and_cli(args, R, k, w):
    """The oracle (CPU restatement of the reference, `-t` = every host core) and the drop-in CLI (FASTQ in, files out) on the same
    bounded sample of the bench workload, written to disk: the contigs the first `cpu_sample_reads` bench reads come from (up to
    `cpu_sample_genomes`), filled up with further contigs of the bench reference to `cpu_sample_bases` (1 Gbp), as DB.fa + DBDIR;
    those reads as FASTQ.  The full 26.8 Gbp index is beyond a CPU budget of seconds (the reference indexes ~2 Mbp/s on one thread;
    the oracle fills its hash map with all threads when no --maxmemory is given), so the sample reference is ~1/27 of it; per-read
    CPU cost grows with the seed hits the reference draws, i.e. the CPU figure is an UPPER bound of what the full reference would give."""
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
    have, bases_have, c = set(contigs), int(sum(int(cl[ci]) for ci in contigs)), 0
    while (len(contigs) < args.cpu_sample_genomes or bases_have < args.cpu_sample_bases * min(args.scale, 1.0)) and c < len(cl):
        if c not in have and cl[c] <= 20_000_000:
            contigs.append(c); have.add(c); bases_have += int(cl[c])
        c += 1
    nproc = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    cores = args.cpu_threads or (min(nproc, quota) if quota else nproc)   # every CPU the box grants this process: the affinity mask AND the container's CPU quota
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
        cpu = {"value": bases / (js["map_seconds"] + t_cls) / 1e9, "unit": "Gbp/s", "cores": cores, "nproc": nproc, "cgroup_cpu_quota": quota, "kind": "port",
               "cores_note": "cores = the threads the oracle ran with = min(hardware threads in the affinity mask, the container's CPU quota: cgroup cpu.max).  Rounds 1-4 ran it with one thread per "
                             "hardware thread (256) on a box whose container is allowed 16 CPUs' worth of time per 100 ms and reported cores = 256: the throughput was that of ~16 CPUs then too",
               "sample": f"{len(pick_reads)} of the bench reads ({bases} bp long enough) vs a {len(contigs)}-contig slice of the bench reference ({ref_bp / 1e6:.1f} Mbp, "
                         f"{100.0 * ref_bp / R['reference_bp']:.2f} % of it), oracle -t {cores}: mapping {js['map_seconds']:.2f} s + classify {t_cls:.2f} s "
                         f"(index build {js['seconds'] - js['map_seconds']:.2f} s excluded, as for the GPU)",
               "mapping_only_value": bases / js["map_seconds"] / 1e9, "map_seconds": js["map_seconds"], "classify_seconds": t_cls, "mappings": js["mappings"],
               "reference_fraction": ref_bp / R["reference_bp"],
               "calibration": None,
               "why": "no ratio between this port and the reference's own `-t N` path can be measured: the reference needs Boost, which neither this image nor the GPU box has, and "
                      "stand-in headers are not allowed (DESIGN.md section 2).  The port flatters the CPU twice: it maps against reference_fraction of the index (a read draws "
                      "~1 / reference_fraction times the chance hits at full size) and it fills its hash map and winnows with N threads, where the reference has a single-slot pool "
                      "(ThreadPool.hpp:176-215).  Read the value as an upper bound of the reference's throughput"}
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
        cls_phases.update({"main: " + ln.split(" at +")[0][12:]: float(ln.split(" at +")[1].split()[0]) for ln in pc.stderr.decode().splitlines() if ln.startswith("INFO, main: ") and " at +" in ln})
        e2e = {"value": bases_all / max(t_map_all - t_setup + t_cli_cls, 1e-9) / 1e9, "unit": "Gbp/s",
               "what": "metamaps mapDirectly (reads FASTQ -> PREFIX, .meta) + metamaps classify (-> .EM.*), wall clock of the two processes minus "
                       "context + reference parse + index build, on every bench read that stems from the cpu_baseline's reference slice",
               "reads": n_all, "bases": bases_all, "map_seconds": t_map_all - t_setup, "setup_seconds": t_setup, "classify_seconds": t_cli_cls,
               "mapping_only_value": bases_all / max(t_map_all - t_setup, 1e-9) / 1e9,
               "map_phases_s": map_phases, "classify_phases_s": cls_phases, "on_the_cpu_sample": small_run, "reads2taxon_identical_to_oracle": same, "sample_files_written_s": round(t_files, 2)}
    return cpu, e2e


def _throttled_s():
    """seconds this container's processes have been stopped for having used up their CPU quota (cgroup v2 cpu.stat), None if unknown"""
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("throttled_usec"):
                return int(ln.split()[1]) / 1e6
    except Exception:
        pass
    return None


LAST_CLI_THROTTLED_S = [None]


def _run_cli_with_rss(cmd, env, timeout):
    """run a child process; returns (CompletedProcess-like, wall seconds, peak resident set in bytes — VmHWM polled from /proc); LAST_CLI_THROTTLED_S[0] = seconds the
    container was throttled for CPU quota meanwhile"""
    import threading
    thr0 = _throttled_s()
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    peak = [0]

    def poll():
        while p.poll() is None:
            try:
                for ln in open(f"/proc/{p.pid}/status"):
                    if ln.startswith("VmHWM:"):
                        peak[0] = max(peak[0], int(ln.split()[1]) * 1024)
            except OSError:
                pass
            time.sleep(0.25)
    th = threading.Thread(target=poll, daemon=True); th.start()
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill(); out, err = p.communicate()
        raise RuntimeError(f"{cmd[0]} {cmd[1]} timed out after {timeout} s")
    wall = time.time() - t0                                        # (before the poller is joined: its 0.25 s nap is not the child's time — rounds 2-3 charged it)
    thr1 = _throttled_s()
    LAST_CLI_THROTTLED_S[0] = round(thr1 - thr0, 3) if thr0 is not None and thr1 is not None else None
    th.join(timeout=1)
    if p.returncode != 0:
        raise RuntimeError(f"{cmd[0]} {cmd[1]} failed ({p.returncode}): {err.decode(errors='replace')[-600:]}")
    return out.decode(errors="replace"), err.decode(errors="replace"), wall, peak[0]


def e2e_cli_full(args, k, w, rank_seed):
    """The drop-in binary at configs[1] scale, once: the device generator's community reference written as a 26.8 GB DB.fa (+ taxonInfo,
    taxonomy, contigNstats) to local disk, the bench's first read batch as FASTQ, then `metamaps mapDirectly --all -r DB.fa -q reads.fq
    -o PREFIX` (no -w: the CLI derives it from the file size as the reference does, parseCmdArgs.hpp:363-374) and `metamaps classify`.
    Everything the resident-data headline leaves out is inside: FASTA / FASTQ parsing, 2-bit packing, H2D, the index build, text
    formatting and file output, two process starts.  Runs after the bench has released the device."""
    from metamaps_amd import capi
    cli = os.path.join(ROOT, "metamaps_amd", "csrc", "metamaps")
    own_tmp = None
    if args.e2e_dir:
        d = args.e2e_dir; os.makedirs(d, exist_ok=True)
    else:
        own_tmp = tempfile.TemporaryDirectory(dir="/tmp"); d = own_tmp.name
    try:
        ctx = capi.Context(int(os.environ.get("LOCAL_RANK", 0)))
        t0 = time.time()
        ref, contig_taxon, n_taxa, desc = build_reference(ctx, args, "community")
        cl = ref.lengths()
        db = os.path.join(d, "db"); os.makedirs(os.path.join(db, "taxonomy"), exist_ok=True)
        fasta = os.path.join(db, "DB.fa")
        nodes = {"1": ("1", "no rank", "root"), "2": ("1", "superkingdom", "Bacteria"), "100": ("2", "phylum", "Synthphyla"), "200": ("100", "order", "Synthales"),
                 "300": ("200", "family", "Synthaceae")}
        per_taxon, cids = {}, []
        with open(fasta, "wb", buffering=1 << 24) as f, open(os.path.join(db, "contigNstats_windowSize_1000.txt"), "w") as ns:
            for ci in range(len(cl)):
                g = int(contig_taxon[ci])
                tid, sp, ge = str(1000000 + g), str(500000 + g // 4), str(100000 + g // 16)
                nodes.setdefault(ge, ("300", "genus", f"Synthus{g // 16}")); nodes.setdefault(sp, (ge, "species", f"Synthus{g // 16} species{g // 4}"))
                nodes.setdefault(tid, (sp, "no rank", f"Synthus{g // 16} species{g // 4} strain{g}"))
                cid = f"C{ci}|kraken:taxid|{tid}|SYN{ci:05d}.1"
                seq = ref.fetch(ci, int(cl[ci]))
                f.write(b">" + cid.encode() + b"\n"); f.write(seq); f.write(b"\n")
                per_taxon.setdefault(tid, []).append(f"{cid}={len(seq)}")
                nwin = -(-len(seq) // 1000)
                if b"N" in seq:                                   # (only the human-like contigs carry N runs)
                    a = np.frombuffer(seq, dtype=np.uint8) == ord("N")
                    counts = np.add.reduceat(a.astype(np.int32), np.arange(0, len(seq), 1000))[:nwin].tolist()
                    ns.write(f"{tid}\t{cid}\t" + ";".join(map(str, counts)) + "\n")
                else:
                    ns.write(f"{tid}\t{cid}\t" + ";".join(["0"] * nwin) + "\n")
                del seq
        with open(os.path.join(db, "taxonInfo.txt"), "w") as f:
            for tid, lst in per_taxon.items():
                f.write(tid + " " + ";".join(lst) + "\n")
        with open(os.path.join(db, "taxonomy", "nodes.dmp"), "w") as f:
            for tid, (par, rank_, _) in nodes.items():
                f.write(f"{tid}\t|\t{par}\t|\t{rank_}\t|\n")
        with open(os.path.join(db, "taxonomy", "names.dmp"), "w") as f:
            for tid, (_, _, name) in nodes.items():
                f.write(f"{tid}\t|\t{name}\t|\t\t|\tscientific name\t|\n")
        open(os.path.join(db, "taxonomy", "merged.dmp"), "w").close()
        err = dict(sub_rate=0.02, ins_rate=0.08, del_rate=0.02) if args.pacbio else dict(sub_rate=0.04, ins_rate=0.03, del_rate=0.05)
        reads, _truth = ctx.synth_reads(ref, seed=rank_seed, n_reads=args.reads, read_len=args.read_len, read_len_min=args.read_len_min, frac_random=0.05, n_abundant=100, **err)
        rl = reads.lengths()
        fq, bases = os.path.join(d, "reads.fq"), 0
        with open(fq, "wb", buffering=1 << 24) as f:
            for r in range(len(rl)):
                sq = reads.fetch(r, int(rl[r]))
                f.write(b"@r%d\n" % r); f.write(sq); f.write(b"\n+\n"); f.write(b"I" * len(sq)); f.write(b"\n")
                bases += len(sq) if len(sq) >= 1000 else 0
        t_files = time.time() - t0
        fasta_bytes = os.path.getsize(fasta)
        # e2e_cli_stream: N_STREAM distinct batches (the seeds of the bench's own batches) in ONE FASTQ — the CLI in steady state
        n_stream = 0 if args.no_e2e_stream else max(1, args.e2e_stream_batches)
        fq_s, bases_s, reads_s = os.path.join(d, "reads_stream.fq"), 0, 0
        t1 = time.time()
        if n_stream:
            with open(fq_s, "wb", buffering=1 << 24) as f:
                for b in range(n_stream):
                    rb, _t = ctx.synth_reads(ref, seed=1000 + 97 * b, n_reads=args.reads, read_len=args.read_len, read_len_min=args.read_len_min, frac_random=0.05, n_abundant=100, **err)
                    buf, ln = rb.fetch_range(0, rb.count)
                    mv, at = memoryview(buf), 0
                    qual = b"I" * int(ln.max())
                    for r, L in enumerate(ln.tolist()):
                        f.write(b"@b%dr%d\n" % (b, r)); f.write(mv[at:at + L]); f.write(b"\n+\n"); f.write(qual[:L]); f.write(b"\n")
                        at += L
                        bases_s += L if L >= 1000 else 0
                    reads_s += len(ln)
                    rb.close(); del buf, mv
        t_files_stream = time.time() - t1
        reads.close(); ref.close(); ctx.close()                   # the device is the CLI's from here on
        env = dict(os.environ, MM_CLI_TIMING="1")
        pre = os.path.join(d, "out")
        out, errtxt, t_map, rss_map = _run_cli_with_rss([cli, "mapDirectly", "--all", "-r", fasta, "-q", fq, "-o", pre], env, 1500)
        laps, phases = {}, {}
        for ln in errtxt.splitlines():
            if ln.startswith("INFO, lap "):
                laps[ln.split(" at +")[0][len("INFO, lap "):]] = float(ln.split(" at +")[1].split()[0])
            if ln.startswith("INFO, time "):
                phases[" ".join(ln.split()[2:-2])] = float(ln.split()[-2])
        t_setup = laps.get("3 index build", 0.0)                 # context + reference parse/pack/upload + index build
        out2, err2, t_cls, rss_cls = _run_cli_with_rss([cli, "classify", "--DB", db, "--mappings", pre], env, 1500)
        cls_phases = {ln.split()[2] + " " + " ".join(ln.split()[3:-2]): float(ln.split()[-2]) for ln in err2.splitlines() if ln.startswith("INFO, time c")}
        cls_main = {ln.split(" at +")[0][12:]: float(ln.split(" at +")[1].split()[0]) for ln in err2.splitlines() if ln.startswith("INFO, main: ") and " at +" in ln}
        par = dict(l.split(" ", 1) for l in open(pre + ".parameters").read().splitlines() if " " in l)
        meta = dict(l.split() for l in open(pre + ".meta"))
        # measurement aid: the single-batch mapDirectly again under other environments / flags ("name:KEY=VAL KEY2=VAL2 --flag value;name2:...")
        variants_1 = {}
        for spec in [x for x in os.environ.get("MM_BENCH_E2E_VARIANTS", "").split(";") if x]:
            vname, _, rest = spec.partition(":")
            venv, vflags = dict(env), []
            for tok in rest.split():
                if "=" in tok and not tok.startswith("-"):
                    kk, _, vv = tok.partition("="); venv[kk] = vv
                else:
                    vflags.append(tok)
            for rep in range(int(os.environ.get("MM_BENCH_E2E_VARIANT_REPS", 2))):
                _ov, ev, tv, _rv = _run_cli_with_rss([cli, "mapDirectly", "--all", "-r", fasta, "-q", fq, "-o", pre + "_v"] + vflags, venv, 1500)
                lv = {ln.split(" at +")[0][len("INFO, lap "):]: float(ln.split(" at +")[1].split()[0]) for ln in ev.splitlines() if ln.startswith("INFO, lap ")}
                pv = {" ".join(ln.split()[2:-2]): float(ln.split()[-2]) for ln in ev.splitlines() if ln.startswith("INFO, time ")}
                variants_1.setdefault(vname, []).append({"wall_s": round(tv, 3), "index_build_lap_s": round(lv.get("3 index build", 0) - lv.get("1 reference parse + pack + upload", 0), 3),
                                                         "mapping_phase_s": round(lv.get("8 write", tv) - lv.get("3 index build", 0.0), 3), "map_sum_s": round(pv.get("6 map", 0), 3),
                                                         "same_file": open(pre).read() == open(pre + "_v").read()})
        stream = None
        if n_stream:
            pre_s = os.path.join(d, "out_stream")
            _o, e_s, t_map_s, rss_s = _run_cli_with_rss([cli, "mapDirectly", "--all", "-r", fasta, "-q", fq_s, "-o", pre_s], env, 1500)
            if os.environ.get("MM_BENCH_E2E_LOG"):
                open(os.path.join(os.environ["MM_BENCH_E2E_LOG"], "cli_stream_map.err"), "w").write(e_s)
            laps_s = {ln.split(" at +")[0][len("INFO, lap "):]: float(ln.split(" at +")[1].split()[0]) for ln in e_s.splitlines() if ln.startswith("INFO, lap ")}
            ph_s = {" ".join(ln.split()[2:-2]): float(ln.split()[-2]) for ln in e_s.splitlines() if ln.startswith("INFO, time ")}
            _o2, e_c, t_cls_s, rss_cs = _run_cli_with_rss([cli, "classify", "--DB", db, "--mappings", pre_s], env, 1500)
            if os.environ.get("MM_BENCH_E2E_LOG"):
                open(os.path.join(os.environ["MM_BENCH_E2E_LOG"], "cli_stream_classify.err"), "w").write(e_c)
            # the same in ONE process: `mapDirectly --then-classify DB` (classify_one on the files just written, with the live contexts): nothing between
            # the two sub-commands — no exit with 150 GB to hand back, no second HIP initialisation waiting behind it — and nothing excluded from the clock
            pre_t = os.path.join(d, "out_stream_tc")
            _o4, e_t, t_tc, rss_t = _run_cli_with_rss([cli, "mapDirectly", "--all", "-r", fasta, "-q", fq_s, "-o", pre_t, "--then-classify", db], env, 1500)
            thr_tc = LAST_CLI_THROTTLED_S[0]
            laps_t = {ln.split(" at +")[0][len("INFO, lap "):]: float(ln.split(" at +")[1].split()[0]) for ln in e_t.splitlines() if ln.startswith("INFO, lap ")}
            ph_t = {" ".join(ln.split()[2:-2]): float(ln.split()[-2]) for ln in e_t.splitlines() if ln.startswith("INFO, time ")}
            t_tc_phase = max(laps_t.get("9 classify", t_tc) - laps_t.get("3 index build", 0.0), 1e-9)
            tc_same = all(open(pre_t + suf, "rb").read() == open(pre_s + suf, "rb").read()
                          for suf in ("", ".meta", ".EM", ".EM.reads2Taxon", ".EM.reads2Taxon.krona", ".EM.WIMP", ".EM.lengthAndIdentitiesPerMappingUnit", ".EM.contigCoverage"))
            variants = {}
            for wv in [x for x in os.environ.get("MM_BENCH_E2E_WORKER_SWEEP", "").split(",") if x]:   # the mapping phase again with other numbers of worker contexts per device
                env_v = dict(env)                                  # "W" or "W@S": W worker contexts per device, S of them inside their mapping section at a time
                if "@" in wv:
                    env_v["MM_CLI_MAP_SLOTS"] = wv.split("@")[1]
                if wv.count("@") > 1:                              # "W@S@B": batches of up to B Mbases
                    env_v["MM_CLI_BATCH_MBASES"] = wv.split("@")[2]
                _o3, e_v, t_v, _r = _run_cli_with_rss([cli, "mapDirectly", "--all", "-r", fasta, "-q", fq_s, "-o", pre_s + "_w" + wv, "--workers-per-gpu", wv.split("@")[0]], env_v, 1500)
                lv = {ln.split(" at +")[0][len("INFO, lap "):]: float(ln.split(" at +")[1].split()[0]) for ln in e_v.splitlines() if ln.startswith("INFO, lap ")}
                pv = {" ".join(ln.split()[2:-2]): float(ln.split()[-2]) for ln in e_v.splitlines() if ln.startswith("INFO, time ")}
                variants[wv] = {"mapping_phase_s": round(lv.get("8 write", t_v) - lv.get("3 index build", 0.0), 3), "phases": pv,
                                "same_file": open(pre_s).read() == open(pre_s + "_w" + wv).read()}
                os.remove(pre_s + "_w" + wv)
            # measurement aid: the stream's mapDirectly again under other environments / flags ("name:KEY=VAL KEY2=VAL2 --flag value;name2:...")
            for spec in [x for x in os.environ.get("MM_BENCH_E2E_STREAM_VARIANTS", "").split(";") if x]:
                vname, _, rest = spec.partition(":")
                venv, vflags = dict(env), []
                for tok in rest.split():
                    if "=" in tok and not tok.startswith("-"):
                        kk, _, vv = tok.partition("="); venv[kk] = vv
                    else:
                        vflags.append(tok)
                for rep in range(int(os.environ.get("MM_BENCH_E2E_VARIANT_REPS", 2))):
                    _o5, e_v, t_v, _r = _run_cli_with_rss([cli, "mapDirectly", "--all", "-r", fasta, "-q", fq_s, "-o", pre_s + "_v"] + vflags, venv, 1500)
                    lv = {ln.split(" at +")[0][len("INFO, lap "):]: float(ln.split(" at +")[1].split()[0]) for ln in e_v.splitlines() if ln.startswith("INFO, lap ")}
                    maps = sorted(float(ln.split(" map ")[1].split()[0]) for ln in e_v.splitlines() if ln.startswith("INFO, worker") and " map " in ln)
                    variants.setdefault("env " + vname, []).append({"mapping_phase_s": round(lv.get("8 write", t_v) - lv.get("3 index build", 0.0), 3),
                                                                    "map_section_s": {"median": maps[len(maps) // 2] if maps else None, "max": maps[-1] if maps else None, "above_40ms": sum(1 for m_ in maps if m_ > 0.04)},
                                                                    "same_file": open(pre_s).read() == open(pre_s + "_v").read()})
                    if os.environ.get("MM_BENCH_E2E_LOG"):
                        open(os.path.join(os.environ["MM_BENCH_E2E_LOG"], f"cli_stream_map_{vname}_{rep}.err"), "w").write(e_v)
                    os.remove(pre_s + "_v")
            cph = {ln.split()[2] + " " + " ".join(ln.split()[3:-2]): float(ln.split()[-2]) for ln in e_c.splitlines() if ln.startswith("INFO, time c")}
            cmain = {ln.split(" at +")[0][12:]: float(ln.split(" at +")[1].split()[0]) for ln in e_c.splitlines() if ln.startswith("INFO, main: ") and " at +" in ln}
            t_phase_s = max(laps_s.get("8 write", t_map_s) - laps_s.get("3 index build", 0.0), 1e-9)
            # the process without what it WAITED for its contexts (they come up beside the parsing of the mappings file; right behind a mapDirectly
            # that has just given 150 GB back, the driver lets the next process' HIP initialisation wait up to 2 s: not classify's work)
            ctx_wait_s = sum(float(ln.split()[-2]) for ln in e_c.splitlines() if ln.startswith("INFO, main: waited for the contexts"))
            t_cls_work_s = max(t_cls_s - ctx_wait_s, 1e-9)
            meta_s = dict(l.split() for l in open(pre_s + ".meta"))
            r_busy, r_thr = ph_s.get("R parse threads busy (summed over the block parser's threads)"), ph_s.get("R parse threads")
            reader_parse_s = (r_busy / r_thr) if r_busy and r_thr else None
            stream = {"what": f"the drop-in CLI in steady state: {n_stream} distinct batches ({reads_s} reads, one {os.path.getsize(fq_s) / 1e9:.1f} GB FASTQ).  value = read bases / (index built -> last classify "
                              "file written) of ONE process, `metamaps mapDirectly --all --then-classify DB`, by the CLI's own laps, nothing excluded (SURVEY D1's metric at the boundary users see; the index "
                              "build is reported apart).  two_processes: the reference's form, `mapDirectly` then `classify`, same files byte for byte — value_all_in counts the mapping phase + the whole "
                              "classify process (with what it waits for the driver right behind a process that gave 150 GB back), value_without_the_wait leaves that wait out (rounds 3-4 reported this one)",
                      "value": bases_s / t_tc_phase / 1e9,
                      "then_classify": {"index_built_to_last_file_s": round(t_tc_phase, 3), "mapping_phase_s": round(laps_t.get("8 write", 0.0) - laps_t.get("3 index build", 0.0), 3),
                                        "classify_s": round(laps_t.get("9 classify", 0.0) - laps_t.get("8 write", 0.0), 3), "process_wall_s": round(t_tc, 3),
                                        "same_files_as_two_processes": bool(tc_same), "cpu_quota_throttled_s": thr_tc, "laps_s": laps_t, "phases_s": ph_t, "peak_host_rss_bytes": int(rss_t)},
                      "two_processes": {"value_all_in": bases_s / (t_phase_s + t_cls_s) / 1e9, "value_without_the_wait": bases_s / (t_phase_s + t_cls_work_s) / 1e9},
                      "fastq_reader": {"bytes": os.path.getsize(fq_s), "parse_s": reader_parse_s, "GB_per_s": (os.path.getsize(fq_s) / reader_parse_s / 1e9) if reader_parse_s else None,
                                       "threads": r_thr,
                                       "what": "block-parallel parse of the mmap-ed FASTQ into batches: busy time summed over the parser's threads / their number = the seconds the file takes the reader when nothing holds it up (it starts when the reference is parsed, i.e. runs beside the index build, and waits for queue slots most of the time): bounds what the mapping phase could take from the reader"},
                      "index_build_note": "these runs follow one another within a second or two, each behind a process that held ~240 GB of the device: their index builds (laps '3 index build' - "
                                          "'1 reference parse + pack + upload', 5-6 s) wait inside hipMalloc for the driver to wipe what the process before released (~27 GB/s); the same build is 1.4-1.8 s "
                                          "with 15 s between the processes or as the second build of a process (profiles/r05_index_build_processes.txt, tools/index_build_repeat.py) - e2e_cli_full's "
                                          "mapDirectly, which starts 17 s after the bench freed its index, shows that case",
                      "batches": n_stream, "reads": int(reads_s), "bases": int(bases_s), "fastq_written_s": round(t_files_stream, 2),
                      "mapping_phase_s": round(t_phase_s, 3), "classify_work_s": round(t_cls_work_s, 3), "classify_waited_for_contexts_s": round(ctx_wait_s, 3), "classify_wall_s": round(t_cls_s, 3), "mapDirectly_wall_s": round(t_map_s, 3),
                      "unit": "Gbp/s", "mapping_phase_value": bases_s / t_phase_s / 1e9,
                      "map_laps_s": laps_s, "map_phases_s": ph_s, "classify_phases_s": cph, "classify_main_s": cmain,
                      "mappings_file_bytes": os.path.getsize(pre_s), "worker_sweep": variants, "peak_host_rss_bytes": {"mapDirectly": int(rss_s), "classify": int(rss_cs)},
                      "meta": {kk: int(v) for kk, v in meta_s.items()}}
        t_ingest = max(t_map - t_setup, 1e-9)                    # (process wall behind the index build: includes the driver's teardown of 150 GB at exit, ~0.5 s)
        t_phase = max(laps.get("8 write", t_map) - t_setup, 1e-9)  # the mapping phase by the CLI's own clock: index built -> last output file written
        t_cls_work = max(t_cls - sum(float(ln.split()[-2]) for ln in err2.splitlines() if ln.startswith("INFO, main: waited for the contexts")), 1e-9)
        return {"what": "metamaps mapDirectly --all (26.8 GB DB.fa + reads FASTQ -> PREFIX, .meta) + metamaps classify (-> .EM.*) on the bench reference and the bench's first read "
                        "batch, all host work inside (FASTA / FASTQ parse, 2-bit packing, H2D, index build, text formatting, file output, two process starts)",
                "reads": int(len(rl)), "bases": int(bases), "fasta_bytes": int(fasta_bytes), "window_derived_by_the_cli": int(par.get("windowSize", -1)),
                "input_files_written_s": round(t_files, 2),
                "mapDirectly_wall_s": round(t_map, 3), "of_which_reference_and_index_s": round(t_setup, 3), "classify_wall_s": round(t_cls, 3),
                "value": bases / (t_map + t_cls) / 1e9, "unit": "Gbp/s",
                "include_ingest": {"value": bases / (t_ingest + t_cls_work) / 1e9, "unit": "Gbp/s",
                                   "what": "the same reads with the reference already indexed and HIP initialised: FASTQ parse + pack + H2D + map + mapQ + D2H + text + write "
                                           f"({t_ingest:.3f} s) + classify without its wait for the contexts ({t_cls_work:.3f} s) — what the resident-data headline leaves out, in one number",
                                   "mapping_only_value": bases / t_ingest / 1e9,
                                   "mapping_phase_s": round(t_phase, 3), "mapping_phase_value": bases / t_phase / 1e9,
                                   "mapping_phase_what": "index built -> last output file written, by the CLI's own laps (without the process exit)"},
                "map_laps_s": laps, "map_phases_s": phases, "classify_phases_s": cls_phases, "classify_main_s": cls_main,
                "peak_host_rss_bytes": {"mapDirectly": int(rss_map), "classify": int(rss_cls)},
                "meta": {kk: int(v) for kk, v in meta.items()}, "variants": variants_1, "e2e_cli_stream": stream}
    finally:
        if own_tmp is not None:
            own_tmp.cleanup()


if __name__ == "__main__":
    main()

#!/bin/bash
# Regenerates the measurement artefacts committed under profiles/ (run on the GPU box through gpurun):
#   kernel-trace statistics of the default bench command (without its CPU / CLI legs) and of the same with the mapping sections serialised
#   (kernels that own the GPU), and HBM traffic per kernel from two separate PMC passes.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; rm -rf $out; mkdir -p $out
FLAGS="--no-cpu-baseline --no-other-shape --no-e2e-full"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 6 --warmup 3 $FLAGS > $out/bench_stats_run.json 2> $out/stats.err
# (MM_L2_ONE_STREAM=1: K5's two launches one behind the other, so that each duration is that of a kernel that owns the GPU and the pair is their sum; by default they run side by side)
MM_L2_ONE_STREAM=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_serialised -- python bench.py --steps 6 --warmup 2 --workers 1 $FLAGS > /dev/null 2> $out/stats_serialised.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --steps 1 --warmup 0 --workers 1 --distinct-batches 1 $FLAGS > /dev/null 2> $out/fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --steps 1 --warmup 0 --workers 1 --distinct-batches 1 $FLAGS > /dev/null 2> $out/write.err
python tools/summarize_profiles.py $out
# K5 phase clocks on the bench batch (10^5 x 10 kb), the probed-list distribution and the random-request ceiling (round 5)
# (the zone kernel's clocks: build them in before calling gpurun — tools/ab_build.sh clocks "-DL2Z_CLOCKS")
timeout 600 python tools/l2_long_phases.py 10000 10000 100000 > $out/l2_phases.txt 2>&1
if [ -n "$EXTRAS" ]; then   # (round 5's K3 studies: the probed-list distribution, the random-request ceiling, the CU sweep, LDS rates, K3's phase clocks)
timeout 600 python tools/probed_lists.py > $out/probed_lists.json 2> /dev/null
(for s in 2m 16m 128m 40; do ./tools/ubench/randread $s; done) > $out/randread.txt 2>&1
# what bounds K3 (round 5): its time against the CUs at work, the random-request ceiling by CUs / waves / loads in flight, LDS instruction rates, its phase clocks
timeout 600 python tools/sf_grid_sweep.py > $out/sf_grid_sweep.txt 2>&1
timeout 300 ./tools/ubench/randread_cus 90 > $out/randread_cus.txt 2>&1
timeout 300 ./tools/ubench/lds_rates > $out/lds_rates.txt 2>&1
(MM_SF_PROF=1 SHAPE=community ITERS=1 timeout 600 python tools/stage_ms.py 2>&1 | grep "MM_SF_PROF\|ms_hit_filter" | tail -3) > $out/sf_phases.txt 2>&1
fi
timeout 600 python tools/l2z_stops.py > $out/l2z_stops.txt 2>&1
timeout 600 python tools/l2z_pivot_hist.py 10000 10000 20000 > $out/l2z_pivot_hist.txt 2>&1
# instruction counts, busy cycles, waits of the three big kernels (one bench step each set, K5's launches one behind the other)
bash tools/k5_counters.sh > $out/sq_counters.txt 2>&1
tail -70 $out/sq_counters.txt

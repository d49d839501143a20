#!/bin/bash
# Kernel statistics and SQ counters of `bench.py --config 3` (mixed 1-50 kb PacBio reads against four resident chunk indexes), run on the
# GPU box through gpurun.  Output: gpurun_out/config3/{kernel_stats_config3.txt, sq_counters_config3.txt, bench_config3.json}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/config3; rm -rf $out; mkdir -p $out
F="--config 3 --no-cpu-baseline --no-other-shape --no-e2e-full"
timeout 900 python bench.py $F --steps 12 --warmup 2 > $out/bench_config3.json 2> $out/bench.err
MM_L2_ONE_STREAM=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py $F --steps 4 --warmup 1 --serialise-map > $out/bench_stats_run.json 2> $out/stats.err
python - $out <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
fs = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(fs[0])))
def short(name):
    n = name.split("(")[0].replace("void ", "")
    if "rocprim" in n:
        n = "rocprim::" + ("radix_sort_onesweep" if "onesweep" in name else "radix_sort_histogram" if "histogram" in name else "other")
    return n[:70]
with open(os.path.join(out, "kernel_stats_config3.txt"), "w") as w:
    w.write("# MM_L2_ONE_STREAM=1 rocprofv3 --kernel-trace --stats -- python bench.py --config 3 --steps 4 --warmup 1 --serialise-map --no-cpu-baseline --no-other-shape --no-e2e-full\n")
    w.write(f"{'calls':>7} {'total_ms':>12} {'avg_ms':>12} {'%':>7}  kernel\n")
    for r in rows[:60]:
        w.write(f"{int(r['Calls']):>7} {float(r['TotalDurationNs'])/1e6:>12.3f} {float(r['AverageNs'])/1e6:>12.3f} {float(r['Percentage']):>7.3f}  {short(r['Name'])}\n")
print(open(os.path.join(out, "kernel_stats_config3.txt")).read())
PY
G="$F --steps 1 --warmup 0 --workers 1 --distinct-batches 1"
(for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "VALUBusy SALUBusy MemUnitBusy"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  MM_L2_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$n -- python bench.py $G > $out/pmc_$n.json 2> $out/pmc_$n.err
  f=$(find $out/pmc_$n -name "*counter_collection.csv" | head -1)
  echo "== $set"
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    if any(t in k for t in ("l2_", "l2z_", "seed_filter", "probe_kernel", "hit_filter", "minimizer_kernel<2>", "sketch_", "sort_hits")):
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items(): print(f"  {k[:60]:60s} {c:24s} {v:.6g}  ({cnt[(k, c)]} launches)")
PY
done) > $out/sq_counters_config3.txt 2>&1
tail -1 $out/bench_config3.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config3', d['value'], d['ms_per_step']); print(json.dumps(d['config'].get('per_step'))); print(json.dumps(d['config'].get('stage_ms_sum', d['config'].get('stage_ms'))))"
cat $out/sq_counters_config3.txt

#!/bin/bash
# kernel statistics of one bench command with ONE worker context (every duration is that of a kernel that owns the GPU): tools/kstats.sh OUTNAME [ENV=VAL ...] -- bench flags
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
name=$1; shift
while [ "$1" != "--" ] && [ -n "$1" ]; do export "$1"; shift; done
shift
out=gpurun_out/kstats_$name; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --workers 1 --no-cpu-baseline --no-other-shape --no-e2e-full "$@" > $out/bench.json 2> $out/err.txt
python - $out "$*" <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
fs = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(fs[0])))
def short(name):
    n = name.split("(")[0].replace("void ", "")
    if "rocprim" in n:
        n = "rocprim::" + ("radix_sort_onesweep" if "onesweep" in name else "radix_sort_histogram" if "histogram" in name else "segmented_sort" if "segmented" in name else "other")
    return n[:70]
d = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
with open(os.path.join(out, "kernel_stats.txt"), "w") as w:
    w.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --workers 1 --no-cpu-baseline --no-other-shape --no-e2e-full {sys.argv[2]}   ({d['ms_per_step']:.1f} ms per step, {d['value']:.2f} Gbp/s under the profiler)\n")
    w.write(f"{'calls':>7} {'total_ms':>12} {'avg_ms':>12} {'%':>7}  kernel\n")
    for r in rows[:70]:
        w.write(f"{int(r['Calls']):>7} {float(r['TotalDurationNs'])/1e6:>12.3f} {float(r['AverageNs'])/1e6:>12.3f} {float(r['Percentage']):>7.3f}  {short(r['Name'])}\n")
print(open(os.path.join(out, "kernel_stats.txt")).read())
PY
rm -rf $out/stats

#!/bin/bash
# Where does a SMALL batch lose its efficiency?  (configs[2]'s per-GPU share, 12 500 reads, runs at 82 % of the 10^5-read rate; the CLI's 0.2 Gbp batches at 70 %.)
# Stage times of one batch alone at 12 500 / 25 000 / 100 000 reads, and the kernel trace of the strong-scaling bench: sum of kernel time per step against the step.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/small_batch; rm -rf $out; mkdir -p $out
for nr in 12500 25000 100000; do echo "== NR=$nr"; SHAPE=community NR=$nr ITERS=4 python tools/stage_ms.py 2>/dev/null | head -1; done > $out/stage_ms.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --scaling strong --reads 12500 --steps 40 --warmup 5 --no-cpu-baseline --no-other-shape --no-e2e-full > $out/bench.json 2> $out/trace.err
python - <<'PY'
import csv, glob, json, collections
out = "gpurun_out/small_batch"
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("bench under rocprof: ms_per_step", round(d["ms_per_step"], 3), "value", round(d["value"], 2))
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "minimizer_kernel<2>" in r[2]]
# the last 40 steps: from the 41st-last K1 launch to the last one
a, b = starts[-41], starts[-1]
seg = rows[a:b]
wall = (seg[-1][1] - seg[0][0]) / 1e6
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in seg: acc[n][0] += 1; acc[n][1] += (e - s) / 1e6
busy, end = 0.0, seg[0][0]
for s, e, n in seg:
    if e > end: busy += (e - max(s, end)) / 1e6; end = e
print(f"40 steps: wall {wall:.1f} ms = {wall / 40:.3f} per step; GPU busy (union of kernel intervals) {busy / 40:.3f} per step; sum of kernel durations {sum(v[1] for v in acc.values()) / 40:.3f} per step; {len(seg) / 40:.0f} launches per step")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:22]: print(f"  {t / 40:8.3f} ms/step  {c / 40:6.1f} launches/step  {n}")
PY
cat $out/stage_ms.txt

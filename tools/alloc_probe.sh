#!/bin/bash
# which allocations reach the driver (or a slab) inside the bench's timed region: tools/alloc_probe.sh [RUNS]
cd $GRAFT_REPO_ROOT
for i in $(seq ${1:-3}); do
MM_ALLOC_TRACE=1 python bench.py --no-cpu-baseline --no-other-shape --no-e2e-full > gpurun_out/ap_$i.json 2> gpurun_out/ap_$i.err
python - $i <<'PY'
import json,sys,re
i=sys.argv[1]
d=json.loads(open(f"gpurun_out/ap_{i}.json").read().strip().splitlines()[-1])
print(i, round(d["ms_per_step"],2), d["step_ms"]["all"])
t0=t1=None; ev=[]
for ln in open(f"gpurun_out/ap_{i}.err"):
    if "timed region starts" in ln: t0=float(ln.split(" at ")[1].split()[0])
    elif "timed region ends" in ln: t1=float(ln.split(" at ")[1].split()[0])
    elif ln.startswith("MM_ALLOC_TRACE") and " at " in ln: ev.append((float(ln.rsplit(" at ",1)[1].split()[0]), ln.strip()))
for t,l in ev:
    if t0 and t >= t0 and (t1 is None or t <= t1): print(f"   +{t-t0:7.1f} ms  {l[:110]}")
PY
done

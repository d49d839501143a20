#!/bin/bash
# HBM traffic of K5's launches for one bench step (FETCH_SIZE / WRITE_SIZE in separate passes, KiB; corrections as in tools/summarize_profiles.py): tools/k5_traffic.sh [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
F="--steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full"
for c in FETCH_SIZE WRITE_SIZE; do
  MM_L2_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/k5t/$c -- python bench.py $F > /dev/null 2> gpurun_out/k5t_$c.err
  f=$(find gpurun_out/k5t/$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    if "l2z_kernel" in k or "l2_kernel" in k: acc[k] += float(row["Counter_Value"])
for k, v in acc.items(): print(sys.argv[2], k, f"{v * 1024 * (2 if sys.argv[2] == 'FETCH_SIZE' else 1) / 1e9:.2f} GB")
PY
done
rm -rf gpurun_out/k5t gpurun_out/k5t_*.err

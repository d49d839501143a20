#!/bin/bash
# HBM bytes (FETCH_SIZE x 2 correction / WRITE_SIZE, separate passes) of the zone kernel when it leaves after a phase (MM_L2_STOP, see tools/l2z_stops.py), one bench step each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
F="--steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full --no-e2e-stream"
for st in ${STOPS:-1 2 3 4 8 5 0}; do
  line="stop $st"
  for c in FETCH_SIZE WRITE_SIZE; do
    MM_L2_STOP=$st MM_L2_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/k5st/$st$c -- python bench.py $F > /dev/null 2> gpurun_out/k5st_$st.err
    f=$(find gpurun_out/k5st/$st$c -name "*counter_collection.csv" | head -1)
    v=$(python - "$f" $c <<'PY'
import csv, sys
acc = 0.0
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "l2z_kernel" in k or "l2_kernel" in k: acc += float(row["Counter_Value"])
print(f"{sys.argv[2]} {acc * 1024 * (2 if sys.argv[2] == 'FETCH_SIZE' else 1) / 1e9:.2f} GB")
PY
)
    line="$line  $v"
  done
  echo "$line"
done
rm -rf gpurun_out/k5st gpurun_out/k5st_*.err

#!/bin/bash
# HBM traffic of the K5/K6 kernel by phase: FETCH_SIZE / WRITE_SIZE (separate passes) with MM_L2_STOP=n (the kernel leaves after phase n;
# 1 setup, 2 pass A, 4 pass B + bounds, 8 first rebuild, 9 first slide round, 5 whole sweep, 0 everything), bench workload, one step.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/l2_traffic; rm -rf $out; mkdir -p $out
for st in ${STOPS:-1 2 4 8 9 5 0}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    MM_L2_STOP=$st timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${c}_$st -- python bench.py --steps 1 --warmup 0 --workers 1 --no-cpu-baseline --no-other-shape > /dev/null 2> $out/${c}_$st.err
  done
  python - $out $st <<'PY'
import csv, sys, glob, collections
out, st = sys.argv[1], sys.argv[2]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = 0.0
    for f in glob.glob(f"{out}/{c}_{st}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "l2_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                acc += float(row["Counter_Value"])
    tot[c] = acc
# FETCH_SIZE counts 64-byte units at half for wide requests (x2, profiles/r01_fetch_size_calibration.txt); WRITE_SIZE in KB units of 64 B likewise as collected before
print(f"stop {st}: FETCH_SIZE {tot['FETCH_SIZE']:.4g}  WRITE_SIZE {tot['WRITE_SIZE']:.4g}", flush=True)
PY
done

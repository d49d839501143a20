#!/bin/bash
# VALU / SALU / LDS instruction counts of the zone kernel when it leaves after a phase (MM_L2_STOP, see tools/l2z_stops.py), one bench step each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
F="--steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full --no-e2e-stream"
for st in ${STOPS:-1 2 3 4 8 5 0}; do
  MM_L2_STOP=$st MM_L2_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/k5s/$st -- python bench.py $F > /dev/null 2> gpurun_out/k5s_$st.err
  f=$(find gpurun_out/k5s/$st -name "*counter_collection.csv" | head -1)
  python - "$f" $st <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "l2z_kernel" in k or "l2_kernel" in k: acc[row["Counter_Name"]] += float(row["Counter_Value"])
print("stop", sys.argv[2], " ".join(f"{c} {v:.4g}" for c, v in sorted(acc.items())))
PY
done
rm -rf gpurun_out/k5s gpurun_out/k5s_*.err

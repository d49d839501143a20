#!/bin/bash
# SQ counters of K5 (l2_kernel / l2z_kernel launches) for one bench step: tools/k5_counters.sh [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
F="--steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full ${K5_BENCH_FLAGS}"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "VALUBusy SALUBusy MemUnitBusy" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  MM_L2_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/k5c/pmc_$n -- python bench.py $F > gpurun_out/k5c_$n.json 2> gpurun_out/k5c_$n.err
  f=$(find gpurun_out/k5c/pmc_$n -name "*counter_collection.csv" | head -1)
  echo "== $set"
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    if "l2_kernel" in k or "l2z_kernel" in k or "seed_filter_stream" in k or "minimizer_kernel<2>" in k:
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in acc:
    for c, v in acc[k].items(): print(f"  {k[:50]:50s} {c:24s} {v:.6g}  ({cnt[(k, c)]} launches)")
PY
done
python - <<'PY'
import json
d = json.loads(open("gpurun_out/k5c_VALUBusy_SALUBusy_MemUnitBusy.json").read().strip().splitlines()[-1])
print("per_step", json.dumps(d["config"]["per_step"]))
PY
rm -rf gpurun_out/k5c

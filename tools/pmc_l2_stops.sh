# per-phase instruction counts of the K5/K6 kernel: rocprofv3 counters with MM_L2_STOP=n (cumulative phases)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for st in 2 3 4 5 0; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
    n=$(echo $set | tr ' ' '_' | cut -c1-30)_$st
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcs_$n -- python tools/l2_stop.py $st > /dev/null 2> gpurun_out/pmcs_$n.err
    f=$(find gpurun_out/pmcs_$n -name "*counter_collection.csv" | head -1)
    python - "$f" $st <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = 0
for row in csv.DictReader(open(sys.argv[1])):
    if "l2_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items(): print(f"stop {sys.argv[2]} {k:24s} {v/3:.5g}")
PY
  done
done

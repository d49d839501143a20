cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "VALUBusy SALUBusy" "OccupancyPercent MemUnitStalled" ; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$n -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_$n.err
  f=$(find gpurun_out/pmc_$n -name "*counter_collection.csv" | head -1)
  echo "== $set ($f)"
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        if "l2_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
except Exception as e:
    print("ERR", e)
for k, v in acc.items(): print(f"  {k:28s} {v:.6g}")
PY
done

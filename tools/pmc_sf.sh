#!/bin/bash
# instruction mix and issue utilisation of the fused probe + filter kernel (K3) on the bench workload, one step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sf; rm -rf $out; mkdir -p $out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -- python bench.py --steps 1 --warmup 0 --workers 1 --no-cpu-baseline --no-other-shape > /dev/null 2> $out/$n.err
done
python - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for kn in ("seed_filter_kernel", "l2_kernel<true, unsigned char, 4, 2>", "minimizer_kernel<2>"):
            if kn in row["Kernel_Name"]:
                acc[(kn, row["Counter_Name"])] += float(row["Counter_Value"])
for k in sorted(acc): print(f"{k[0]:40s} {k[1]:24s} {acc[k]:.5g}")
PY

#!/bin/bash
# instruction mix and issue utilisation of the fused probe + filter kernel (K3) on the bench workload, one step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sf; rm -rf $out; mkdir -p $out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -- python bench.py --steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full > /dev/null 2> $out/$n.err
done
python - $out <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(float)
names = {"seed_filter_stream_kernel": "K3 seed_filter_stream_kernel", "l2_kernel<true, unsigned char, 4, 2>": "K5 l2_kernel<true,u8,4,2>", "minimizer_kernel<2>": "K1 minimizer_kernel<2>"}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for kn in names:
            if kn in row["Kernel_Name"]:
                acc[(kn, row["Counter_Name"])] += float(row["Counter_Value"])
print("# rocprofv3 --kernel-trace --pmc <SQ / GRBM counters, five separate passes> -- python bench.py --steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full")
print("# one launch each; counters are sums over the 8 XCDs / 32 shader engines / 256 CUs.  issue slots = 1024 SIMDs x active cycles / 4 (a wave64 VALU instruction")
print("# occupies its SIMD for four cycles); clock 2.4 GHz.")
print(f"{'kernel':28s} {'active ms':>9s} {'VALU instr':>11s} {'VALU slots used':>15s} {'SALU/VALU':>9s} {'LDS instr':>10s} {'LDS busy':>8s} {'bank conflict cycles':>20s} {'waves waiting':>13s} {'VMEM rd instr':>13s}")
for kn, nm in names.items():
    g = lambda c: acc.get((kn, c), 0.0)
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    if cyc <= 0: continue
    slots = 1024.0 * cyc / 4.0
    print(f"{nm:28s} {cyc / 2.4e6:9.2f} {g('SQ_INSTS_VALU'):11.4g} {g('SQ_INSTS_VALU') / slots:15.2f} {g('SQ_INSTS_SALU') / max(g('SQ_INSTS_VALU'), 1):9.2f} {g('SQ_INSTS_LDS'):10.4g} "
          f"{g('SQ_LDS_IDX_ACTIVE') / (256.0 * cyc):8.2f} {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):20.2f} {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):13.2f} {g('SQ_INSTS_VMEM_RD'):13.4g}")
PY

#!/bin/bash
# K5 (l2_kernel): where do its re-reads go?  L2 (TCC) requests, hits and misses, and the requests the L2 sends on to the fabric (EA), for one
# bench step.  FETCH_SIZE (profiles/r03_pmc_hbm_traffic.txt) counts what leaves the L2; whether such a request is then served by the
# memory-side Infinity Cache or by HBM is not visible in the TCC counters.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/l2_cache; rm -rf $out; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -oE "TCC_[A-Za-z0-9_]+" | sort -u | tr '\n' ' ' > $out/avail.txt
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -- python bench.py --steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full > /dev/null 2> $out/$n.err
done
python - $out <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(float)
names = {"l2_kernel<true, unsigned char, 4, 2>": "K5 l2_kernel<true,u8,4,2>", "l2_kernel<true, unsigned char, 2, 2>": "K5 l2_kernel<true,u8,2,2>", "seed_filter_stream_kernel": "K3 seed_filter_stream_kernel", "minimizer_kernel<2>": "K1 minimizer_kernel<2>"}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for kn in names:
            if kn in row["Kernel_Name"]:
                acc[(kn, row["Counter_Name"])] += float(row["Counter_Value"])
print("# rocprofv3 --kernel-trace --pmc <TCC counters, separate passes> -- python bench.py --steps 1 --warmup 0 --workers 1 --distinct-batches 1 --no-cpu-baseline --no-other-shape --no-e2e-full")
print("# one launch each, sums over the 16 x 8 L2 channels.  L2 request = one 128-byte line access (TCC_REQ); EA RDREQ = read request the L2 sends to the fabric")
print("# (64 bytes unless counted under _32B); hit share = TCC_HIT / (TCC_HIT + TCC_MISS)")
cols = ["TCC_REQ_sum", "TCC_READ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum"]
print(f"{'kernel':30s} " + " ".join(f"{c[4:-4]:>16s}" for c in cols) + f" {'L2 hit share':>13s} {'EA read GB':>11s}")
for kn, nm in names.items():
    g = lambda c: acc.get((kn, c), 0.0)
    if g("TCC_REQ_sum") <= 0 and g("TCC_HIT_sum") <= 0: continue
    hs = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0)
    ea_gb = ((g("TCC_EA0_RDREQ_sum") - g("TCC_EA0_RDREQ_32B_sum")) * 64 + g("TCC_EA0_RDREQ_32B_sum") * 32) / 1e9
    print(f"{nm:30s} " + " ".join(f"{g(c):16.5g}" for c in cols) + f" {hs:13.3f} {ea_gb:11.2f}")
PY
